cd $GRAFT_REPO_ROOT
TTX_DE_LANE=1 timeout -k 10 600 python3 -m pytest tests/test_gpu_parity.py -m gpu -x -q -k "ising_sweep_bit_exact or bond_groups" 2>&1 | tail -4 || exit 1
for v in "TTX_DE_LANE=1" "TTX_DE_LANE=0"; do
  echo "== $v: $(env $v timeout -k 10 600 python3 bench.py --workload d256 --arith exact --steps 1 --warmup 0 --no-cpu-baseline --no-extras 2>/dev/null | python3 -c 'import json,sys; j=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(round(j["ms_per_step"],2), "ms", j["config"]["sweeps"], j["config"]["integral"], j["value"])')"
done
