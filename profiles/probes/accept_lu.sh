cd $GRAFT_REPO_ROOT
timeout -k 10 600 python3 -m pytest tests/test_gpu_parity.py -m gpu -x -q -k "ising_sweep_bit_exact or bond_groups" 2>&1 | tail -2 || exit 1
for v in "TTX_SWEEP_TAIL=1" "TTX_SWEEP_TAIL=0" "TTX_SWEEP_TAIL=1" "TTX_SWEEP_TAIL=0"; do
for w in "c64 exact 20 5" "c16 exact 20 5"; do set -- $w
  echo "== $v $1 $2: $(env $v timeout -k 10 300 python3 bench.py --workload $1 --arith $2 --steps $3 --warmup $4 --no-cpu-baseline --no-extras 2>/dev/null | python3 -c 'import json,sys; j=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(round(j["ms_per_step"],3), "ms", j["config"]["integral"], j["config"]["sweeps"])')"
done; done
