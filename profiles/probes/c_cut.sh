cd $GRAFT_REPO_ROOT
timeout -k 10 600 python3 -m pytest tests/test_gpu_parity.py -m gpu -x -q -k "ising_sweep_bit_exact or bond_groups" 2>&1 | tail -3 || exit 1
for v in "TTX_C_CUT=1" "TTX_C_CUT=0" "TTX_C_CUT=1" "TTX_C_CUT=0"; do
  echo "== $v: $(env $v timeout -k 10 300 python3 bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-extras 2>/dev/null | python3 -c 'import json,sys; j=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(round(j["ms_per_step"],3), "ms", j["config"]["integral"])')"
done
