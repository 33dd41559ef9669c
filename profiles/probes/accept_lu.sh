cd $GRAFT_REPO_ROOT
timeout -k 10 900 python3 -m pytest tests/test_gpu_parity.py tests/test_gpu_fast.py tests/test_gpu_boundary.py -m gpu -x -q -k "ising_sweep_bit_exact or bond_groups or variants or host_callback_bit or fast or mvn or stdnorm" 2>&1 | tail -4 || exit 1
for w in "d256 fast" "d256 exact" "mvn128 fast" "mvn128 exact"; do set -- $w
  echo "== $1 $2: $(timeout -k 10 300 python3 bench.py --workload $1 --arith $2 --steps 2 --warmup 1 --no-cpu-baseline --no-extras 2>/dev/null | python3 -c 'import json,sys; j=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(round(j["ms_per_step"],2), "ms", j["config"]["sweeps"], j["config"]["integral"], {k: round(v,1) for k,v in j["kernel_ms_per_step"].items()})')"
done
