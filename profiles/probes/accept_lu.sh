cd $GRAFT_REPO_ROOT
timeout -k 10 800 python3 -m pytest tests/test_gpu_parity.py -m gpu -x -q -k "variants or config5 or full_size_d256 or bond_groups" 2>&1 | tail -2 || exit 1
TTX_DE_LOT_POINT=0 timeout -k 10 600 python3 -m pytest tests/test_gpu_parity.py tests/test_gpu_campaigns.py -m gpu -x -q -k "ising_sweep_bit_exact or fuzz_random" 2>&1 | tail -2 || exit 1
for w in "d256 exact 2 1" "d256 exact 2 1"; do set -- $w
  echo "== $1 $2: $(timeout -k 10 300 python3 bench.py --workload $1 --arith $2 --steps $3 --warmup $4 --no-cpu-baseline --no-extras 2>/dev/null | python3 -c 'import json,sys; j=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(round(j["ms_per_step"],3), "ms", j["config"]["sweeps"], j["config"]["integral"], {k: round(v,2) for k,v in j["kernel_ms_per_step"].items()})')"
done
