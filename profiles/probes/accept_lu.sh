cd $GRAFT_REPO_ROOT
timeout -k 10 900 python3 -m pytest tests/test_gpu_parity.py -m gpu -x -q -k "ising_sweep_bit_exact or bond_groups or variants" 2>&1 | tail -3 || exit 1
for v in "TTX_DE_LOT_POINT=0" "TTX_DE_LOT_POINT=1" "TTX_DE_LOT_POINT=0" "TTX_DE_LOT_POINT=1"; do
for w in "d256 exact 2 1"; do set -- $w
  echo "== $v $1 $2: $(env $v timeout -k 10 300 python3 bench.py --workload $1 --arith $2 --steps $3 --warmup $4 --no-cpu-baseline --no-extras 2>/dev/null | python3 -c 'import json,sys; j=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(round(j["ms_per_step"],3), "ms", {k: round(v,2) for k,v in j["kernel_ms_per_step"].items()})')"
done; done
