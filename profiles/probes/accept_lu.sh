cd $GRAFT_REPO_ROOT
export TTX_LIB=$GRAFT_REPO_ROOT/ttcross_amd/lib/libttx_hl.so
timeout -k 10 800 python3 -m pytest tests/test_gpu_parity.py -m gpu -x -q -k "variants or bond_groups or ising_sweep" 2>&1 | tail -2 || exit 1
unset TTX_LIB
for v in "TTX_DUMMY=1" "TTX_LIB=$GRAFT_REPO_ROOT/ttcross_amd/lib/libttx_hl.so" "TTX_DUMMY=1" "TTX_LIB=$GRAFT_REPO_ROOT/ttcross_amd/lib/libttx_hl.so"; do
for w in "d256 exact 2 1"; do set -- $w
  echo "== ${v:0:12} $1 $2: $(env $v timeout -k 10 300 python3 bench.py --workload $1 --arith $2 --steps $3 --warmup $4 --no-cpu-baseline --no-extras 2>/dev/null | python3 -c 'import json,sys; j=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(round(j["ms_per_step"],3), "ms", j["config"]["integral"], {k: round(v,2) for k,v in j["kernel_ms_per_step"].items()})')"
done; done
