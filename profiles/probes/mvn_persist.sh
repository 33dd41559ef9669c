cd $GRAFT_REPO_ROOT
timeout -k 10 900 python3 -m pytest tests/test_gpu_fast.py tests/test_gpu_boundary.py -m gpu -q -k "mvn" 2>&1 | grep -v "^  \|^$" | tail -30
for v in "TTX_FAST_PERSIST=0" "TTX_FAST_PERSIST=1"; do
  echo "== $v: $(env $v timeout -k 10 300 python3 bench.py --workload mvn128 --arith fast --steps 3 --warmup 1 --no-cpu-baseline --no-extras 2>/dev/null | python3 -c 'import json,sys; j=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(round(j["ms_per_step"],2), "ms", j["config"]["sweeps"], j["config"]["integral"], j["value"])')"
done
