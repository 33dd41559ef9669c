cd $GRAFT_REPO_ROOT
for w in c6 c16 c64 d32 d64 d256 mvn128; do for ar in exact fast; do
  echo "== $w $ar: $(timeout -k 10 300 python3 bench.py --workload $w --arith $ar --steps 3 --warmup 1 --no-cpu-baseline --no-extras 2>&1 | tail -1 | python3 -c 'import json,sys
try:
    j=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(round(j["ms_per_step"],3), "ms", round(j["value"]/1e6,1), "M evals/s", j["config"]["sweeps"], "sweeps", j["config"]["integral"], j["config"]["arith"], j["config"]["sweep_path"])
except Exception as e: print("ERR", e)')"
done; done
