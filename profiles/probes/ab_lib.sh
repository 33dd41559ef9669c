# A/B of two builds of the library in one call: bash profiles/probes/ab_lib.sh <other lib> <workload> <arith> [steps]
cd $GRAFT_REPO_ROOT
for v in "TTX_DUMMY=1" "TTX_LIB=$GRAFT_REPO_ROOT/$1" "TTX_DUMMY=1" "TTX_LIB=$GRAFT_REPO_ROOT/$1"; do
  echo "== ${v:0:9} $2 $3: $(env $v timeout -k 10 300 python3 bench.py --workload $2 --arith $3 --steps ${4:-3} --warmup 1 --no-cpu-baseline --no-extras 2>/dev/null | python3 -c 'import json,sys; j=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(round(j["ms_per_step"],3), "ms", j["config"]["integral"], {k: round(v,2) for k,v in j["kernel_ms_per_step"].items()})')"
done
