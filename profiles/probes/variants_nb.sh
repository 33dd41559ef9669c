# workgroups per bond group of the cluster kernel (C_64 and C_16, final kernels of round 3)
cd $GRAFT_REPO_ROOT
for nb in 4 6 8 10 12 16; do for w in c64 c16; do
  echo "== TTX_CLUSTER_NB=$nb $w: $(TTX_CLUSTER_NB=$nb timeout -k 10 300 python3 bench.py --workload $w --steps 20 --warmup 5 --no-cpu-baseline --no-extras 2>/dev/null | python3 -c 'import json,sys; j=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(round(j["ms_per_step"],3), "ms")')"
done; done
