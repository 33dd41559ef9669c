# N1: parity tests, then svd timing with the polled report against copies + synchronisation
cd $GRAFT_REPO_ROOT
timeout -k 10 600 python3 -m pytest tests/test_gpu_parity.py tests/test_gpu_campaigns.py tests/test_gpu_boundary.py -m gpu -x -q -k "ort_svd or uploaded_train or tt_lib or ttio or tt_generics" 2>&1 | tail -3 || exit 1
for v in "TTX_SVD_POLL=1" "TTX_SVD_POLL=0" "TTX_SVD_POLL=1" "TTX_SVD_POLL=0"; do
  echo "== $v svd_d64: $(env $v timeout -k 10 300 python3 bench.py --workload svd_d64 --steps 8 --warmup 2 --no-cpu-baseline 2>/dev/null | python3 -c 'import json,sys; j=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(round(j["ms_per_step"],2), "ms")')"
done
