# N1: parity tests, then timings with thread-count variants
cd $GRAFT_REPO_ROOT
timeout -k 10 600 python3 -m pytest tests/test_gpu_parity.py tests/test_gpu_campaigns.py -m gpu -x -q -k "ort_svd or uploaded_train or tt_lib or ttio or tt_generics" 2>&1 | tail -5 || exit 1
for v in "TTX_QR_OWN=1" "TTX_JAC_THREADS=512" "TTX_JAC_THREADS=1024" "TTX_JAC_THREADS=128"; do
  for w in svd_d64; do
    echo "== $v $w: $(env $v timeout -k 10 300 python3 bench.py --workload $w --steps 5 --warmup 2 --no-cpu-baseline 2>/dev/null | python3 -c 'import json,sys; j=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(round(j["ms_per_step"],2), "ms")')"
  done
done
