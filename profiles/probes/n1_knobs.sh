for q in 1024 512 256; do for j in 1024 512 256; do
  TTX_QR_THREADS=$q TTX_JAC_THREADS=$j timeout -k 10 120 python bench.py --workload svd_d64 --steps 3 --warmup 1 --no-cpu-baseline 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('qr $q jac $j', round(d['ms_per_step'],2), 'ms ranks_out', d['config']['ranks_out_max'])"
done; done
TTX_QR_THREADS=512 timeout -k 10 120 python bench.py --workload ort_d64 --steps 3 --warmup 1 --no-cpu-baseline 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('ort qr 512', round(d['ms_per_step'],2))"
timeout -k 10 120 python bench.py --workload ort_d64 --steps 3 --warmup 1 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('ort qr 1024', round(d['ms_per_step'],2), d.get('cpu_baseline'))"
