for nb in 4 6 8 10 12 16; do
  TTX_CLUSTER_NB=$nb TTX_LIB=$PWD/ttcross_amd/lib/libttx_v_x.so timeout -k 10 120 python bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-extras 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('NB $nb', round(d['ms_per_step'],3), 'ms', d['config']['integral'])"
done
