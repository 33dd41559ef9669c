# development: time the C_64 bench with several builds of libttx (TTX_LIB)
for v in "$@"; do
  TTX_LIB=$PWD/ttcross_amd/lib/libttx_v_$v.so timeout -k 10 120 python bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-extras 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('$v', round(d['ms_per_step'],3), 'ms', d['config']['integral'])"
done
