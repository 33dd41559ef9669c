#!/bin/bash
# per-role time stamps of k_halfstep_det (TTX_STAMPS builds; group 3 = mid chain, group 0 = chain end), D_256
mkdir -p gpurun_out
for gsel in 3 0; do
  TTX_LIB=$PWD/ttcross_amd/lib/libttx_stamps$gsel.so timeout -k 10 300 python bench.py --workload d256 --steps 1 --warmup 0 --no-extras --no-cpu-baseline > gpurun_out/team_stamps_$gsel.json 2> gpurun_out/team_stamps_$gsel.err
  echo "group $gsel:"; grep "stamps kernel 1" gpurun_out/team_stamps_$gsel.err | tail -1
done
