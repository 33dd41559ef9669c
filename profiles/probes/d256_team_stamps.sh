#!/bin/bash
# per-role time stamps of k_halfstep_det (TTX_STAMPS builds; group 3 = mid chain, group 0 = chain end), D_256.
# Run through gpurun from the repo root; the two debug libraries are built here (hipcc is on the GPU box) and not kept in the tree.
mkdir -p gpurun_out
for gsel in 3 0; do
  hipcc --offload-arch=gfx950 -O3 -std=c++17 -ffp-contract=off -fPIC -shared -Iinclude -DTTX_STAMPS -DDET_STAMPG=$gsel -o /tmp/libttx_stamps$gsel.so ttcross_amd/csrc/ttx_engine.hip || exit 1
  TTX_LIB=/tmp/libttx_stamps$gsel.so timeout -k 10 300 python bench.py --workload d256 --steps 1 --warmup 0 --no-extras --no-cpu-baseline > gpurun_out/team_stamps_$gsel.json 2> gpurun_out/team_stamps_$gsel.err
  echo "group $gsel:"; grep "stamps kernel 1" gpurun_out/team_stamps_$gsel.err | tail -1
done
