#!/bin/bash
# D_256 with and without the 16-wave teams of k_halfstep_det (run through gpurun from the repo root)
set -e
mkdir -p gpurun_out
python -m pytest tests/test_gpu_parity.py -x -q -m gpu -k "de_kernel_variants or ising_sweep or bond_groups" > gpurun_out/team_tests.log 2>&1 || { tail -30 gpurun_out/team_tests.log; exit 1; }
tail -3 gpurun_out/team_tests.log
for u in ${TEAM_UNITS:-0 256 512}; do
  TTX_DE_TEAM6_UNITS=${TEAM6_UNITS:-1024} TTX_DE_TEAM_UNITS=$u timeout -k 10 300 python bench.py --workload d256 --steps 1 --warmup 0 --no-extras --no-cpu-baseline > gpurun_out/team_d256_$u.json 2>gpurun_out/team_d256_$u.err
  python - <<P
import json
d=json.load(open("gpurun_out/team_d256_$u.json"))
print("units<=$u", d["ms_per_step"], d["kernel_ms_per_step"], d["config"].get("integral"), d["config"].get("neval_per_step"))
P
done
