#!/bin/bash
# D_256: thresholds of the 14-wave (TEAM_UNITS) and 6-wave (TEAM6_UNITS) teams
mkdir -p gpurun_out
for pair in ${PAIRS:-0:1024 128:1024 256:1024 256:2048}; do
  u=${pair%%:*}; v=${pair##*:}
  TTX_DE_TEAM6_UNITS=$v TTX_DE_TEAM_UNITS=$u timeout -k 10 300 python bench.py --workload d256 --steps 1 --warmup 0 --no-extras --no-cpu-baseline > gpurun_out/team6_${u}_${v}.json 2>gpurun_out/team6_${u}_${v}.err
  python - <<P
import json
d=json.load(open("gpurun_out/team6_${u}_${v}.json"))
print("team14<=$u team6<=$v", round(d["ms_per_step"],1), round(d["kernel_ms_per_step"]["halfstep"],1), d["config"].get("integral"), d["config"].get("neval_per_step"))
P
done
