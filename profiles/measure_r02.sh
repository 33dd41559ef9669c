# round-2 measurements on the GPU box (run through gpurun): kernel statistics and counters of the heavy workloads
#   bash profiles/measure_r02.sh <tag> <workload> [pmc]
set -e
R=$GRAFT_REPO_ROOT; TAG=$1; WL=$2
O=$R/gpurun_out/$TAG
mkdir -p $O
cd /tmp && export TMPDIR=/tmp
rm -rf /tmp/st_$TAG /tmp/p1_$TAG /tmp/p2_$TAG
rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/st_$TAG -- python3 $R/bench.py --workload $WL --steps 1 --warmup 0 --no-cpu-baseline --no-extras > $O/bench_under_rocprof.json 2> $O/rocprof_stats.err
cp $(find /tmp/st_$TAG -name "*kernel_stats.csv" | head -1) $O/kernel_stats.csv
head -8 $O/kernel_stats.csv
if [ "$3" = "pmc" ]; then
  # separate counter passes (never combined with the trace domains gpurun refuses)
  # SQ: 8 slots per pass; GRBM independent.  WAIT_ANY + WAIT_INST_ANY + ACTIVE_INST_ANY ~ WAVE_CYCLES (quad-cycles)
  rocprofv3 --pmc SQ_WAVES SQ_WAVE_CYCLES SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_ANY SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_BUSY_CYCLES GRBM_GUI_ACTIVE --kernel-trace --output-format csv -d /tmp/p1_$TAG -- python3 $R/bench.py --workload $WL --steps 1 --warmup 0 --no-cpu-baseline --no-extras > /dev/null 2> $O/pmc1.err
  python3 $R/profiles/aggregate_pmc.py sq:/tmp/p1_$TAG > $O/pmc_valu_waves.csv || true
  head -30 $O/pmc_valu_waves.csv
fi
