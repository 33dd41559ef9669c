# round-3 measurements on the GPU box (run through gpurun):
#   bash profiles/measure_r03.sh <workload> <arith> [groups]     -> gpurun_out/r03_<workload>_<arith>/
# kernel statistics (rocprofv3 --kernel-trace --stats), then counters in SEPARATE passes (never combined with the trace domains
# gpurun refuses): FETCH_SIZE, WRITE_SIZE (TCC: they do not fit one pass), SQ wave / VALU / wait counters.
#   bash profiles/measure_r03.sh calib                           -> FETCH_SIZE of the K2 streaming kernel on a known byte count
set -e
R=$GRAFT_REPO_ROOT; WL=$1; AR=${2:-exact}
cd /tmp && export TMPDIR=/tmp
if [ "$WL" = "calib" ]; then
  O=$R/gpurun_out/r03_calib; mkdir -p $O; rm -rf /tmp/cal_f /tmp/cal_w
  cat > /tmp/cal.py <<'P'
import sys, os
sys.path.insert(0, os.environ["GRAFT_REPO_ROOT"])
from ttcross_amd import engine as E
ms, by = E.k_residual_bench(1 << 22, 32, 5)
print("k_resid_argmax_stream rows 4194304 rank 32: algorithmic bytes per launch", by, "avg ms", ms)
P
  rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d /tmp/cal_f -- python3 /tmp/cal.py > $O/calib.txt 2> $O/calib_f.err
  rocprofv3 --pmc WRITE_SIZE --kernel-trace --output-format csv -d /tmp/cal_w -- python3 /tmp/cal.py > /dev/null 2> $O/calib_w.err
  python3 $R/profiles/aggregate_pmc.py pmc_fetch:/tmp/cal_f pmc_write:/tmp/cal_w > $O/pmc_fetch_write_calib.csv
  cat $O/calib.txt; grep k_resid $O/pmc_fetch_write_calib.csv
  exit 0
fi
G=${3:-0}
O=$R/gpurun_out/r03_${WL}_${AR}; mkdir -p $O
CMD="python3 $R/bench.py --workload $WL --arith $AR --steps 1 --warmup 0 --no-cpu-baseline --no-extras"
if [ "$G" != "0" ]; then CMD="$CMD --groups $G"; fi
rm -rf /tmp/st_$WL$AR /tmp/pf_$WL$AR /tmp/pw_$WL$AR /tmp/ps_$WL$AR
rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/st_$WL$AR -- $CMD > $O/bench_under_rocprof.json 2> $O/rocprof_stats.err
cp $(find /tmp/st_$WL$AR -name "*kernel_stats.csv" | head -1) $O/kernel_stats.csv
rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d /tmp/pf_$WL$AR -- $CMD > /dev/null 2> $O/pmc_f.err
rocprofv3 --pmc WRITE_SIZE --kernel-trace --output-format csv -d /tmp/pw_$WL$AR -- $CMD > /dev/null 2> $O/pmc_w.err
python3 $R/profiles/aggregate_pmc.py pmc_fetch:/tmp/pf_$WL$AR pmc_write:/tmp/pw_$WL$AR > $O/pmc_fetch_write.csv
rocprofv3 --pmc SQ_WAVES SQ_WAVE_CYCLES SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_ANY SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_BUSY_CYCLES GRBM_GUI_ACTIVE --kernel-trace --output-format csv -d /tmp/ps_$WL$AR -- $CMD > /dev/null 2> $O/pmc_sq.err
python3 $R/profiles/aggregate_pmc.py sq:/tmp/ps_$WL$AR > $O/pmc_sq.csv
head -6 $O/kernel_stats.csv | cut -c1-160
head -8 $O/pmc_fetch_write.csv
