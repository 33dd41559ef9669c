# N1 (dtt_svd / dtt_ort on the D_64 train): kernel statistics and an MFMA-busy counter pass
set -e
R=$GRAFT_REPO_ROOT; cd /tmp && export TMPDIR=/tmp
O=$R/gpurun_out/r03_n1; mkdir -p $O
for w in svd_d64 ort_d64; do
  CMD="python3 $R/bench.py --workload $w --steps 5 --warmup 2 --no-cpu-baseline"
  rm -rf /tmp/n1s_$w /tmp/n1p_$w
  rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/n1s_$w -- $CMD > $O/bench_under_rocprof_$w.json 2> $O/stats_$w.err
  cp $(find /tmp/n1s_$w -name "*kernel_stats.csv" | head -1) $O/kernel_stats_$w.csv
  rocprofv3 --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_VALU_MFMA_F64 SQ_BUSY_CYCLES GRBM_GUI_ACTIVE --kernel-trace --output-format csv -d /tmp/n1p_$w -- $CMD > /dev/null 2> $O/pmc_$w.err || echo "pmc pass failed for $w"
  python3 $R/profiles/aggregate_pmc.py mfma:/tmp/n1p_$w > $O/pmc_mfma_$w.csv || true
  head -12 $O/kernel_stats_$w.csv | cut -c1-150
  head -12 $O/pmc_mfma_$w.csv
done
