set -e
R=$GRAFT_REPO_ROOT; rm -rf /tmp/st_* /tmp/pf_* /tmp/pw_*
O=$R/gpurun_out/m
mkdir -p $O
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/st_$$ -- python3 $R/bench.py --steps 5 --warmup 2 --no-cpu-baseline --no-extras > $O/bench_under_rocprof.json 2> $O/rocprof_stats.err
cp $(find /tmp/st_$$ -name "*kernel_stats.csv" | head -1) $O/kernel_stats.csv
rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d /tmp/pf_$$ -- python3 $R/bench.py --steps 2 --warmup 1 --no-cpu-baseline --no-extras > /dev/null 2> $O/pmc_f.err
rocprofv3 --pmc WRITE_SIZE --kernel-trace --output-format csv -d /tmp/pw_$$ -- python3 $R/bench.py --steps 2 --warmup 1 --no-cpu-baseline --no-extras > /dev/null 2> $O/pmc_w.err
python3 $R/profiles/aggregate_pmc.py pmc_fetch:/tmp/pf_$$ pmc_write:/tmp/pw_$$ > $O/pmc_fetch_write.csv
head -5 $O/kernel_stats.csv; grep k_sweep_cluster $O/pmc_fetch_write.csv
