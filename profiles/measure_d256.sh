# rocprofv3 kernel statistics of the full-size D_256 workload (BASELINE config 5) -- run on the GPU box via gpurun
set -e
R=$GRAFT_REPO_ROOT; TAG=${1:-d256}; WL=${2:-d256}
O=$R/gpurun_out/$TAG
mkdir -p $O
cd /tmp && export TMPDIR=/tmp
rm -rf /tmp/st_$TAG
rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/st_$TAG -- python3 $R/bench.py --workload $WL --steps 1 --warmup 0 --no-cpu-baseline --no-extras > $O/bench_under_rocprof.json 2> $O/rocprof_stats.err
cp $(find /tmp/st_$TAG -name "*kernel_stats.csv" | head -1) $O/kernel_stats.csv
head -12 $O/kernel_stats.csv
