# HBM traffic counters (separate passes, as MI355X_MICROARCH.md prescribes) of a bench workload:
#   bash profiles/measure_traffic.sh <tag> <workload>        (run through gpurun)
set -e
R=$GRAFT_REPO_ROOT; TAG=$1; WL=$2
O=$R/gpurun_out/$TAG
mkdir -p $O
cd /tmp && export TMPDIR=/tmp
rm -rf /tmp/pf_$TAG /tmp/pw_$TAG
rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d /tmp/pf_$TAG -- python3 $R/bench.py --workload $WL --steps 1 --warmup 0 --no-cpu-baseline --no-extras > /dev/null 2> $O/pmc_f.err
rocprofv3 --pmc WRITE_SIZE --kernel-trace --output-format csv -d /tmp/pw_$TAG -- python3 $R/bench.py --workload $WL --steps 1 --warmup 0 --no-cpu-baseline --no-extras > /dev/null 2> $O/pmc_w.err
python3 $R/profiles/aggregate_pmc.py pmc_fetch:/tmp/pf_$TAG pmc_write:/tmp/pw_$TAG > $O/pmc_fetch_write.csv
head -12 $O/pmc_fetch_write.csv
