R=$GRAFT_REPO_ROOT; cd /tmp && export TMPDIR=/tmp
rm -rf /tmp/st_dc
rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/st_dc -- python3 $R/bench.py --workload d256 --arith exact --steps 1 --warmup 0 --no-cpu-baseline --no-extras > /dev/null 2> /dev/null
python3 - <<'P'
import csv,glob
f=glob.glob('/tmp/st_dc/**/*kernel_stats.csv',recursive=True)[0]
for i,r in enumerate(csv.reader(open(f))):
    if i<12: print(r[0][:70].ljust(70), r[1], r[2], r[3][:9])
P
