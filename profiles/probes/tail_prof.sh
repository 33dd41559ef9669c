R=$GRAFT_REPO_ROOT; cd /tmp && export TMPDIR=/tmp
for v in 1 0; do
rm -rf /tmp/st_tl
TTX_SWEEP_TAIL=$v rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/st_tl -- python3 $R/bench.py --workload c16 --steps 5 --warmup 2 --no-cpu-baseline --no-extras > /dev/null 2> /dev/null
echo "== TTX_SWEEP_TAIL=$v"
python3 - <<'P'
import csv,glob
f=glob.glob('/tmp/st_tl/**/*kernel_stats.csv',recursive=True)[0]
for i,r in enumerate(csv.reader(open(f))):
    if i<9: print(r[0][:60].ljust(60), r[1], r[2], r[3][:9], r[5], r[6])
P
done
