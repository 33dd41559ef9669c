"""Kernel timeline from `rocprofv3 --kernel-trace --output-format csv -d DIR -- python3 bench.py ...`.

usage: python3 profiles/trace_timeline.py DIR [START [COUNT]]
START: index into the list of k_sweep_cluster launches (negative counts from the end) at which to begin, or the word
`run` to begin at the k_init_samples launch of the 3rd complete run; COUNT: rows to print (default 40)."""
import csv
import glob
import sys

f = glob.glob(sys.argv[1] + '/**/*kernel_trace.csv', recursive=True)[0]
rows = list(csv.DictReader(open(f)))
rows.sort(key=lambda r: int(r['Start_Timestamp']))
start = sys.argv[2] if len(sys.argv) > 2 else "20"
count = int(sys.argv[3]) if len(sys.argv) > 3 else 40
if start == "run":
    inits = [i for i, r in enumerate(rows) if 'k_init_samples' in r['Kernel_Name']]
    i0 = inits[2] - 8
else:
    idx = [i for i, r in enumerate(rows) if 'k_sweep_cluster' in r['Kernel_Name']]
    i0 = idx[int(start)]
t0 = int(rows[i0]['Start_Timestamp'])
prev_end = t0
for r in rows[i0:i0 + count]:
    s, e = int(r['Start_Timestamp']), int(r['End_Timestamp'])
    print(f"{(s - t0) / 1e3:9.1f} us  gap {(s - prev_end) / 1e3:6.1f}  dur {(e - s) / 1e3:7.1f}  {r['Kernel_Name'][:60]}")
    prev_end = max(prev_end, e)
