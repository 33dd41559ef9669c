import csv, sys, glob
f = glob.glob(sys.argv[1] + '/**/*kernel_trace.csv', recursive=True)[0]
rows = list(csv.DictReader(open(f)))
rows.sort(key=lambda r: int(r['Start_Timestamp']))
# last run: find last 16 k_sweep_cluster launches
idx = [i for i, r in enumerate(rows) if 'k_sweep_cluster' in r['Kernel_Name']]
i0 = idx[int(sys.argv[2])]
t0 = int(rows[i0]['Start_Timestamp'])
prev_end = t0
for r in rows[i0:i0 + 40]:
    s, e = int(r['Start_Timestamp']), int(r['End_Timestamp'])
    print(f"{(s - t0) / 1e3:9.1f} us  gap {(s - prev_end) / 1e3:6.1f}  dur {(e - s) / 1e3:7.1f}  {r['Kernel_Name'][:50]}")
    prev_end = e
