"""Two-queue kernel timeline of the last full training step of a rocprofv3 --kernel-trace run (developer tool)."""
import csv, glob, sys
f = sorted(glob.glob(sys.argv[1] + '/**/*kernel_trace.csv', recursive=True))[-1]
rows = list(csv.DictReader(open(f)))
rows.sort(key=lambda r: int(r['Start_Timestamp']))
gb = [i for i, r in enumerate(rows) if 'gather_batch' in r['Kernel_Name']]
a, b = gb[-3], gb[-2]
t0 = int(rows[a]['Start_Timestamp'])
lo = float(sys.argv[2]) if len(sys.argv) > 2 else 0
for r in rows[a:b + 1]:
    s, e = int(r['Start_Timestamp']), int(r['End_Timestamp'])
    nm = r['Kernel_Name'].replace('(anonymous namespace)::', '').replace('unsigned short', 'bf16')[:60]
    if (s - t0) / 1e3 >= lo:
        print('%8.1f %7.1f q%s %s' % ((s - t0) / 1e3, (e - s) / 1e3, r.get('Queue_Id', '?'), nm))
