"""Print the kernel timeline of the last training step of a rocprofv3 --kernel-trace run (developer tool)."""
import csv, sys, glob
f = sorted(glob.glob(sys.argv[1] + '/**/*kernel_trace.csv', recursive=True))[-1]
rows = list(csv.DictReader(open(f)))
rows.sort(key=lambda r: int(r['Start_Timestamp']))
ad = [i for i, r in enumerate(rows) if 'adam_kernel' in r['Kernel_Name']]
a, b = ad[-2] + 1, ad[-1] + 1
seg = rows[a:b + 14]
t0 = int(seg[0]['Start_Timestamp']); prev = t0; busy = 0
for r in seg:
    s, e = int(r['Start_Timestamp']), int(r['End_Timestamp'])
    nm = r['Kernel_Name'].replace('(anonymous namespace)::', '').replace('unsigned short', 'bf16')[:70]
    g = '%sx%sx%s' % (r.get('Grid_Size_X', '?'), r.get('Grid_Size_Y', '?'), r.get('Grid_Size_Z', '?'))
    print('%8.1f  dur %7.1f  gap %6.1f  %-14s %s' % ((s - t0) / 1e3, (e - s) / 1e3, (s - prev) / 1e3, g, nm))
    busy += e - s; prev = e
print('span %.1f us busy %.1f us' % ((prev - t0) / 1e3, busy / 1e3))
