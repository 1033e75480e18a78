"""Summarise a rocprofv3 kernel_stats.csv (developer tool)."""
import csv, sys, glob
f = sorted(glob.glob(sys.argv[1] + '/**/*kernel_stats.csv', recursive=True))[-1]
rows = list(csv.DictReader(open(f)))
tot = sum(float(r['TotalDurationNs']) for r in rows)
n = int(sys.argv[2]) if len(sys.argv) > 2 else 25
for r in rows[:n]:
    print('%-86s calls %6s avg %9.1f us  %5.1f%%' % (r['Name'].replace('(anonymous namespace)::', '')[:86], r['Calls'], float(r['AverageNs']) / 1e3, 100 * float(r['TotalDurationNs']) / tot))
print('total kernel ms', tot / 1e6)
