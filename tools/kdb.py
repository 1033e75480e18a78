"""Per-launch-shape kernel durations from a rocprofv3 results database (rocpd sqlite): python tools/kdb.py <db> [stream] [top]
Takes the last third of the trace (the warm epochs of tools/prof_grouped.py)."""
import collections
import re
import sqlite3
import sys

db = sqlite3.connect(sys.argv[1])
only = int(sys.argv[2]) if len(sys.argv) > 2 and sys.argv[2] != "all" else None
top = int(sys.argv[3]) if len(sys.argv) > 3 else 40
rows = list(db.cursor().execute("select name, start, end, grid_x, grid_y, grid_z, workgroup_x, stream_id from kernels order by start"))
sub = rows[2 * len(rows) // 3:]
agg = collections.defaultdict(lambda: [0, 0])
for r in sub:
    if only is not None and r[7] != only:
        continue
    nm = re.sub(r"\(anonymous namespace\)::", "", r[0]).replace("void ", "").replace("unsigned short", "bf16")
    nm = nm.split("(")[0][:58]
    agg[(nm, r[3] // max(1, r[6]), r[4], r[5], r[7])][0] += 1
    agg[(nm, r[3] // max(1, r[6]), r[4], r[5], r[7])][1] += r[2] - r[1]
tot = sum(v[1] for v in agg.values())
print("span %.3f ms, kernel time %.3f ms" % ((sub[-1][2] - sub[0][1]) / 1e6, tot / 1e6))
for k, v in sorted(agg.items(), key=lambda kv: -kv[1][1])[:top]:
    print("%4d x %7.1f us = %5.1f%%  grid %5d x%3d x%3d  s%d  %s" % (v[0], v[1] / 1e3 / v[0], 100 * v[1] / tot, k[1], k[2], k[3], k[4], k[0]))
