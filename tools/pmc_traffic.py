"""Average HBM-side traffic per launch and kernel from two rocprofv3 --pmc passes (FETCH_SIZE, WRITE_SIZE), with the
gfx950 corrections of MI355X_MICROARCH.md (HBM section): both counters are in KiB, FETCH_SIZE tallies 128-B requests at
64 B (x2 for wide coalesced reads), WRITE_SIZE is exact for 16-B stores and float atomics.
usage: pmc_traffic.py <dir_fetch> <dir_write> [out.json]"""
import csv, glob, json, sys
from collections import defaultdict


def per_kernel(d, counter):
    f = sorted(glob.glob(d + '/**/*counter_collection.csv', recursive=True))[-1]
    acc = defaultdict(lambda: [0, 0.0])
    for r in csv.DictReader(open(f)):
        if r['Counter_Name'] != counter:
            continue
        k = r['Kernel_Name'].replace('(anonymous namespace)::', '')
        acc[k][0] += 1
        acc[k][1] += float(r['Counter_Value'])
    return {k: (n, s / n) for k, (n, s) in acc.items()}


fe = per_kernel(sys.argv[1], 'FETCH_SIZE')
wr = per_kernel(sys.argv[2], 'WRITE_SIZE')
out = {}
for k in sorted(fe, key=lambda k: -fe[k][0] * fe[k][1]):
    n, f = fe[k]
    w = wr.get(k, (0, 0.0))[1]
    out[k] = {"launches": n, "fetch_bytes_corrected": 2 * f * 1024, "write_bytes": w * 1024, "hbm_bytes": (2 * f + w) * 1024}
    print('%-90s n %5d  fetch %9.3f MB  write %9.3f MB' % (k[:90], n, 2 * f * 1024 / 1e6, w * 1024 / 1e6))
# whole training step: everything but the one-off kernels (uploads, torch fills, the stream probe), per gather_batch launch
steps = max([v["launches"] for k, v in out.items() if "gather_batch" in k] or [1])
skip = ("nchw_to_nhwc", "at::native", "probe_", "__amd_rocclr_copyBuffer")
tot = sum(v["launches"] * v["hbm_bytes"] for k, v in out.items() if not any(x in k for x in skip))
out["__per_step__"] = {"steps": steps, "hbm_bytes": tot / steps}
print('per training step (%d steps): %.1f MB' % (steps, tot / steps / 1e6))
if len(sys.argv) > 3:
    json.dump(out, open(sys.argv[3], 'w'), indent=1)
