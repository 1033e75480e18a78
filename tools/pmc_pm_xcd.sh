set -e
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/pmx; mkdir -p $O; cd /tmp; export TMPDIR=/tmp
FAVAE_PM_XCD=0 timeout -s KILL 120 rocprofv3 --pmc FETCH_SIZE -d $O/f0 -o f --output-format csv -- python3 $R/tools/prof_step.py bf16 512 1 > $O/f0.log 2>&1
FAVAE_PM_XCD=1 timeout -s KILL 120 rocprofv3 --pmc FETCH_SIZE -d $O/f1 -o f --output-format csv -- python3 $R/tools/prof_step.py bf16 512 1 > $O/f1.log 2>&1
cd $R
python - <<'PY'
import csv, collections, glob
for tag in ('f0','f1'):
    f = glob.glob('gpurun_out/pmx/%s/*counter_collection.csv' % tag)[0]
    d = collections.defaultdict(lambda: [0, 0.0])
    for r in csv.DictReader(open(f)):
        if r['Counter_Name'] != 'FETCH_SIZE': continue
        n = r['Kernel_Name']
        if 'igemm_pm' in n:
            k = n[n.index('igemm_pm'):][:40]
            d[k][0] += 1; d[k][1] += float(r['Counter_Value'])
    for k, (n, v) in sorted(d.items()):
        print(tag, k, n, '%.2f MB per launch (x2 KiB-corrected)' % (v * 1024 * 2 / n / 1e6))
PY
