# second counter set for tools/pmc_gemm.sh (instruction mix)
cd /tmp && export TMPDIR=/tmp
for n in "$@"; do
  d=$GRAFT_REPO_ROOT/gpurun_out/pmc_gemm2/$(echo "$n" | tr ' .' '__')
  mkdir -p "$d"
  ONLY="$n" timeout -s KILL 90 rocprofv3 --pmc SQ_WAVES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_SMEM SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_WAIT_INST_LDS SQ_INSTS_MFMA \
      -d "$d" -o p --output-format csv -- python3 $GRAFT_REPO_ROOT/tools/bench_gemm.py bf16 > "$d/run.log" 2>&1 || { echo "pass failed for $n"; tail -5 "$d/run.log"; exit 1; }
  python3 - "$d" <<'PY'
import csv, glob, sys, collections
acc = collections.defaultdict(list)
for f in glob.glob(sys.argv[1] + '/**/*counter_collection.csv', recursive=True):
    for r in csv.DictReader(open(f)):
        if 'igemm' in r['Kernel_Name']:
            acc[r['Counter_Name']].append(float(r['Counter_Value']))
w = sum(acc['SQ_WAVES']) / len(acc['SQ_WAVES'])
print(sys.argv[1].split('/')[-1], 'waves %.0f; per wave:' % w, {k: round(sum(v) / len(v) / w, 1) for k, v in acc.items() if k != 'SQ_WAVES'})
PY
done
