"""Summarise tools/pmc_gemm.sh passes: mean counter values per launch of the GEMM kernels (developer tool)."""
import csv, glob, os, sys, collections
root = sys.argv[1] if len(sys.argv) > 1 else 'gpurun_out/pmc_gemm'
for d in sorted(glob.glob(os.path.join(root, '*'))):
    files = glob.glob(os.path.join(d, '**', '*counter_collection.csv'), recursive=True)
    if not files:
        continue
    acc = collections.defaultdict(lambda: collections.defaultdict(list))
    for f in files:
        for r in csv.DictReader(open(f)):
            k = r['Kernel_Name']
            if 'igemm' not in k:
                continue
            acc[k[:60]][r['Counter_Name']].append(float(r['Counter_Value']))
    print('==', os.path.basename(d))
    for k, cs in acc.items():
        m = {c: sum(v) / len(v) for c, v in cs.items()}
        wc = m.get('SQ_WAVE_CYCLES', 0) or 1
        print('  %s  launches %d' % (k, len(next(iter(cs.values())))))
        print('    wave quad-cycles %.3g: parked %.0f%%  issue-stall %.0f%%  active %.0f%%' % (
            wc, 100 * m.get('SQ_WAIT_ANY', 0) / wc, 100 * m.get('SQ_WAIT_INST_ANY', 0) / wc, 100 * m.get('SQ_ACTIVE_INST_ANY', 0) / wc))
        print('    mfma busy cycles %.3g  busy cycles %.3g  lds idx active %.3g  bank conflict %.3g (%.0f%% of lds cycles)' % (
            m.get('SQ_VALU_MFMA_BUSY_CYCLES', 0), m.get('SQ_BUSY_CYCLES', 0), m.get('SQ_LDS_IDX_ACTIVE', 0), m.get('SQ_LDS_BANK_CONFLICT', 0),
            100 * m.get('SQ_LDS_BANK_CONFLICT', 0) / max(1, m.get('SQ_LDS_IDX_ACTIVE', 0))))
