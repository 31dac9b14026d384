import csv, glob, collections, sys
f = glob.glob(sys.argv[1] + '/*/*counter_collection.csv')[0]
rows = list(csv.DictReader(open(f)))
agg = collections.defaultdict(lambda: collections.defaultdict(list))
for r in rows:
    agg[r['Kernel_Name'][:60]][r['Counter_Name']].append(float(r['Counter_Value']))
for k, d in agg.items():
    if 'gemm' not in k and len(sys.argv) < 3: continue
    print(k)
    for c, v in sorted(d.items()):
        print('   %-28s n=%d avg=%.4g' % (c, len(v), sum(v[1:]) / max(len(v) - 1, 1)))
    if 'GRBM_GUI_ACTIVE' in d and 'SQ_VALU_MFMA_BUSY_CYCLES' in d:
        g = sum(d['GRBM_GUI_ACTIVE'][1:]) / max(len(d['GRBM_GUI_ACTIVE']) - 1, 1) / 8
        m = sum(d['SQ_VALU_MFMA_BUSY_CYCLES'][1:]) / max(len(d['SQ_VALU_MFMA_BUSY_CYCLES']) - 1, 1) / 1024
        print('   cycles/XCD %.4g  mfma busy/SIMD %.4g  util %.3f' % (g, m, m / g))
