import csv, glob, collections, sys
tot = collections.defaultdict(list)
for p in sorted(glob.glob('gpurun_out/pmc/p*/**/*counter_collection.csv', recursive=True)):
    for r in csv.DictReader(open(p)):
        if 'demod_' not in r['Kernel_Name']: continue
        tot[r['Counter_Name']].append(float(r['Counter_Value']))
m = {c: sum(v) / len(v) for c, v in tot.items()}
for c in sorted(m): print('   %-22s %.4g  (n=%d)' % (c, m[c], len(tot[c])))
wc = m['SQ_WAVE_CYCLES']
ms = float(sys.argv[1]) if len(sys.argv) > 1 else 3.64
px = 1000 * 576 * 720
print('waves', m['SQ_WAVES'], 'VALU/wave', m['SQ_INSTS_VALU'] / m['SQ_WAVES'], 'VALU per px', m['SQ_INSTS_VALU'] * 64 / px / (64/63.0)  )
print('fractions of WAVE_CYCLES: active_any %.3f wait_any %.3f wait_inst_any %.3f ; active_valu %.3f' % (m['SQ_ACTIVE_INST_ANY'] / wc, m['SQ_WAIT_ANY'] / wc, m['SQ_WAIT_INST_ANY'] / wc, m['SQ_ACTIVE_INST_VALU'] / wc))
print('FETCH raw GB %.3f (x2 = %.3f)  WRITE GB %.3f ; algorithmic read %.3f write %.3f' % (m['FETCH_SIZE'] * 1024 / 1e9, m['FETCH_SIZE'] * 2048 / 1e9, m['WRITE_SIZE'] * 1024 / 1e9, px * 4 / 1e9, px * 12 / 1e9))
print('clock GHz', m['GRBM_GUI_ACTIVE'] / 8 / (ms * 1e-3) / 1e9, ' L2 hit', m['TCC_HIT_sum'] / (m['TCC_HIT_sum'] + m['TCC_MISS_sum']))
print('VALU issue rate: %.3g wave-instr/s (ubench peak at 2 waves/SIMD 0.825e12)' % (m['SQ_INSTS_VALU'] / (ms * 1e-3)))
