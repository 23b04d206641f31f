"""Summary of tools/pmc_any.sh: per-kernel mean duration and counters of the dispatches whose name contains KSUB."""
import collections, csv, glob, os, sys
out, ksub = sys.argv[1], sys.argv[2]
def rows(sub, pattern):
    for p in sorted(glob.glob(os.path.join(out, sub, '**', pattern), recursive=True)):
        with open(p) as fh:
            for r in csv.DictReader(fh):
                yield r
d = [(int(r['End_Timestamp']) - int(r['Start_Timestamp']), r) for r in rows('kt', '*kernel_trace.csv') if ksub in r['Kernel_Name']]
if d:
    ds = sorted(x for x, _ in d)
    r = d[-1][1]
    print('kernel %s' % r['Kernel_Name'][:100])
    print('  launches %d  median %.4f ms  min %.4f  grid %s x wg %s  VGPR %s AGPR %s SGPR %s LDS %s scratch %s'
          % (len(ds), ds[len(ds) // 2] / 1e6, ds[0] / 1e6, r['Grid_Size_X'], r['Workgroup_Size_X'], r['VGPR_Count'], r.get('Accum_VGPR_Count'), r['SGPR_Count'], r['LDS_Block_Size'], r['Scratch_Size']))
m = {}
for sub in ('s1', 's2', 'm1', 'm2'):
    per = collections.defaultdict(list)
    for r in rows(sub, '*counter_collection.csv'):
        if ksub in r['Kernel_Name']:
            per[r['Counter_Name']].append(float(r['Counter_Value']))
    for c, v in per.items():
        v = v[len(v) // 2:]            # the later dispatches (warm)
        m[c] = sum(v) / len(v)
for c in sorted(m):
    print('   %-24s %.6g' % (c, m[c]))
if 'SQ_WAVE_CYCLES' in m:
    wc = m['SQ_WAVE_CYCLES']
    g = lambda k: m.get(k, float('nan'))
    print('shares of SQ_WAVE_CYCLES: issuing %.3f, parked in s_waitcnt / barrier %.3f, issue stalls %.3f; VALU active %.3f; LDS issue stall %.3f'
          % (g('SQ_ACTIVE_INST_ANY') / wc, g('SQ_WAIT_ANY') / wc, g('SQ_WAIT_INST_ANY') / wc, g('SQ_ACTIVE_INST_VALU') / wc, g('SQ_WAIT_INST_LDS') / wc))
    print('wave instructions per wave: VALU %.0f  SALU %.0f  LDS %.0f  VMEM %.0f  SMEM %.0f   (waves %.0f)'
          % (tuple(g(k) / g('SQ_WAVES') for k in ('SQ_INSTS_VALU', 'SQ_INSTS_SALU', 'SQ_INSTS_LDS', 'SQ_INSTS_VMEM', 'SQ_INSTS_SMEM')) + (g('SQ_WAVES'),)))
if 'FETCH_SIZE' in m and 'WRITE_SIZE' in m:
    print('HBM traffic per launch: 2 x FETCH_SIZE %.3f GB + WRITE_SIZE %.3f GB' % (2 * m['FETCH_SIZE'] * 1024 / 1e9, m['WRITE_SIZE'] * 1024 / 1e9))
