#!/usr/bin/env python3
"""Float32 arithmetic of one kernel's loops, counted from the ISA (hipcc -save-temps .s file; tools/dev_build.sh leaves one).

usage: isa_fma_census.py file.s mangled-name-substring [min-instructions] [pixels-per-body]

For every loop (backward branch) of the kernel: vector float instructions by kind and the FMA-EQUIVALENTS they stand for -
one per scalar float instruction (fma / fmac / fmamk / fmaak / mul / add / sub / max / min), two per packed one (v_pk_*_f32) -
plus what is vector but not arithmetic (moves, conversions, integer / compare / select, cross-lane).  bench.py's
roofline_valu.fma_equivalents_per_pixel is the interior stage A + stage B figure this prints for the headline instance
(profiles/r04_headline_bound.txt)."""
import collections
import re
import sys

FLOAT1 = ('v_fma_f32', 'v_fmac_f32', 'v_fmamk_f32', 'v_fmaak_f32', 'v_mul_f32', 'v_add_f32', 'v_sub_f32', 'v_subrev_f32',
          'v_max_f32', 'v_min_f32', 'v_mac_f32', 'v_mad_f32')
FLOAT2 = ('v_pk_fma_f32', 'v_pk_mul_f32', 'v_pk_add_f32')
FLOAT64 = ('v_fma_f64', 'v_mul_f64', 'v_add_f64')


def kind(op):
    base = re.sub(r'_(e32|e64|dpp|sdwa)$', '', op)
    if base in FLOAT1:
        return 'f32'
    if base in FLOAT2:
        return 'pk'
    if base in FLOAT64:
        return 'f64'
    if base.startswith('v_mov') or base.startswith('v_accvgpr') or base.startswith('v_pk_mov'):
        return 'mov'
    if base.startswith('v_cvt') or base.startswith('v_rndne') or base.startswith('v_perm'):
        return 'cvt'
    if base.startswith(('v_readlane', 'v_readfirstlane', 'v_writelane', 'v_permlane', 'v_swap')):
        return 'lane'
    if base.startswith('v_'):
        return 'int'
    return None


def main():
    path, name = sys.argv[1], sys.argv[2]
    min_ins = int(sys.argv[3]) if len(sys.argv) > 3 else 100
    px = float(sys.argv[4]) if len(sys.argv) > 4 else 4.0
    lines = open(path).read().split('\n')
    start = next(i for i, l in enumerate(lines) if l.startswith('_Z') and name in l.split(':')[0] and ':' in l)
    end = next(i for i in range(start, len(lines)) if lines[i].strip().startswith('s_endpgm'))
    print(lines[start].split(':')[0][:170])
    labels, ins = {}, []
    for l in lines[start:end]:
        t = l.split(';')[0].strip()
        if not t or t.startswith('.') and not t.endswith(':'):
            continue
        if t.endswith(':'):
            labels[t[:-1]] = len(ins)
            continue
        ins.append(t)
    for i, t in enumerate(ins):
        m = re.match(r's_(?:cbranch_\w+|branch)\s+(\S+)', t)   # interior bodies close with an unconditional backward branch
        if not m or m.group(1) not in labels or labels[m.group(1)] > i or i - labels[m.group(1)] < min_ins:
            continue
        body = ins[labels[m.group(1)]:i]
        c = collections.Counter()
        ops = collections.Counter()
        for b in body:
            op = b.split(' ')[0]
            k = kind(op)
            if k:
                c[k] += 1
                ops[(k, re.sub(r'_(e32|e64)$', '', op))] += 1
        eq = c['f32'] + 2 * c['pk'] + c['f64']
        vec = sum(c.values())
        print('loop of %4d instructions: vector %3d = scalar float %3d + packed float %3d + f64 %d + mov %3d + cvt %2d + lane %2d + int/cmp/select %3d'
              '  ->  %3d FMA-equivalents per body = %.1f per pixel (%.1f vector instructions per pixel)'
              % (len(body), vec, c['f32'], c['pk'], c['f64'], c['mov'], c['cvt'], c['lane'], c['int'], eq, eq / px, vec / px))
        print('      ' + ', '.join('%s %d' % (o, n) for (k, o), n in sorted(ops.items(), key=lambda kv: -kv[1]) if k in ('f32', 'pk', 'mov')))


main()
