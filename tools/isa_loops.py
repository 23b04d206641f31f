#!/usr/bin/env python3
"""Instruction mix of the loops of one kernel in a hipcc -save-temps .s file.

usage: isa_loops.py file.s kernel-name-substring [min-instructions]
Prints, for every backward branch target (loop) of the kernel, the number of instructions between the label and the
branch by class (VALU / packed / SALU / LDS / VMEM / SMEM / waitcnt / branch).
"""
import collections
import re
import sys


def classify(op):
    if op.startswith('v_pk_'):
        return 'vpk'
    if op.startswith('v_'):
        return 'valu'
    if op.startswith('ds_'):
        return 'lds'
    if op.startswith(('global_', 'buffer_', 'flat_', 'scratch_')):
        return 'vmem'
    if op.startswith('s_load') or op.startswith('s_buffer_load') or op.startswith('s_memtime'):
        return 'smem'
    if op.startswith('s_waitcnt'):
        return 'wait'
    if op.startswith('s_cbranch') or op.startswith('s_branch'):
        return 'branch'
    if op.startswith('s_'):
        return 'salu'
    return 'other'


def main():
    path, name = sys.argv[1], sys.argv[2]
    min_ins = int(sys.argv[3]) if len(sys.argv) > 3 else 50
    lines = open(path).read().split('\n')
    start = None
    for i, l in enumerate(lines):
        if l.startswith('_Z') and name in l.split(':')[0] and ':' in l:
            start = i
            break
    assert start is not None, 'kernel not found'
    end = start
    while not lines[end].strip().startswith('s_endpgm'):
        end += 1
    # there may be several s_endpgm; take the last before .section/.end
    j = end
    while j < len(lines) and not lines[j].startswith('\t.section') and not lines[j].startswith('.Lfunc_end'):
        if lines[j].strip().startswith('s_endpgm'):
            end = j
        j += 1
    body = lines[start:end + 1]
    labels = {}
    ins = []
    for l in body:
        s = l.strip()
        if not s or s.startswith(';') or s.startswith('.') and not s.endswith(':'):
            continue
        if s.endswith(':'):
            labels[s[:-1]] = len(ins)
            continue
        op = s.split()[0]
        ins.append((op, s))
    print('kernel instructions:', len(ins))
    for idx, (op, s) in enumerate(ins):
        if op.startswith('s_cbranch') or op == 's_branch':
            tgt = s.split()[-1]
            if tgt in labels and labels[tgt] <= idx and idx - labels[tgt] >= min_ins:
                seg = ins[labels[tgt]:idx + 1]
                c = collections.Counter(classify(o) for o, _ in seg)
                ops = collections.Counter(o for o, _ in seg)
                print('loop %s: %d instructions  %s' % (tgt, len(seg), dict(c)))
                print('   top ops:', ops.most_common(14))


main()
