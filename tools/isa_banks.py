#!/usr/bin/env python3
"""VGPR bank conflicts of the vector instructions of one kernel's loops (hipcc -save-temps .s file).

usage: isa_banks.py file.s mangled-name-substring [min-instructions]
For every loop (backward branch) of the kernel: scalar VALU instructions with three VGPR sources, how many of them read two
different registers of the same bank (index mod 4: 3.03 instead of 2.08 cycles at three waves per SIMD,
profiles/r02_ubench_bank.txt), and the packed instructions (whose issue interval does not depend on the banks)."""
import re
import sys


def vregs(operand):
    m = re.fullmatch(r'v(\d+)', operand)
    if m:
        return [int(m.group(1))]
    m = re.fullmatch(r'v\[(\d+):(\d+)\]', operand)
    if m:
        return list(range(int(m.group(1)), int(m.group(2)) + 1))
    return []


def main():
    path, name = sys.argv[1], sys.argv[2]
    min_ins = int(sys.argv[3]) if len(sys.argv) > 3 else 100
    lines = open(path).read().split('\n')
    start = next(i for i, l in enumerate(lines) if l.startswith('_Z') and name in l.split(':')[0] and ':' in l)
    end = next(i for i in range(start, len(lines)) if lines[i].strip().startswith('s_endpgm'))
    print(lines[start].split(':')[0][:150])
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
        m = re.match(r's_cbranch_\w+\s+(\S+)', t)
        if not m or m.group(1) not in labels or labels[m.group(1)] > i or i - labels[m.group(1)] < min_ins:
            continue
        body = ins[labels[m.group(1)]:i]
        three = conflict = two = conflict2 = pk = 0
        for b in body:
            op, _, rest = b.partition(' ')
            if not op.startswith('v_') or op.startswith('v_pk_'):
                pk += op.startswith('v_pk_')
                continue
            ops = [o.strip() for o in rest.split(',')]
            srcs = [r for o in ops[1:] for r in vregs(o.split(' ')[0])]
            if op.startswith(('v_fmac', 'v_mac')):
                srcs += vregs(ops[0])
            srcs = sorted(set(srcs))
            banks = [r % 4 for r in srcs]
            if len(srcs) >= 3:
                three += 1
                conflict += len(set(banks)) < len(banks)
            elif len(srcs) == 2:
                two += 1
                conflict2 += banks[0] == banks[1]
        print('loop of %4d instructions: 3-VGPR-source VALU %3d (same-bank pair in %3d), 2-source %3d (same bank %3d), packed %3d'
              % (len(body), three, conflict, two, conflict2, pk))


main()
