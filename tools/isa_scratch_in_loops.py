#!/usr/bin/env python3
"""Spills inside loops: isa_scratch_in_loops.py file.s [kernel-substring] -> per kernel the scratch instructions in total and inside every
loop of more than 300 instructions (the interior bodies of the streaming kernels), from a hipcc -save-temps .s file."""
import re
import subprocess
import sys

path = sys.argv[1]
want = sys.argv[2] if len(sys.argv) > 2 else 'kernel'
cur, kern = None, {}
for l in open(path).read().split('\n'):
    m = re.match(r'^(_Z\w+):', l)
    if m:
        cur = m.group(1)
        kern[cur] = []
    elif cur:
        kern[cur].append(l)
for name, body in kern.items():
    if want not in name:
        continue
    labels, ins = {}, []
    for l in body:
        t = l.split(';')[0].strip()
        if not t:
            continue
        m = re.match(r'^(\.LBB\d+_\d+):', t)
        if m:
            labels[m.group(1)] = len(ins)
            continue
        if t.startswith('.'):
            continue
        ins.append(t)
    loops = []
    for i, t in enumerate(ins):
        m = re.match(r's_c?branch\w* (\.LBB\S+)', t)
        if m and m.group(1) in labels and labels[m.group(1)] <= i:
            a = labels[m.group(1)]
            if i - a > 300:
                loops.append((i - a, sum(1 for x in ins[a:i] if x.startswith('scratch_'))))
    tot = sum(1 for x in ins if x.startswith('scratch_'))
    dem = subprocess.run(['c++filt', name], capture_output=True, text=True).stdout.strip().replace('cm::', '')
    dem = re.sub(r'\(.*', '', dem)
    print('%-150s scratch %3d | loops (instructions, scratch): %s' % (dem[:150], tot, loops))
