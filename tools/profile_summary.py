# -*- coding: utf-8 -*-
"""Summary of tools/profile_bench.sh: per-kernel durations of the timed launches (warm-ups excluded), HBM traffic with the
FETCH_SIZE correction of MI355X_MICROARCH.md, SQ issue / wait shares and instructions per pixel; also writes traffic.json."""
import collections
import csv
import glob
import json
import os
import sys

out = sys.argv[1]
STEPS, PIXELS = 20, 1000 * 720 * 576
KERNEL = 'demod_'


def rows(sub, pattern):
    for p in sorted(glob.glob(os.path.join(out, sub, '**', pattern), recursive=True)):
        with open(p) as fh:
            for r in csv.DictReader(fh):
                yield r


# ---- kernel trace: the last STEPS launches of the demodulator are the timed ones
durs = []
for r in rows('kt', '*kernel_trace.csv'):
    if KERNEL in r['Kernel_Name']:
        durs.append((int(r['Start_Timestamp']), int(r['End_Timestamp']) - int(r['Start_Timestamp']), r['Kernel_Name']))
durs.sort()
timed = [d for _, d, _ in durs[-STEPS:]]
print('kernel: %s' % (durs[-1][2][:110] if durs else '?'))
if timed:
    mean = sum(timed) / len(timed) / 1e6
    print('rocprofv3 --kernel-trace: %d launches of it in the run, the last %d (the timed steps): mean %.4f ms, min %.4f, max %.4f'
          % (len(durs), len(timed), mean, min(timed) / 1e6, max(timed) / 1e6))
    print('  -> algorithmic 16 B/pixel: %.1f GB/s = %.4f of 8 TB/s; %.1f Gpixel/s' % (16 * PIXELS / mean / 1e6, 16 * PIXELS / mean / 1e6 / 8000, PIXELS / mean / 1e6))
for r in rows('kt', '*kernel_stats.csv'):
    print('  stats: %-90s calls %s avg %.4f ms  %s %%' % (r['Name'][:90], r['Calls'], float(r['AverageNs']) / 1e6, r['Percentage']))

if os.path.isdir(os.path.join(out, 'kt_all')):      # the driver's default command: headline + other_configs (BASELINE configs 3 / 4)
    print('kernel stats of the default run (python3 bench.py --gpus 1 --steps 20 --warmup 5: the headline and other_configs):')
    for r in rows('kt_all', '*kernel_stats.csv'):
        if float(r['Percentage']) >= 0.5:
            print('  stats: %-90s calls %s avg %.4f ms  %s %%' % (r['Name'][:90], r['Calls'], float(r['AverageNs']) / 1e6, r['Percentage']))

# ---- counters: mean over the last STEPS dispatches of the demodulator in each pass
m = {}
for sub in ('m1', 'm2', 's1', 's2'):
    per = collections.defaultdict(list)
    for r in rows(sub, '*counter_collection.csv'):
        if KERNEL in r['Kernel_Name']:
            per[r['Counter_Name']].append((int(r['Dispatch_Id']), float(r['Counter_Value'])))
    for c, v in per.items():
        v.sort()
        tail = [x for _, x in v[-STEPS:]]
        m[c] = sum(tail) / len(tail)
for c in sorted(m):
    print('   %-24s %.5g' % (c, m[c]))
if 'FETCH_SIZE' in m and 'WRITE_SIZE' in m:
    fetch, write = m['FETCH_SIZE'] * 1024, m['WRITE_SIZE'] * 1024
    hbm = 2 * fetch + write
    print('HBM traffic per launch: 2 x FETCH_SIZE %.3f GB + WRITE_SIZE %.3f GB = %.3f GB; algorithmic %.3f GB (x %.3f)'
          % (2 * fetch / 1e9, write / 1e9, hbm / 1e9, 16 * PIXELS / 1e9, hbm / (16.0 * PIXELS)))
    tj = {'workload': 'PAL-BG 2D comb demodulate, 720x576, 1000 PAL-encoded frames, 1 x MI355X: python3 bench.py --gpus 1 --steps 20 --warmup 5',
          'frames': 1000, 'source': 'tools/profile_bench.sh: rocprofv3 --pmc FETCH_SIZE GRBM_GUI_ACTIVE / --pmc WRITE_SIZE TCC_HIT_sum TCC_MISS_sum '
          '(separate runs), mean of the 20 timed launches', 'fetch_size_kb_raw': round(m['FETCH_SIZE']), 'write_size_kb': round(m['WRITE_SIZE']),
          'hbm_bytes_per_launch': int(round(hbm, -6)),
          'correction': 'hbm = (2 * FETCH_SIZE + WRITE_SIZE) * 1024 (MI355X_MICROARCH.md: FETCH_SIZE tallies 128-B requests at 64 B)',
          'algorithmic_bytes_per_launch': 16 * PIXELS}
    try:      # the kernel description bench.py compares with its own (config.kernel of the line of the same run)
        with open(os.path.join(out, 'bench_line.json')) as fh:
            tj['kernel'] = json.loads([ln for ln in fh if ln.startswith('{')][-1])['config']['kernel']
    except (OSError, ValueError, KeyError, IndexError):
        pass
    if 'TCC_HIT_sum' in m:
        tj['l2_hit_rate'] = round(m['TCC_HIT_sum'] / (m['TCC_HIT_sum'] + m['TCC_MISS_sum']), 3)
    # what the figure belongs to: bench.py refuses it for another build of the library (roofline.traffic_source)
    import datetime
    import hashlib
    import subprocess
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    lib = os.environ.get('CM_LIB') or os.path.join(root, 'color_modem_amd', 'libcolor_modem_hip.so')
    h = hashlib.sha256()
    with open(lib, 'rb') as fh:
        for block in iter(lambda: fh.read(1 << 20), b''):
            h.update(block)
    tj['lib_sha16'] = h.hexdigest()[:16]
    sys.path.insert(0, root)
    import bench
    tj['src_sha16'] = bench.sources_sha16()
    tj['date'] = datetime.date.today().isoformat()
    try:
        tj['head'] = subprocess.check_output(['git', '-C', root, 'rev-parse', '--short', 'HEAD'], stderr=subprocess.DEVNULL).decode().strip()
    except (OSError, subprocess.CalledProcessError):
        tj['head'] = os.environ.get('CM_HEAD', 'unknown (no .git on the GPU box)')
    with open(os.path.join(out, 'traffic.json'), 'w') as fh:
        json.dump(tj, fh, indent=1)
if 'SQ_WAVE_CYCLES' in m:
    wc = m['SQ_WAVE_CYCLES']
    g = lambda k: m.get(k, float('nan'))
    print('shares of SQ_WAVE_CYCLES: issuing (ACTIVE_INST_ANY) %.3f, parked in s_waitcnt / barrier (WAIT_ANY) %.3f, issue stalls (WAIT_INST_ANY) %.3f; '
          'VALU active %.3f' % (g('SQ_ACTIVE_INST_ANY') / wc, g('SQ_WAIT_ANY') / wc, g('SQ_WAIT_INST_ANY') / wc, g('SQ_ACTIVE_INST_VALU') / wc))
    print('instructions per pixel (wave instructions x 64 lanes / pixels): VALU %.1f  SALU %.1f  LDS %.2f  VMEM %.3f  SMEM %.2f'
          % tuple(g(k) * 64.0 / PIXELS for k in ('SQ_INSTS_VALU', 'SQ_INSTS_SALU', 'SQ_INSTS_LDS', 'SQ_INSTS_VMEM', 'SQ_INSTS_SMEM')))
