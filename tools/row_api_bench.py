"""Per-row protocol cost (Modem.demodulate / Modem.modulate, one row per call): python tools/row_api_bench.py [package dir]
The optional argument puts another checkout's color_modem_amd first on the path (before / after comparisons)."""
import sys, time, numpy
if len(sys.argv) > 1:
    sys.path.insert(0, sys.argv[1])
sys.path.insert(1, '.'); sys.path.insert(2, 'tests')
import stacks
import color_modem_amd
from color_modem_amd import testing
import os
if os.environ.get('CM_ROW_COPY'):      # device-resident history with one upload / download per call instead of pinned zero-copy rows
    from color_modem_amd import engine
    engine.RowSession.ZERO_COPY = False
print('package:', color_modem_amd.__path__[0])
for name, size in (('pal_d', (720, 576)), ('ntsc_comb_3d', (720, 480)), ('secam', (720, 576))):
    m = stacks.make(name, size)
    comp = testing.synthetic_composite(1, size[1], size[0])[0]
    for y in range(0, 8, 2): m.demodulate(0, y, comp[y])          # plan creation, session buffers
    best = 1e9
    for rep in range(3):
        t0 = time.time()
        for y in range(0, size[1], 2): m.demodulate(1 + rep, y, comp[y])
        for y in range(1, size[1], 2): m.demodulate(1 + rep, y, comp[y])
        best = min(best, time.time() - t0)
    print('%-14s demodulate: %.1f ms per frame, %.1f us per row' % (name, best * 1e3, best / size[1] * 1e6))
for name, size in (('pal_s', (720, 576)), ('secam_avg', (720, 576))):
    enc = stacks.make(name, size)
    rgb = testing.synthetic_rgb(1, size[1], size[0])[0]
    for y in range(0, 8, 2): enc.modulate(0, y, rgb[0, y], rgb[1, y], rgb[2, y])
    best = 1e9
    for rep in range(3):
        t0 = time.time()
        for y in range(0, size[1], 2): enc.modulate(1 + rep, y, rgb[0, y], rgb[1, y], rgb[2, y])
        best = min(best, time.time() - t0)
    print('%-14s modulate:   %.1f us per row' % (name, best / (size[1] // 2) * 1e6))
# the other families: wrapped PAL combs, Proto-SECAM, NIIR, D2-MAC (decoders and encoders, one row per call)
import am_stacks
from color_modem_amd import line
from color_modem_amd.color import mac
lc625 = line.LineConfig((720, 576))
others = [('simple3d_pald', stacks.make('simple3d_pald', (720, 576))), ('proto', am_stacks.STACKS['proto'](line.LineConfig((720, 576), line.LineStandard.FRENCH_819))),
          ('niir', am_stacks.STACKS['niir'](lc625)), ('niir_hue', am_stacks.STACKS['niir_hue'](lc625)), ('mac', mac.MacModem(lc625))]
for name, m in others:
    H = 576
    rgb = testing.synthetic_rgb(1, H, 720)[0]
    enc_rows = [numpy.asarray(m.modulate(0, y, rgb[0, y], rgb[1, y], rgb[2, y])) for y in range(0, 16, 2)]
    def per_row(fn):      # median over the rows of a field (a call that grows the plan's per-line tables rebuilds the plan: tens of ms, once)
        ts = []
        for y in range(0, H, 2):
            t0 = time.time(); fn(y); ts.append(time.time() - t0)
        return float(numpy.median(ts))
    t_mod = per_row(lambda y: m.modulate(1, y, rgb[0, y], rgb[1, y], rgb[2, y]))
    row = enc_rows[-1]
    for y in range(0, 8, 2): m.demodulate(0, y, row)
    t_dem = per_row(lambda y: m.demodulate(1, y, row))
    print('%-14s modulate %.1f us per row, demodulate %.1f us per row (medians)' % (name, t_mod * 1e6, t_dem * 1e6), flush=True)
# several rows per call (round 4: Modem.demodulate_rows / modulate_rows - one launch and one synchronisation per group)
print('rows per call | us per row (host wall, numpy in -> numpy out), a field of 288 rows fed in groups')
for name, size in (('pal_d', (720, 576)), ('ntsc_comb_3d', (720, 480)), ('secam', (720, 576)), ('simple3d_pald', (720, 576))):
    m = stacks.make(name, size)
    comp = testing.synthetic_composite(1, size[1], size[0])[0]
    field = comp[0::2]
    cells = []
    for group in (1, 4, 16, 72, len(field)):
        best = 1e9
        for rep in range(4):
            t0 = time.time()
            for i in range(0, len(field), group):
                if group == 1: m.demodulate(rep, 2 * i, field[i])
                else: m.demodulate_rows(rep, 2 * i, field[i:i + group])
            best = min(best, time.time() - t0)
        cells.append('%4d: %6.2f' % (group, best / len(field) * 1e6))
    print('%-14s demodulate  %s' % (name, '   '.join(cells)), flush=True)
