"""Where the comb wrappers around the NIIR modems (generic.py) lose their accuracy: python tests/nested_niir_probe.py [n_seeds]
TEST TOOL (uses oracle/).  Random pictures through Simple3DCombModem(NiirModem, avg=weighted_avg) and SimpleCombModem(
HueCorrectingNiirModem) against oracle/cm_oracle_generic.py; for the worst one: the plane, the row, the column and what the oracle's
components look like there."""
import sys, warnings
import numpy
sys.path.insert(0, '.'); sys.path.insert(0, 'tests')
warnings.filterwarnings('ignore')
from color_modem_amd import image, testing
from oracle import cm_oracle_generic as og
import stacks

N = int(sys.argv[1]) if len(sys.argv) > 1 else 40
for name, (w, h) in (('simple3d_niir', (640, 35)), ('simple3d_niir', (720, 38)), ('simple_niir_hue', (960, 6))):
    worst = (0.0, None)
    errs = []
    for seed in range(N):
        modem = stacks.make_nested(name, (w, h))
        im = image.ImageModem(modem)
        rgb = testing.synthetic_rgb(2, h, w, seed=1000 + seed)
        comp = og.modulate_frames(modem, rgb.astype(numpy.float64), 1723).astype(numpy.float32)
        got, want = im.demodulate_frames(comp, first_frame=1723), og.demodulate_frames(modem, comp, 1723)
        e = max(stacks.rel_err(a, b) for a, b in zip(got, want))
        errs.append(e)
        if e > worst[0]:
            worst = (e, (seed, got, want, comp))
    errs.sort()
    print('%-16s %4dx%-3d  %d pictures: median %.2e  90%% %.2e  max %.2e' % (name, w, h, N, errs[N // 2], errs[int(N * 0.9)], errs[-1]))
    seed, got, want, comp = worst[1]
    d = numpy.abs(got.astype(numpy.float64) - want)
    f, p, y, x = numpy.unravel_index(numpy.argmax(d), d.shape)
    print('   worst: seed %d frame %d plane %d row %d col %d  |diff| %.3e  max|ref| %.3f;  samples beyond 5e-6 x max|ref|: %d of %d, rows %s'
          % (seed, f, p, y, x, d[f, p, y, x], numpy.abs(want[f]).max(), int((d[f] > 5e-6 * numpy.abs(want[f]).max()).sum()), d[f].size,
             sorted(set(numpy.nonzero(d[f] > 5e-6 * numpy.abs(want[f]).max())[1].tolist()))[:12]))
    print('   the three planes at that sample: got %s  want %s' % (got[f, :, y, x], want[f, :, y, x]))
