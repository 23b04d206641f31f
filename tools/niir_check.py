"""NIIR decoder against the float64 oracle in one small-batch mode: python tools/niir_check.py [rows|scan|auto] [W] [H] [frames] (TEST TOOL: uses oracle/)"""
import sys, numpy
sys.path.insert(0, '.'); sys.path.insert(0, 'tests')
import am_stacks
from color_modem_amd import image, line, testing
from oracle import cm_oracle_am as oa
mode = sys.argv[1] if len(sys.argv) > 1 else 'rows'
W = int(sys.argv[2]) if len(sys.argv) > 2 else 720
H = int(sys.argv[3]) if len(sys.argv) > 3 else 40
F = int(sys.argv[4]) if len(sys.argv) > 4 else 2
for stack in ('niir', 'niir_hue'):
    lc = line.LineConfig((W, H), line.LineStandard.GERBER_625)
    modem = am_stacks.STACKS[stack](lc)
    worst = 0.0
    over = 0
    for seed in range(3):
        rgb = testing.synthetic_rgb(F, H, W, seed=1000 + 10 * seed)
        comp = oa.modulate_frames(modem, rgb.astype(numpy.float64), 3 + seed).astype(numpy.float32)
        want = oa.demodulate_frames(modem, comp.astype(numpy.float64), 3 + seed)
        eng = image.ImageModem(modem)._engine()
        eng.set_small_batch(mode)
        got = eng.demodulate_frames(comp, first_frame=3 + seed)
        err = numpy.abs(got - want) / numpy.abs(want).max()
        worst = max(worst, err.max())
        over += int((err > 1e-5).sum())
        ix = numpy.unravel_index(err.argmax(), err.shape)
    print('%-9s %s %dx%d: worst %.3e at %s, samples > 1e-5: %d' % (stack, mode, W, H, worst, ix, over))
