"""Round 3's wrong-result event (profiles/r03_scan_notes.txt item 4): demod_scan_kernel at chunks of 24 / 32 samples WITHOUT the 256-register cap
(-DCM_SCAN_AGPR=1: the compiler parks values in AGPRs).  Runs the scan decoder of the library CM_LIB points at against the streaming kernel,
with nothing / NaNs / a huge finite pattern left in every register and LDS byte before each launch (tests/poison.py).
  CM_LIB=build_ab/libscanagpr.so python tools/scan_agpr_probe.py        (TEST TOOL)"""
import sys, numpy, torch
sys.path.insert(0, '.'); sys.path.insert(0, 'tests')
import stacks, poison
from color_modem_amd import image, testing
for size in ((720, 64), (1280, 64), (1440, 32), (1920, 32)):
    eng = image.ImageModem(stacks.make('pal_d', size))._engine()
    comp = torch.from_numpy(testing.synthetic_composite(2, size[1], size[0], seed=5)).cuda()
    eng.set_small_batch('rows')
    want = eng.demodulate_frames(comp, first_frame=1).cpu().numpy()
    eng.set_small_batch('scan')
    res = []
    for pattern in (None, None, 0x7fc0babe, 0x7fc0babe, 0x7f7fffff, 0x00000000, None):
        if pattern is not None:
            poison.poison(pattern)
        got = eng.demodulate_frames(comp, first_frame=1).cpu().numpy()
        bad = ~numpy.isfinite(got)
        err = numpy.abs(numpy.where(bad, 0, got) - want).max() / numpy.abs(want).max()
        res.append('%s: %.1e%s' % ('none' if pattern is None else hex(pattern), err, ' +%d non-finite' % bad.sum() if bad.any() else ''))
    print('%dx%d  %s' % (size[0], size[1], ' | '.join(res)), flush=True)
