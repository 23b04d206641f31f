"""One frame of the wide shapes on the scan decoder (chunks of 24 / 32 samples): us around one call.  CM_LIB selects the build.  python tools/scan_wide_bench.py"""
import sys, torch
sys.path.insert(0, '.'); sys.path.insert(0, 'tests')
import stacks
from color_modem_amd import image, testing
for stack, size in (('pal_d', (960, 576)), ('pal_d', (1280, 576)), ('ntsc_comb', (1440, 480)), ('pal_d', (1920, 576)), ('pal_3d', (1920, 576))):
    eng = image.ImageModem(stacks.make(stack, size))._engine()
    eng.set_small_batch('scan')
    res = []
    for F in (1, 4):
        comp = torch.from_numpy(testing.synthetic_composite(F, size[1], size[0], seed=3)).cuda()
        out = torch.empty((F, 3, size[1], size[0]), dtype=torch.float32, device='cuda')
        for _ in range(3): eng.demodulate_frames(comp, 0, out=out)
        torch.cuda.synchronize()
        ts = []
        for _ in range(9):
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record(); eng.demodulate_frames(comp, 0, out=out); e1.record(); torch.cuda.synchronize(); ts.append(e0.elapsed_time(e1))
        res.append('%d frame(s) %.0f us' % (F, 1e3 * sorted(ts)[4]))
    print('%-10s %dx%d: %s' % (stack, size[0], size[1], ', '.join(res)), flush=True)
