"""Decoder throughput across image widths, one table per stack: python tools/quick_bench_widths.py [STACK ...] [--widths=720,1024,...] [--u8] [--mpix=295]

Every row is the same stack at another image width (= sampling rate = filter-set shape); the 720-wide row is the tuned instance of
rounds 1 - 5, the others name the instance that served them (`describe()`): a tuned shape of cm_shapes_wide.h or the run-time shape.
Frames are scaled so that every row moves about the same number of pixels (--mpix, default 295 M).  A launch of 295 Mpixel is only 2 - 4
fills of the device with workgroups (64 scan lines each, 4 - 6 workgroups per CU), so the last, partly empty fill weighs differently at every
width (2376 workgroups on 1024 slots at 1920 samples per line: 2.3 fills take the time of 3); --mpix=1200 makes that a few per cent."""
import sys
import torch
sys.path.insert(0, '.'); sys.path.insert(0, 'tests')
import stacks
from color_modem_amd import image, testing

args = [a for a in sys.argv[1:] if not a.startswith('--')]
widths = (720, 800, 960, 1024, 1280, 1440, 1600, 1920)
for a in sys.argv[1:]:
    if a.startswith('--widths'):
        widths = tuple(int(v) for v in a.split('=')[1].split(','))
u8 = '--u8' in sys.argv
mpix = 295.0
for a in sys.argv[1:]:
    if a.startswith('--mpix'):
        mpix = float(a.split('=')[1])
names = args or ['pal_d', 'pal_3d', 'pal_s', 'simple3d_pald', 'simple3d_pal3d', 'ntsc_comb', 'ntsc_comb_3d', 'ntsc', 'secam']
for name in names:
    h = 480 if name.startswith('ntsc') else 576
    base = None
    for w in widths:
        F = max(8, int(mpix * 1e6 / (w * h)) // 4 * 4)
        try:
            eng = image.ImageModem(stacks.make(name, (w, h)))._engine()
            if u8:
                comp = (torch.rand((F, h, w), device='cuda') * 160 + 40).to(torch.uint8)
                out = torch.empty((F, h, w, 3), dtype=torch.uint8, device='cuda')
                run = lambda: eng.demodulate_frames_u8(comp, 0, out=out)
            else:
                comp = torch.from_numpy(testing.synthetic_composite(4, h, w)).cuda().repeat(F // 4, 1, 1).contiguous()
                out = torch.empty((F, 3, h, w), dtype=torch.float32, device='cuda')
                run = lambda: eng.demodulate_frames(comp, 0, out=out)
            for _ in range(2):
                run()
            torch.cuda.synchronize()
            ts = []
            for _ in range(5):
                e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                e0.record(); run(); e1.record(); torch.cuda.synchronize(); ts.append(e0.elapsed_time(e1))
            ms = sorted(ts)[2]
            gpx = F * w * h / ms / 1e6
            if base is None:
                base = gpx
            print('%-16s %4dx%-4d %5d frames %8.3f ms %7.1f Gpx/s  %5.2f of the first row   %s' % (name + (' u8' if u8 else ''), w, h, F, ms, gpx, gpx / base, eng.describe()), flush=True)
            del comp, out
        except Exception as e:       # a stack the width does not serve (SECAM below 640): say so and go on
            print('%-16s %4dx%-4d  %s: %s' % (name, w, h, type(e).__name__, str(e)[:120]), flush=True)
    print(flush=True)
