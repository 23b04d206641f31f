"""Small-batch latency of the Proto-SECAM / NIIR / MAC kernels: one frame, HIP events around one call."""
import sys, time, numpy, torch
sys.path.insert(0, '.'); sys.path.insert(0, 'tests')
import am_stacks
from color_modem_amd import image, line, testing
from color_modem_amd.color import mac
def timeit(fn):
    for _ in range(3): fn()
    torch.cuda.synchronize(); ts = []
    for _ in range(7):
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record(); fn(); e1.record(); torch.cuda.synchronize(); ts.append(e0.elapsed_time(e1))
    return 1e3 * sorted(ts)[3]
for stack, size, std in (('proto', (720, 736), 'FRENCH_819'), ('niir', (720, 576), 'GERBER_625'), ('niir_hue', (720, 576), 'GERBER_625')):
    lc = line.LineConfig(size, getattr(line.LineStandard, std))
    eng = image.ImageModem(am_stacks.STACKS[stack](lc))._engine()
    W, H = size
    for F in (1, 4, 16, 32, 48):
        rgb = torch.rand((F, 3, H, W), device='cuda'); comp = eng.modulate_frames(rgb, 0)
        for mode in ('rows', 'scan', 'auto'):
            eng.set_small_batch(mode)
            print('%-9s %2d frame(s) %dx%d %-5s: modulate %.0f us, demodulate %.0f us' % (stack, F, W, H, mode, timeit(lambda: eng.modulate_frames(rgb, 0)), timeit(lambda: eng.demodulate_frames(comp, 0))), flush=True)
# the NIIR decoder (hue path in float64) in auto mode over the hand-over point of the two kernels
lc = line.LineConfig((720, 576), line.LineStandard.GERBER_625)
eng = image.ImageModem(am_stacks.STACKS['niir'](lc))._engine()
for F in (1, 4, 16, 64, 256):
    comp = eng.modulate_frames(torch.rand((F, 3, 576, 720), device='cuda'), 0)
    t = timeit(lambda: eng.demodulate_frames(comp, 0))
    print('niir auto %3d frame(s) 720x576: demodulate %.0f us  (%.1f Gpixel/s)' % (F, t, F * 720 * 576 / t / 1e3), flush=True)
lc = line.LineConfig((720, 576))
eng = image.ImageModem(mac.MacModem(lc))._engine()
rgb = torch.rand((1, 3, 576, 720), device='cuda'); comp = eng.modulate_frames(rgb, 0)
print('mac       one frame 720x576: modulate %.0f us, demodulate %.0f us' % (timeit(lambda: eng.modulate_frames(rgb, 0)), timeit(lambda: eng.demodulate_frames(comp, 0))))
