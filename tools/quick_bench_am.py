"""Throughput of the Proto-SECAM / NIIR kernels: python tools/quick_bench_am.py [frames]"""
import sys, torch
sys.path.insert(0, '.'); sys.path.insert(0, 'tests')
import am_stacks
from color_modem_amd import image, line
F = int(sys.argv[1]) if len(sys.argv) > 1 else 200
for stack, size, std in (('proto', (720, 736), 'FRENCH_819'), ('proto_avg', (720, 736), 'FRENCH_819'), ('niir', (720, 576), 'GERBER_625'),
                         ('niir_hue', (720, 576), 'GERBER_625')):
    lc = line.LineConfig(size, getattr(line.LineStandard, std))
    eng = image.ImageModem(am_stacks.STACKS[stack](lc))._engine()
    W, H = size
    rgb = torch.rand((F, 3, H, W), dtype=torch.float32, device='cuda')
    comp = torch.empty((F, H, W), dtype=torch.float32, device='cuda')
    out = torch.empty((F, 3, H, W), dtype=torch.float32, device='cuda')
    res = []
    rgb8 = torch.randint(0, 256, (F, H, W, 3), dtype=torch.uint8, device='cuda')
    comp8 = torch.empty((F, H, W), dtype=torch.uint8, device='cuda')
    out8 = torch.empty((F, H, W, 3), dtype=torch.uint8, device='cuda')
    cases = [('mod', lambda: eng.modulate_frames(rgb, 0, out=comp)), ('demod', lambda: eng.demodulate_frames(comp, 0, out=out))]
    if W % 16 == 0:
        cases += [('mod_u8', lambda: eng.modulate_frames_u8(rgb8, 0, out=comp8)), ('demod_u8', lambda: eng.demodulate_frames_u8(comp8, 0, out=out8))]
    for name, fn in cases:
        for _ in range(2): fn()
        torch.cuda.synchronize()
        ts = []
        for _ in range(5):
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record(); fn(); e1.record(); torch.cuda.synchronize(); ts.append(e0.elapsed_time(e1))
        ms = sorted(ts)[2]
        res.append('%s %.3f ms %.1f Gpx/s' % (name, ms, F * W * H / ms / 1e6))
    print('%-10s %dx%d x %d frames: %s' % (stack, W, H, F, ', '.join(res)))
