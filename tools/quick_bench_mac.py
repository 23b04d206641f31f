"""MAC path throughput: python tools/quick_bench_mac.py [frames]  (algorithmic bytes: 12960 per row either way)"""
import sys, torch
sys.path.insert(0, '.')
from color_modem_amd import comb, image, line
from color_modem_amd.color import mac
F = int(sys.argv[1]) if len(sys.argv) > 1 else 1000
H = 576
for name, avg in (('MacModem', False), ('ColorAveragingModem(MacModem)', True)):
    m = mac.MacModem(line.LineConfig((720, H)))
    eng = image.ImageModem(comb.ColorAveragingModem(m) if avg else m)._engine()
    rgb = torch.rand((F, 3, H, 720), dtype=torch.float32, device='cuda')
    comp = torch.empty((F, H, 1080), dtype=torch.float32, device='cuda')
    back = torch.empty((F, 3, H, 720), dtype=torch.float32, device='cuda')
    for label, fn in (('modulate', lambda: eng.modulate_frames(rgb, 0, out=comp)), ('demodulate', lambda: eng.demodulate_frames(comp, 0, out=back))):
        if avg and label == 'demodulate':
            continue
        for _ in range(2): fn()
        torch.cuda.synchronize()
        ts = []
        for _ in range(5):
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record(); fn(); e1.record(); torch.cuda.synchronize(); ts.append(e0.elapsed_time(e1))
        ms = sorted(ts)[2]
        print('%-30s %-10s %d frames %.3f ms  %.1f Gpx/s  %.0f GB/s algorithmic (12960 B/row)' % (name, label, F, ms, F * H * 720 / ms / 1e6, F * H * 12960 / ms / 1e6))
