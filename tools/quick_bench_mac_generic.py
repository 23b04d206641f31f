"""MAC resampling kernels and byte boundary throughput: python tools/quick_bench_mac_generic.py [frames]"""
import sys, torch
sys.path.insert(0, '.')
from color_modem_amd import image, line
from color_modem_amd.color import mac
F = int(sys.argv[1]) if len(sys.argv) > 1 else 200
H = 576
def t(fn):
    for _ in range(2): fn()
    torch.cuda.synchronize(); ts = []
    for _ in range(5):
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record(); fn(); e1.record(); torch.cuda.synchronize(); ts.append(e0.elapsed_time(e1))
    return sorted(ts)[2]
for w, cw in ((720, 720), (768, 1080), (1920, 1080)):
    eng = image.ImageModem(mac.MacModem(line.LineConfig((w, H)), cw))._engine()
    rgb = torch.rand((F, 3, H, w), device='cuda'); comp = torch.empty((F, H, cw), device='cuda'); back = torch.empty((F, 3, H, 720), device='cuda')
    m1 = t(lambda: eng.modulate_frames(rgb, 0, out=comp)); m2 = t(lambda: eng.demodulate_frames(comp, 0, out=back))
    print('float  rows %4d line %4d: modulate %.3f ms (%.1f Gpx/s)  demodulate %.3f ms (%.1f Gpx/s)  [%d frames]' % (w, cw, m1, F * H * w / m1 / 1e6, m2, F * H * 720 / m2 / 1e6, F))
eng = image.ImageModem(mac.MacModem(line.LineConfig((720, H))))._engine()
rgb8 = torch.randint(0, 256, (F, H, 720, 3), dtype=torch.uint8, device='cuda'); comp8 = torch.empty((F, H, 1080), dtype=torch.uint8, device='cuda'); back8 = torch.empty((F, H, 720, 3), dtype=torch.uint8, device='cuda')
m1 = t(lambda: eng.modulate_frames_u8(rgb8, 0, out=comp8)); m2 = t(lambda: eng.demodulate_frames_u8(comp8, 0, out=back8))
print('uint8  rows  720 line 1080: modulate %.3f ms (%.1f Gpx/s)  demodulate %.3f ms (%.1f Gpx/s)' % (m1, F * H * 720 / m1 / 1e6, m2, F * H * 720 / m2 / 1e6))
