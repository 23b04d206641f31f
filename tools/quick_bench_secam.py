"""SECAM decoder timing (BASELINE config 4, decode half): python tools/quick_bench_secam.py [frames] [variant] [width]"""
import sys, os, numpy, torch
sys.path.insert(0, '.'); sys.path.insert(0, 'tests')
from color_modem_amd import image, line, testing
from color_modem_amd.color import secam
F = int(sys.argv[1]) if len(sys.argv) > 1 else 1000
vn = sys.argv[2] if len(sys.argv) > 2 else 'SECAM'
W = int(sys.argv[3]) if len(sys.argv) > 3 else 720
modem = secam.SecamModem(line.LineConfig((W, 576)), getattr(secam.SecamVariant, vn))
im = image.ImageModem(modem); eng = im._engine()
rgb = torch.from_numpy(testing.synthetic_rgb(4, 576, W, seed=3)).cuda().repeat(F // 4, 1, 1, 1).contiguous()
comp = eng.modulate_frames(rgb, 0)
del rgb
out = torch.empty((F, 3, 576, W), dtype=torch.float32, device='cuda')
for _ in range(2): eng.demodulate_frames(comp, 0, out=out)
torch.cuda.synchronize()
ts = []
for _ in range(5):
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record(); eng.demodulate_frames(comp, 0, out=out); e1.record(); torch.cuda.synchronize(); ts.append(e0.elapsed_time(e1))
ms = sorted(ts)[2]
print('secam demod %-9s %4dx576 frames %d  ms %.3f  Gpx/s %.1f  %s' % (vn, W, F, ms, F * 576 * W / ms / 1e6, eng.describe().split(chr(10))[0][:90]))
