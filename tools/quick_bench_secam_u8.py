"""SECAM decoder timing on byte images: python tools/quick_bench_secam_u8.py [frames]"""
import sys, torch
sys.path.insert(0, '.'); sys.path.insert(0, 'tests')
import stacks
from color_modem_amd import image, testing
F = int(sys.argv[1]) if len(sys.argv) > 1 else 1000
modem = stacks.make('secam', (720, 576))
eng = image.ImageModem(modem)._engine()
rgb = torch.from_numpy(testing.synthetic_rgb(4, 576, 720, seed=3)).cuda().repeat(F // 4, 1, 1, 1).contiguous()
rgb8 = (rgb.permute(0, 2, 3, 1) * 255).round().clamp(0, 255).to(torch.uint8).contiguous()
comp8 = eng.modulate_frames_u8(rgb8, 0)
out = torch.empty((F, 576, 720, 3), dtype=torch.uint8, device='cuda')
for _ in range(2): eng.demodulate_frames_u8(comp8, 0, out=out)
torch.cuda.synchronize()
ts = []
for _ in range(5):
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record(); eng.demodulate_frames_u8(comp8, 0, out=out); e1.record(); torch.cuda.synchronize(); ts.append(e0.elapsed_time(e1))
ms = sorted(ts)[2]
print('secam demod u8 frames', F, 'ms %.3f' % ms, 'Gpx/s %.1f' % (F * 576 * 720 / ms / 1e6))
