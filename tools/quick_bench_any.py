"""Decoder throughput at any image size: python tools/quick_bench_any.py STACK WIDTH HEIGHT [frames]"""
import sys, torch
sys.path.insert(0, '.'); sys.path.insert(0, 'tests')
import stacks
from color_modem_amd import image, testing
stack, w, h = sys.argv[1], int(sys.argv[2]), int(sys.argv[3])
F = int(sys.argv[4]) if len(sys.argv) > 4 else 400
eng = image.ImageModem(stacks.make(stack, (w, h), explicit=False))._engine()
comp = torch.from_numpy(testing.synthetic_composite(4, h, w)).cuda().repeat(F // 4, 1, 1).contiguous()
out = torch.empty((F, 3, h, w), dtype=torch.float32, device='cuda')
for _ in range(2): eng.demodulate_frames(comp, 0, out=out)
torch.cuda.synchronize()
ts = []
for _ in range(5):
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record(); eng.demodulate_frames(comp, 0, out=out); e1.record(); torch.cuda.synchronize(); ts.append(e0.elapsed_time(e1))
ms = sorted(ts)[2]
print('%-14s %4dx%-4d %4d frames  %.3f ms  %.1f Gpx/s   %s' % (stack, w, h, F, ms, F * w * h / ms / 1e6, eng.describe()))
