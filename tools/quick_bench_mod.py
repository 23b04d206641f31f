"""Encoder throughput: python tools/quick_bench_mod.py [stack] [frames]"""
import sys, torch
sys.path.insert(0, '.'); sys.path.insert(0, 'tests')
import stacks
from color_modem_amd import image
stack = sys.argv[1] if len(sys.argv) > 1 else 'pal_s'
F = int(sys.argv[2]) if len(sys.argv) > 2 else 1000
h = 480 if stack.startswith('ntsc') else 576
eng = image.ImageModem(stacks.make(stack, (720, h)))._engine()
x = torch.rand((F, 3, h, 720), dtype=torch.float32, device='cuda')
out = torch.empty((F, h, 720), dtype=torch.float32, device='cuda')
for _ in range(2): eng.modulate_frames(x, 0, out=out)
torch.cuda.synchronize()
ts = []
for _ in range(5):
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record(); eng.modulate_frames(x, 0, out=out); e1.record(); torch.cuda.synchronize(); ts.append(e0.elapsed_time(e1))
ms = sorted(ts)[2]
print('%-10s mod %d frames %.3f ms %.1f Gpx/s %.0f GB/s' % (stack, F, ms, F * 720 * h / ms / 1e6, 16 * F * 720 * h / ms / 1e6))
