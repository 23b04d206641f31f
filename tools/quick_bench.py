import sys, time, numpy, torch
sys.path.insert(0, '.'); sys.path.insert(0, 'tests')
import stacks
import os
from color_modem_amd import image, testing, _native
if os.environ.get('CM_LIB'): _native.LIB_PATH = os.environ['CM_LIB']
F = int(sys.argv[1]) if len(sys.argv) > 1 else 200
modem = stacks.make('pal_d', (720, 576))
im = image.ImageModem(modem)
comp = torch.from_numpy(testing.synthetic_composite(4, 576, 720)).cuda().repeat(F // 4, 1, 1).contiguous()
if os.environ.get('CM_ZERO'): comp.zero_()   # power check: all-zero operands switch fewer bits
out = torch.empty((F, 3, 576, 720), dtype=torch.float32, device='cuda')
eng = im._engine()
print(eng.describe())
for _ in range(2):
    eng.demodulate_frames(comp, 0, out=out)
torch.cuda.synchronize()
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
ts = []
for _ in range(5):
    e0.record(); eng.demodulate_frames(comp, 0, out=out); e1.record(); torch.cuda.synchronize()
    ts.append(e0.elapsed_time(e1))
ms = sorted(ts)[len(ts) // 2]
px = F * 576 * 720
print('frames', F, 'ms', ms, 'Mpx/s', px / ms / 1e3, 'GB/s(16B/px)', 16 * px / ms / 1e6, 'all', ts)
