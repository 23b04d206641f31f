"""One AM kernel at a time (profiling): python tools/quick_bench_am1.py STACK mod|demod [frames] [reps]"""
import sys, torch
sys.path.insert(0, '.'); sys.path.insert(0, 'tests')
import am_stacks
from color_modem_amd import image, line
stack, what = sys.argv[1], sys.argv[2]
F = int(sys.argv[3]) if len(sys.argv) > 3 else 400
reps = int(sys.argv[4]) if len(sys.argv) > 4 else 5
size, std = ((720, 736), 'FRENCH_819') if stack.startswith('proto') else ((720, 576), 'GERBER_625')
lc = line.LineConfig(size, getattr(line.LineStandard, std))
eng = image.ImageModem(am_stacks.STACKS[stack](lc))._engine()
W, H = size
rgb = torch.rand((F, 3, H, W), dtype=torch.float32, device='cuda')
comp = torch.empty((F, H, W), dtype=torch.float32, device='cuda')
out = torch.empty((F, 3, H, W), dtype=torch.float32, device='cuda')
eng.modulate_frames(rgb, 0, out=comp)
fn = (lambda: eng.modulate_frames(rgb, 0, out=comp)) if what == 'mod' else (lambda: eng.demodulate_frames(comp, 0, out=out))
for _ in range(2): fn()
torch.cuda.synchronize()
ts = []
for _ in range(reps):
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record(); fn(); e1.record(); torch.cuda.synchronize(); ts.append(e0.elapsed_time(e1))
ms = sorted(ts)[len(ts) // 2]
print('%-10s %s %dx%d x %d frames: %.3f ms %.1f Gpx/s' % (stack, what, W, H, F, ms, F * W * H / ms / 1e6))
