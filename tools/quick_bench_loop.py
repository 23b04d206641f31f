"""Run the headline demodulation back to back for N seconds (power / clock probes): python tools/quick_bench_loop.py [seconds]"""
import sys, time, torch
sys.path.insert(0, '.'); sys.path.insert(0, 'tests')
import stacks
from color_modem_amd import image, testing
secs = float(sys.argv[1]) if len(sys.argv) > 1 else 5
eng = image.ImageModem(stacks.make('pal_d', (720, 576)))._engine()
F = 1000
comp = torch.from_numpy(testing.synthetic_composite(4, 576, 720)).cuda().repeat(F // 4, 1, 1).contiguous()
out = torch.empty((F, 3, 576, 720), dtype=torch.float32, device='cuda')
eng.demodulate_frames(comp, 0, out=out); torch.cuda.synchronize()
t0 = time.time(); n = 0
while time.time() - t0 < secs:
    for _ in range(20): eng.demodulate_frames(comp, 0, out=out)
    torch.cuda.synchronize(); n += 20
dt = time.time() - t0
print('launches', n, 'ms per launch %.3f' % (dt / n * 1e3), 'Gpx/s %.1f' % (n * F * 576 * 720 / dt / 1e9))
