"""Does the time of a wide decoder depend on where its buffers lie?  python tools/align_probe.py [stack] [width] [mpix]
One process: the same engine and input values, the input / output buffers carved out of two large allocations at different byte
offsets (and once after a fresh allocation of both), HIP events, median of 5."""
import sys
import torch
sys.path.insert(0, '.'); sys.path.insert(0, 'tests')
import stacks
from color_modem_amd import image, testing

name = sys.argv[1] if len(sys.argv) > 1 else 'pal_d'
w = int(sys.argv[2]) if len(sys.argv) > 2 else 1920
mpix = float(sys.argv[3]) if len(sys.argv) > 3 else 1200.0
h = 480 if name.startswith('ntsc') else 576
F = int(mpix * 1e6 / (w * h)) // 4 * 4
eng = image.ImageModem(stacks.make(name, (w, h)))._engine()
src = torch.from_numpy(testing.synthetic_composite(4, h, w)).cuda().repeat(F // 4, 1, 1).contiguous()
n_in, n_out = F * h * w, F * 3 * h * w
SLACK = 1 << 22      # floats: 16 MiB


def timed(comp, out):
    for _ in range(2):
        eng.demodulate_frames(comp, 0, out=out)
    torch.cuda.synchronize()
    ts = []
    for _ in range(5):
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record(); eng.demodulate_frames(comp, 0, out=out); e1.record(); torch.cuda.synchronize(); ts.append(e0.elapsed_time(e1))
    return sorted(ts)[2]


print('%s %dx%d, %d frames; %s' % (name, w, h, F, eng.describe().split(';')[0]))
for trial in range(3):
    big_in = torch.empty(n_in + SLACK, dtype=torch.float32, device='cuda')
    big_out = torch.empty(n_out + SLACK, dtype=torch.float32, device='cuda')
    print('allocation %d: in at 0x%x, out at 0x%x' % (trial, big_in.data_ptr(), big_out.data_ptr()))
    for off_in, off_out in ((0, 0), (0, 64), (0, 1024), (0, 16384), (0, 1 << 18), (0, 1 << 20), (64, 0), (1024, 0), (1 << 18, 0), (1 << 20, 1 << 19)):
        comp = big_in[off_in:off_in + n_in].view(F, h, w)
        comp.copy_(src)
        out = big_out[off_out:off_out + n_out].view(F, 3, h, w)
        ms = timed(comp, out)
        print('   in +%8d B  out +%8d B   %8.3f ms  %6.1f Gpx/s' % (off_in * 4, off_out * 4, ms, F * w * h / ms / 1e6), flush=True)
    del big_in, big_out, comp, out
    torch.cuda.empty_cache()
    pad = torch.empty((trial + 1) * 300_000_000, dtype=torch.float32, device='cuda')    # move the next allocation elsewhere
