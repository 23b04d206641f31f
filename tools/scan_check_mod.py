"""Row-parallel modulator against the streaming modulator: python tools/scan_check_mod.py [stack] [width] [height] [frames]"""
import os, sys, numpy, torch
sys.path.insert(0, '.'); sys.path.insert(0, 'tests')
import stacks
from color_modem_amd import image, testing, _native
if os.environ.get('CM_LIB'): _native.LIB_PATH = os.environ['CM_LIB']
stack = sys.argv[1] if len(sys.argv) > 1 else 'pal_s'
W = int(sys.argv[2]) if len(sys.argv) > 2 else 720
H = int(sys.argv[3]) if len(sys.argv) > 3 else 576
F = int(sys.argv[4]) if len(sys.argv) > 4 else 1
im = image.ImageModem(stacks.make(stack, (W, H)))
eng = im._engine()
rgb = torch.from_numpy(testing.synthetic_rgb(F, H, W)).cuda()
outs = {}
for mode in ('rows', 'scan'):
    eng.set_small_batch(mode)
    out = torch.empty((F, H, W), dtype=torch.float32, device='cuda')
    for _ in range(3): eng.modulate_frames(rgb, 1, out=out)
    torch.cuda.synchronize()
    ts = []
    for _ in range(9):
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record(); eng.modulate_frames(rgb, 1, out=out); e1.record(); torch.cuda.synchronize(); ts.append(e0.elapsed_time(e1))
    outs[mode] = out.cpu().numpy()
    print('%-5s %-9s %.1f us per launch (median of 9; %d frame(s) of %dx%d)' % (stack, mode, 1e3 * sorted(ts)[4], F, W, H))
d = numpy.abs(outs['scan'] - outs['rows'])
print('%-5s scan vs rows: max |diff| %.3g of %.3g at %s' % (stack, d.max(), numpy.abs(outs['rows']).max(), numpy.unravel_index(d.argmax(), d.shape)))
