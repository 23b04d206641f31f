"""Row-parallel scan kernel against the streaming kernel: python tools/scan_check.py [stack] [width] [height] [frames]
(CM_LIB=... picks a development library)."""
import os, sys, numpy, torch
sys.path.insert(0, '.'); sys.path.insert(0, 'tests')
import stacks
from color_modem_amd import image, testing, _native
if os.environ.get('CM_LIB'): _native.LIB_PATH = os.environ['CM_LIB']
stack = sys.argv[1] if len(sys.argv) > 1 else 'pal_d'
W = int(sys.argv[2]) if len(sys.argv) > 2 else 720
H = int(sys.argv[3]) if len(sys.argv) > 3 else 576
F = int(sys.argv[4]) if len(sys.argv) > 4 else 1
im = image.ImageModem(stacks.make(stack, (W, H)))
eng = im._engine()
comp = torch.from_numpy(testing.synthetic_composite(F, H, W)).cuda()
if 'secam' in stack:      # the FM discriminator is ill-conditioned on noise: a valid signal from the library's own encoder
    comp = eng.modulate_frames(torch.from_numpy(testing.synthetic_rgb(F, H, W)).cuda(), 1)
outs = {}
for mode in ('rows', 'segments', 'scan'):
    try:
        eng.set_small_batch(mode)
    except NotImplementedError as e:
        print(mode, 'not available:', e)
        continue
    out = torch.empty((F, 3, H, W), dtype=torch.float32, device='cuda')
    for _ in range(3): eng.demodulate_frames(comp, 1, out=out)
    torch.cuda.synchronize()
    ts = []
    for _ in range(9):
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record(); eng.demodulate_frames(comp, 1, out=out); e1.record(); torch.cuda.synchronize(); ts.append(e0.elapsed_time(e1))
    outs[mode] = out.cpu().numpy()
    print('%-9s %.1f us per launch (median of 9; %d frame(s) of %dx%d)' % (mode, 1e3 * sorted(ts)[4], F, W, H))
ref = outs['rows']
for mode in ('segments', 'scan'):
    if mode in outs:
        d = numpy.abs(outs[mode] - ref)
        i = numpy.unravel_index(d.argmax(), d.shape)
        print('%-9s vs rows: max |diff| %.3g of %.3g at %s; rows with diff > 1e-5: %s' % (mode, d.max(), numpy.abs(ref).max(), i,
              sorted(set(numpy.argwhere(d > 1e-5)[:, 2].tolist()))[:20]))
        if d.max() > 1e-5:
            f, p, r, _ = i
            print('   row', r, 'plane', p, ':', numpy.round(d[f, p, r, ::48], 6))
