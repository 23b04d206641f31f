"""Comb wrappers around the PAL delay-line decoders, 720x576: python tools/quick_bench_wrapped.py [frames] [stack ...] [float]
(a trailing `float`: the float rows only - one kernel per run for the PMC passes of tools/pmc_any.sh)"""
import sys, torch
sys.path.insert(0, '.'); sys.path.insert(0, 'tests')
import stacks
from color_modem_amd import image, testing
F = int(sys.argv[1]) if len(sys.argv) > 1 else 400
FLOAT_ONLY = sys.argv[-1] == 'float'
if FLOAT_ONLY: del sys.argv[-1]
names = sys.argv[2:] or ['simple3d_pald', 'simple_pald', 'simple3d_pal3d', 'simple3d_pald_notch', 'simple3d_pald_minavg']
for name in names:
    eng = image.ImageModem(stacks.make(name, (720, 576)))._engine()
    comp = torch.from_numpy(testing.synthetic_composite(4, 576, 720)).cuda().repeat(F // 4, 1, 1).contiguous()
    out = torch.empty((F, 3, 576, 720), dtype=torch.float32, device='cuda')
    for _ in range(2): eng.demodulate_frames(comp, 0, out=out)
    torch.cuda.synchronize()
    ts = []
    for _ in range(5):
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record(); eng.demodulate_frames(comp, 0, out=out); e1.record(); torch.cuda.synchronize(); ts.append(e0.elapsed_time(e1))
    ms = sorted(ts)[2]
    print('%-22s frames %d  ms %.3f  Gpx/s %.1f' % (name, F, ms, F * 576 * 720 / ms / 1e6), flush=True)
    if FLOAT_ONLY:
        del comp, out
        continue
    c8 = torch.randint(0, 256, (F, 576, 720), dtype=torch.uint8, device='cuda')
    o8 = torch.empty((F, 576, 720, 3), dtype=torch.uint8, device='cuda')
    for _ in range(2): eng.demodulate_frames_u8(c8, 0, out=o8)
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record(); eng.demodulate_frames_u8(c8, 0, out=o8); e1.record(); torch.cuda.synchronize()
    print('%-22s uint8: ms %.3f  Gpx/s %.1f' % (name, e0.elapsed_time(e1), F * 576 * 720 / e0.elapsed_time(e1) / 1e6), flush=True)
    del comp, out, c8, o8
