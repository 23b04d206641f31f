"""Small-batch latency through the fused byte boundary (ImageModem's one picture in, one picture out - cli.py's workload):
python tools/u8_small_batch.py   (HIP events around one call, and the same launch replayed from a HIP graph, per small-batch mode)"""
import sys, numpy, torch
sys.path.insert(0, '.'); sys.path.insert(0, 'tests')
import stacks
from color_modem_amd import image


def timed(fn):
    for _ in range(3): fn()
    torch.cuda.synchronize()
    ts = []
    for _ in range(7):
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record(); fn(); e1.record(); torch.cuda.synchronize(); ts.append(e0.elapsed_time(e1))
    side = torch.cuda.Stream()
    with torch.cuda.stream(side):
        fn()
        graph = torch.cuda.CUDAGraph()
        with torch.cuda.graph(graph, stream=side):
            for _ in range(20): fn()
    torch.cuda.synchronize()
    tg = []
    for _ in range(5):
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record(); graph.replay(); e1.record(); torch.cuda.synchronize(); tg.append(e0.elapsed_time(e1) / 20)
    return 1e3 * sorted(ts)[3], 1e3 * sorted(tg)[2]


rng = numpy.random.default_rng(3)
for stack, size in (('pal_d', (720, 576)), ('pal_3d', (720, 576)), ('ntsc_comb_3d', (720, 480)), ('secam', (720, 576)), ('pal_s', (720, 576)),
                    ('ntsc', (720, 480))):
    W, H = size
    eng = image.ImageModem(stacks.make(stack, size))._engine()
    for F in (1, 4):
        comp8 = torch.from_numpy(rng.integers(40, 200, size=(F, H, W), dtype=numpy.uint8)).cuda()
        rgb8 = torch.from_numpy(rng.integers(0, 256, size=(F, H, W, 3), dtype=numpy.uint8)).cuda()
        out_d = torch.empty((F, H, W, 3), dtype=torch.uint8, device='cuda')
        out_m = torch.empty((F, H, W), dtype=torch.uint8, device='cuda')
        for mode in ('rows', 'auto'):
            eng.set_small_batch(mode)
            a = timed(lambda: eng.demodulate_frames_u8(comp8, 0, out=out_d))
            line = '%-13s %d frame(s) %dx%d  %-5s  decode u8: %6.1f us / call, %6.1f us replayed' % (stack, F, W, H, mode, a[0], a[1])
            if stack in ('pal_s', 'ntsc', 'secam'):
                b = timed(lambda: eng.modulate_frames_u8(rgb8, 0, out=out_m))
                line += '   encode u8: %6.1f us / call, %6.1f us replayed' % b
            print(line, flush=True)
