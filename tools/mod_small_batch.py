"""Small-batch latency of the encoders: python tools/mod_small_batch.py  (one frame / a few frames, HIP events around one call and
the same launch replayed from a HIP graph; the per-row protocol)"""
import sys, time, numpy, torch
sys.path.insert(0, '.'); sys.path.insert(0, 'tests')
import stacks
from color_modem_amd import image, testing
for stack, size in (('pal_s', (720, 576)), ('ntsc', (720, 480)), ('secam', (720, 576))):
    W, H = size
    im = image.ImageModem(stacks.make(stack, size)); eng = im._engine()
    for F in (1, 2, 4, 16):
        rgb = torch.from_numpy(testing.synthetic_rgb(F, H, W)).cuda()
        out = torch.empty((F, H, W), dtype=torch.float32, device='cuda')
        for _ in range(3): eng.modulate_frames(rgb, 0, out=out)
        torch.cuda.synchronize()
        ts = []
        for _ in range(7):
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record(); eng.modulate_frames(rgb, 0, out=out); e1.record(); torch.cuda.synchronize(); ts.append(e0.elapsed_time(e1))
        side = torch.cuda.Stream()
        with torch.cuda.stream(side):
            eng.modulate_frames(rgb, 0, out=out)
            graph = torch.cuda.CUDAGraph()
            with torch.cuda.graph(graph, stream=side):
                for _ in range(20): eng.modulate_frames(rgb, 0, out=out)
        torch.cuda.synchronize()
        tg = []
        for _ in range(5):
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record(); graph.replay(); e1.record(); torch.cuda.synchronize(); tg.append(e0.elapsed_time(e1) / 20)
        print('%-6s modulate %2d frame(s) of %dx%d: %.1f us around one call, %.1f us replayed from a graph' % (stack, F, W, H, 1e3 * sorted(ts)[3], 1e3 * sorted(tg)[2]), flush=True)
    enc = stacks.make(stack, size)
    rgb = testing.synthetic_rgb(1, H, W)[0]
    for y in range(0, 8, 2): enc.modulate(0, y, rgb[0, y], rgb[1, y], rgb[2, y])
    t0 = time.perf_counter()
    for y in range(0, H, 2): enc.modulate(1, y, rgb[0, y], rgb[1, y], rgb[2, y])
    dt = time.perf_counter() - t0
    print('%-6s per-row protocol (Modem.modulate): %.1f us per row' % (stack, dt / (H // 2 + H % 2) * 1e6), flush=True)
