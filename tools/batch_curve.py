"""Throughput against batch size: PAL-D 720x576 demodulate_frames at F = 1 .. 1000 frames resident in HBM, the
single-image PIL entry point, and the per-row protocol.  python tools/batch_curve.py [stack] -> profiles/r03_batch_curve.txt"""
import sys, time, numpy, torch
sys.path.insert(0, '.'); sys.path.insert(0, 'tests')
import os
import stacks
from color_modem_amd import image, testing, _native
if os.environ.get('CM_LIB'): _native.LIB_PATH = os.environ['CM_LIB']
stack = sys.argv[1] if len(sys.argv) > 1 else 'pal_d'
W, H = 720, 576
modem = stacks.make(stack, (W, H))
im = image.ImageModem(modem)
eng = im._engine()
print(eng.describe())
base = torch.from_numpy(testing.synthetic_composite(4, H, W)).cuda()
if len(sys.argv) > 2: eng.set_small_batch(sys.argv[2])    # auto | rows | segments | scan
print('%6s %10s %12s %10s %12s   (tensor in HBM -> tensor in HBM: HIP events around one call, median of 7 | the same launch replayed 20 x from a HIP graph: no host in the loop)' % ('frames', 'ms', 'us / frame', 'Gpixel/s', 'graph ms'))
FRAMES = (1, 2, 4, 8, 16, 32, 64, 128, 256, 512, 1000) if len(sys.argv) < 4 else tuple(int(v) for v in sys.argv[3].split(','))
for F in FRAMES:
    comp = base.repeat((F + 3) // 4, 1, 1)[:F].contiguous()
    out = torch.empty((F, 3, H, W), dtype=torch.float32, device='cuda')
    for _ in range(3): eng.demodulate_frames(comp, 0, out=out)
    torch.cuda.synchronize()
    ts = []
    for _ in range(7):
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record(); eng.demodulate_frames(comp, 0, out=out); e1.record(); torch.cuda.synchronize(); ts.append(e0.elapsed_time(e1))
    ms = sorted(ts)[3]
    gms = float('nan')
    try:
        side = torch.cuda.Stream()
        with torch.cuda.stream(side):
            eng.demodulate_frames(comp, 0, out=out)
            graph = torch.cuda.CUDAGraph()
            with torch.cuda.graph(graph, stream=side):
                for _ in range(20): eng.demodulate_frames(comp, 0, out=out)
        torch.cuda.synchronize()
        tg = []
        for _ in range(5):
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record(); graph.replay(); e1.record(); torch.cuda.synchronize(); tg.append(e0.elapsed_time(e1) / 20)
        gms = sorted(tg)[2]
    except Exception as e:
        print('graph capture failed:', e)
    print('%6d %10.4f %12.1f %10.2f %12.4f' % (F, ms, ms * 1e3 / F, F * W * H / ms / 1e6, gms), flush=True)
# what the host pays per call (no synchronisation: the launch queue absorbs the kernels)
comp = base[:1].contiguous(); out = torch.empty((1, 3, H, W), dtype=torch.float32, device='cuda')
for _ in range(50): eng.demodulate_frames(comp, 0, out=out)
torch.cuda.synchronize(); t0 = time.perf_counter()
for _ in range(2000): eng.demodulate_frames(comp, 0, out=out)
t1 = time.perf_counter(); torch.cuda.synchronize(); t2 = time.perf_counter()
print('one frame, 2000 calls back to back: host %.1f us per call to enqueue, %.1f us per call until the last one has finished' % ((t1 - t0) / 2000 * 1e6, (t2 - t0) / 2000 * 1e6))
# host-visible latency of one frame: numpy in -> numpy out (pageable memory, includes both PCIe copies), and the PIL image path
comp1 = testing.synthetic_composite(1, H, W)
for _ in range(3): im.demodulate_frames(comp1, 0)
t0 = time.perf_counter()
for _ in range(20): im.demodulate_frames(comp1, 0)
dt = (time.perf_counter() - t0) / 20
print('one frame, numpy in -> numpy out (host wall, PCIe included): %.3f ms  (%.2f Gpixel/s)' % (dt * 1e3, W * H / dt / 1e9))
try:
    from PIL import Image
    img = Image.frombytes('L', (W, H), numpy.uint8(numpy.clip(comp1[0] * 153 + 51, 0, 255)).tobytes())
    for _ in range(3): im.demodulate(img, 0)
    t0 = time.perf_counter()
    for _ in range(20): im.demodulate(img, 0)
    dt = (time.perf_counter() - t0) / 20
    print('one frame, ImageModem.demodulate(PIL image) (fused uint8 boundary, host wall): %.3f ms' % (dt * 1e3))
except Exception as e:
    print('PIL path:', e)
# the per-row protocol
m = stacks.make(stack, (W, H))
rows = comp1[0]
for y in range(0, 8, 2): m.demodulate(0, y, rows[y])
best = 1e9
for rep in range(3):
    t0 = time.perf_counter()
    for y in range(0, H, 2): m.demodulate(1 + rep, y, rows[y])
    for y in range(1, H, 2): m.demodulate(1 + rep, y, rows[y])
    best = min(best, time.perf_counter() - t0)
print('per-row protocol (Modem.demodulate, one row per call): %.1f ms per frame, %.1f us per row' % (best * 1e3, best / H * 1e6))
