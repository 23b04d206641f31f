# -*- coding: utf-8 -*-
"""Headline benchmark: Mpixels/s demodulated, 720x576 PAL-BG 2D comb (PalDModem), synthetic stream.

    python bench.py [--gpus N] [--steps K] [--warmup W] [--frames F]

One "step" = one pass of the hot path (cm_demodulate_frames, include/color_modem_hip.h) over one
batch of F frames per GPU that is already resident in HBM.  The input is a valid colour signal:
smoothed random RGB frames encoded by this library's own PAL modulator on the device, outside the
timed region.

N > 1: one rank per GPU (torch.distributed, backend nccl = RCCL).  Either the caller launches this
file under torch.distributed.run (the driver does), or - plain `python bench.py --gpus N` - this
process starts that launcher as a CHILD before anything here touches the GPU and relays rank 0's
line.  Frames are independent, so every rank demodulates its own F-frame stream with
no data-path collective ("weak" scaling, BASELINE.json configs[4]); the only collectives inside the
timed region are the two barriers.  After it, the output gather the survey asks to be timed on its
own (8e) is measured separately on a bounded batch (`gather_ms`), and `rccl_ranks` is the world
size RCCL itself reports after an all_reduce on device memory.

Prints ONE JSON line (rank 0).  `roofline.achieved` = algorithmic bytes (16 B/pixel: 4 read +
12 written) / mean duration of the demod kernel measured with HIP events on the
launch stream.  `check` = frames of the timed output compared with the float64 CPU oracle after
the timed region.  `cpu_baseline` = the same oracle (a port of the reference's per-line algorithm,
oracle/cm_oracle.cpp) timed on this host's cores on a bounded sample of the same workload.
"""

import argparse
import json
import os
import socket
import subprocess
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

WIDTH, HEIGHT = 720, 576
HBM_PEAK_GBPS = 8000.0      # MI355X HBM3E spec peak (MI355X_MICROARCH.md)
BYTES_PER_PIXEL = 16        # float32: one composite sample read, R, G, B written
# vector-pipe side of the same kernel: float32 FMA-equivalents the kernel executes per pixel, counted
# from the ISA of the interior bodies of this tree (tools/isa_fma_census.py: 1 per scalar float instruction, 2 per v_pk_*_f32;
# stage A 76.5 + stage B 129.0, the per-stage table in profiles/r04_headline_bound.txt), against the chip's vector float32
# peak and against what a bare v_fma_f32 loop sustains under the board's power cap (profiles/r01_ubench_valu.txt)
FMA_EQ_PER_PIXEL = 205.5
VALU_PEAK_TFLOPS = 157.3
VALU_SUSTAINED_TFLOPS = 119.0


def synthetic_stream(torch, eng, n_frames, seed, first_frame, device, height=HEIGHT, width=WIDTH, keep_rgb=False):
    """composite[F, 576, 720] float32: F distinct frames of smoothed uniform RGB (4-tap box along the line), encoded by
    the library's own PAL modulator (PalDModem.modulate = qam.py:28-32 with the V switch of pal.py:48-52) under the
    frame numbers the decoder will be given.  Generated on the device (a repeated frame would sit in the Infinity
    Cache and hide HBM traffic).  `eng` may be any encoder engine (other_configs: NTSC, SECAM); keep_rgb also
    returns the rgb[F, 3, H, W] pictures."""
    gen = torch.Generator(device=device)
    gen.manual_seed(seed)
    out = torch.empty((n_frames, height, width), dtype=torch.float32, device=device)
    pictures = torch.empty((n_frames, 3, height, width), dtype=torch.float32, device=device) if keep_rgb else None
    chunk = 40
    for f0 in range(0, n_frames, chunk):
        n = min(chunk, n_frames - f0)
        wide = torch.rand((n, 3, height, width + 3), generator=gen, device=device, dtype=torch.float32)
        rgb = ((wide[..., 0:width] + wide[..., 1:width + 1] + wide[..., 2:width + 2] + wide[..., 3:width + 3]) * 0.25).contiguous()
        eng.modulate_frames(rgb, first_frame + f0, out=out[f0:f0 + n])
        if keep_rgb:
            pictures[f0:f0 + n] = rgb
    torch.cuda.synchronize()
    return (out, pictures) if keep_rgb else out


def _timed(torch, fn, reps=5, warm=2):
    """Mean HIP-event time (ms) of `fn` over `reps` launches after `warm` untimed ones, on the stream the library launches on."""
    for _ in range(warm):
        fn()
    torch.cuda.synchronize()
    ev = [(torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)) for _ in range(reps)]
    for a, b in ev:
        a.record()
        fn()
        b.record()
    torch.cuda.synchronize()
    ts = [a.elapsed_time(b) for a, b in ev]
    return sum(ts) / len(ts)


def _rel_err(numpy, got, want):
    """max |got - want| / max |want| per plane (the tolerance convention of SURVEY App. C), worst plane."""
    got = numpy.asarray(got, dtype=numpy.float64)
    want = numpy.asarray(want, dtype=numpy.float64)
    if want.ndim == 2:
        got, want = got[None], want[None]
    return max(float(numpy.max(numpy.abs(g - w)) / numpy.max(numpy.abs(w))) for g, w in zip(got, want))


def other_configs(torch, device, frames):
    """BASELINE.json configs[2] and configs[3], timed AFTER the headline's timed region so that the
    driver's record covers every single-GPU configuration: HIP events over 5 launches after 2 warm-ups, inputs resident in HBM,
    two frames of every output compared with the float64 oracle.  16 algorithmic bytes per pixel in either direction (the round
    trip: 32)."""
    import numpy
    from color_modem_amd import comb, image, line
    from color_modem_amd.color import ntsc, secam
    from oracle import cm_oracle
    res = []

    def entry(workload, ms, px, bytes_per_px, err, kernel):
        gbps = bytes_per_px * px / (ms * 1e-3) / 1e9
        return {'workload': workload, 'ms': round(ms, 4), 'mpixels_s': round(px / (ms * 1e-3) / 1e6, 1),
                'roofline': {'bound': 'hbm', 'achieved': round(gbps, 1), 'peak': HBM_PEAK_GBPS, 'unit': 'GB/s',
                             'frac': round(gbps / HBM_PEAK_GBPS, 4), 'algorithmic_bytes_per_launch': bytes_per_px * px},
                'check': {'max_rel_err': float('%.3g' % err), 'tolerance': 1e-5, 'frames': 2, 'what': 'vs the float64 CPU oracle'},
                'frames': frames, 'kernel': kernel.split('\n')[0][:160]}

    # configs[2]: NTSC 3D comb (Simple3DCombModem(NtscCombModem), comb.py:96-127 over ntsc.py:61-82), 720x480
    w, h = 720, 480
    lc = line.LineConfig((w, h))
    modem = comb.Simple3DCombModem(ntsc.NtscCombModem(lc))
    eng = image.ImageModem(modem)._engine()
    enc = image.ImageModem(ntsc.NtscModem(lc))._engine()
    comp = synthetic_stream(torch, enc, frames, 2234, 0, device, h, w)
    out = torch.empty((frames, 3, h, w), dtype=torch.float32, device=device)
    ms = _timed(torch, lambda: eng.demodulate_frames(comp, 0, out=out))
    pick = max(0, frames - 2)
    want = cm_oracle.demodulate_frames_f32(modem, comp[pick:pick + 2].cpu().numpy(), pick, 2)
    got = out[pick:pick + 2].cpu().numpy()
    err = max(_rel_err(numpy, got[i], want[i]) for i in range(got.shape[0]))
    res.append(entry('NTSC 3D comb (Simple3DCombModem(NtscCombModem)) demodulate, 720x480, %d-frame synthetic stream (NTSC-encoded on '
                     'the device)' % frames, ms, frames * w * h, 16, err, eng.describe()))
    del comp, out

    # configs[3]: SECAM IIIb encode, decode and the round trip (secam.py:240-304), 720x576
    w, h = WIDTH, HEIGHT
    modem = secam.SecamModem(line.LineConfig((w, h)))
    eng = image.ImageModem(modem)._engine()
    comp, rgb = synthetic_stream(torch, eng, frames, 3234, 0, device, h, w, keep_rgb=True)
    comp2 = torch.empty_like(comp)
    out = torch.empty((frames, 3, h, w), dtype=torch.float32, device=device)
    px = frames * w * h
    ms_enc = _timed(torch, lambda: eng.modulate_frames(rgb, 0, out=comp2))
    ms_dec = _timed(torch, lambda: eng.demodulate_frames(comp, 0, out=out))

    def round_trip():
        eng.modulate_frames(rgb, 0, out=comp2)
        eng.demodulate_frames(comp2, 0, out=out)
    ms_rt = _timed(torch, round_trip)
    rgb_h = rgb[pick:pick + 2].cpu().numpy()
    want_c = cm_oracle.modulate_frames_f32(modem, rgb_h, pick, 2)
    got_c = comp2[pick:pick + 2].cpu().numpy()
    err_enc = max(_rel_err(numpy, got_c[i], want_c[i]) for i in range(got_c.shape[0]))
    want = cm_oracle.demodulate_frames_f32(modem, got_c, pick, 2)          # the decoder on what the device's encoder produced
    got = out[pick:pick + 2].cpu().numpy()
    err_dec = max(_rel_err(numpy, got[i], want[i]) for i in range(got.shape[0]))
    want_rt = cm_oracle.demodulate_frames_f32(modem, want_c.astype(numpy.float32), pick, 2)   # the oracle's own round trip
    err_rt = max(_rel_err(numpy, got[i], want_rt[i]) for i in range(got.shape[0]))
    kern = eng.describe()
    name = 'SECAM IIIb (SecamModem) %s, 720x576, %d frames of smoothed random RGB'
    kern_enc = 'secam_mod_kernel; calls per workgroup 64 (one wavefront, three-plane input tiles), halo 0'      # cm_plan_describe names the decoder
    res.append(entry(name % ('encode', frames), ms_enc, px, 16, err_enc, kern_enc))
    res.append(entry(name % ('decode', frames), ms_dec, px, 16, err_dec, kern))
    res.append(entry(name % ('encode + decode round trip through HBM', frames), ms_rt, px, 32, max(err_enc, err_dec), 'secam_mod_kernel, then ' + kern))
    # each leg against the oracle on the same input is the parity statement; the device's round trip against the oracle's own
    # round trip is reported beside it (the discriminator amplifies the encoder's float32 rounding by ~ 1 / fdev: not gated)
    res[-1]['check']['end_to_end_vs_oracle_round_trip'] = float('%.3g' % err_rt)
    return res


def library_sha16():
    """first 16 hex digits of the SHA-256 of the shared object this process runs (profiles/traffic.json is tied to it)"""
    import hashlib
    from color_modem_amd import _native
    h = hashlib.sha256()
    with open(_native.LIB_PATH, 'rb') as fh:
        for block in iter(lambda: fh.read(1 << 20), b''):
            h.update(block)
    return h.hexdigest()[:16]


def sources_sha16():
    """first 16 hex digits of the SHA-256 over the library's sources (csrc/*, include/color_modem_hip.h, names and contents) and the compile
    flags of __graft_entry__.build(): the same kernels whatever directory they were compiled in (hipcc derives the compilation-unit id it
    embeds in an object from the output path, so the shared object's own hash differs between checkouts of one tree)"""
    import hashlib
    import __graft_entry__ as entry
    csrc = os.path.join(ROOT, 'color_modem_amd', 'csrc')
    files = sorted(os.path.join(csrc, f) for f in os.listdir(csrc)) + [os.path.join(ROOT, 'include', 'color_modem_hip.h')]
    h = hashlib.sha256(' '.join(entry.HIPCC_FLAGS).encode())
    for path in files:
        h.update(os.path.basename(path).encode())
        with open(path, 'rb') as fh:
            h.update(fh.read())
    return h.hexdigest()[:16]


class ClockSampler(object):
    """The shader clock the board holds while the timed steps run, read from sysfs (pp_dpm_sclk: the level marked '*') by a host thread
    every few milliseconds - the kernel is power-bound, and what it gives back shows here.  The card is the one whose
    PCI address is the HIP device's (a box shows every GPU of its host in sysfs, only one of them to HIP); without a match, the busiest
    card, said so in 'source'.  None when no file is readable (another driver layout, a restricted container)."""

    def __init__(self, pci_bus_id):
        import glob
        import threading
        self.paths = sorted(glob.glob('/sys/class/drm/card*/device/pp_dpm_sclk'))
        self.mine = [p for p in self.paths if pci_bus_id and os.path.basename(os.path.realpath(os.path.dirname(p))).lower().startswith(pci_bus_id.lower())]
        self.watch = self.mine[:1] or self.paths
        self.samples, self._stop = dict((p, []) for p in self.watch), threading.Event()
        self._thread = threading.Thread(target=self._run, daemon=True)

    @staticmethod
    def _read(path):
        try:
            with open(path) as fh:
                for ln in fh:
                    if ln.rstrip().endswith('*'):
                        return float(ln.split(':')[1].split('M')[0])
        except (OSError, ValueError, IndexError, TypeError):
            return None
        return None

    def _run(self):
        while not self._stop.is_set():
            for p in self.watch:
                v = self._read(p)
                if v is not None:
                    self.samples[p].append(v)
            self._stop.wait(0.003)

    def start(self):
        if self.watch:
            self._thread.start()

    def stop(self):
        self._stop.set()
        if self._thread.is_alive():
            self._thread.join(1.0)
        series = [(sorted(v), p) for p, v in self.samples.items() if v]
        if not series:
            return None
        s, path = max(series, key=lambda e: e[0][len(e[0]) // 2])
        how = 'the card at the HIP device\'s PCI address' if self.mine else 'the busiest of %d cards (no card matched the HIP device\'s PCI address)' % len(series)
        return {'samples': len(s), 'min': s[0], 'median': s[len(s) // 2], 'max': s[-1], 'source': path + ': ' + how}


def host_threads():
    try:
        return max(1, len(os.sched_getaffinity(0)))
    except (AttributeError, OSError):
        return max(1, os.cpu_count() or 1)


def cpu_model():
    try:
        with open('/proc/cpuinfo') as fh:
            for ln in fh:
                if ln.startswith('model name'):
                    return ln.split(':', 1)[1].strip()
    except OSError:
        pass
    return ''


def cpu_baseline(modem, comp_host, first_frame):
    """Time the CPU oracle on frames of the same workload: one thread, then every logical CPU this process may run on,
    then half and a quarter of them (SMT siblings and a container's CPU quota can make fewer threads the faster choice);
    `value` / `cores` report the fastest of the three, `tried` all of them."""
    from oracle import cm_oracle
    threads = host_threads()
    n = comp_host.shape[0]
    n1 = max(1, min(4, n))
    t0 = time.perf_counter()
    cm_oracle.demodulate_frames_f32(modem, comp_host[:n1], first_frame, 1)
    single = n1 * WIDTH * HEIGHT / (time.perf_counter() - t0) / 1e6
    tried = {}
    for nt in sorted(set(max(1, min(threads // d, n)) for d in (1, 2, 4)), reverse=True):
        t0 = time.perf_counter()
        cm_oracle.demodulate_frames_f32(modem, comp_host, first_frame, nt)
        tried[nt] = round(n * WIDTH * HEIGHT / (time.perf_counter() - t0) / 1e6, 3)
    best = max(tried, key=lambda c: tried[c])
    return {'value': tried[best], 'unit': 'Mpixels/s', 'cores': best, 'kind': 'port',
            'single_thread_value': round(single, 3), 'cpu': cpu_model(), 'logical_cpus': threads,
            'tried': {str(c): v for c, v in sorted(tried.items())},
            'sample': '%d frames 720x576 of the benchmark stream (PAL-encoded), PAL-D demodulate, float64 C++ oracle '
                      '(oracle/cm_oracle.cpp), frames sharded over threads: all %d logical CPUs this process may run on, half '
                      'and a quarter of them - value is the fastest (cores = its thread count); single-thread figure on %d '
                      'frames. Reference itself (numpy/scipy, 1 core, survey container): 0.52 Mpixels/s' % (n, threads, n1)}


def check_frames(torch, modem, comp, out, first_frame, picks):
    """Frames `picks` of the timed output against the float64 oracle (CPU, after the timed region)."""
    import numpy
    from oracle import cm_oracle
    worst, bad = 0.0, 0
    for f in picks:
        want = cm_oracle.demodulate_frames_f32(modem, comp[f:f + 1].cpu().numpy(), first_frame + f, 1).astype(numpy.float64)
        got = out[f:f + 1].cpu().numpy().astype(numpy.float64)
        for p in range(3):
            worst = max(worst, float(numpy.max(numpy.abs(got[0, p] - want[0, p])) / numpy.max(numpy.abs(want[0, p]))))
            bad += int(numpy.count_nonzero(numpy.abs(got[0, p] - want[0, p]) > 1e-6 + 1e-5 * numpy.abs(want[0, p])))
    return worst, bad


def count_gpus():
    """GPUs this process could use, counted WITHOUT touching the HIP runtime (the parent of the ranks must not hold a
    context on GPU 0): the KFD topology in sysfs lists every node, GPUs are the ones with SIMDs; HIP_VISIBLE_DEVICES /
    ROCR_VISIBLE_DEVICES / CUDA_VISIBLE_DEVICES narrow the set.  Returns 0 when sysfs is not there (then the launch is a
    gloo rehearsal unless CM_BENCH_BACKEND says otherwise)."""
    n = 0
    base = '/sys/class/kfd/kfd/topology/nodes'
    try:
        for node in os.listdir(base):
            try:
                with open(os.path.join(base, node, 'properties')) as fh:
                    props = dict(ln.split()[:2] for ln in fh if len(ln.split()) >= 2)
                if int(props.get('simd_count', '0')) > 0:
                    n += 1
            except (OSError, ValueError):
                pass
    except OSError:
        return 0
    for var in ('HIP_VISIBLE_DEVICES', 'ROCR_VISIBLE_DEVICES', 'CUDA_VISIBLE_DEVICES'):
        v = os.environ.get(var)
        if v is not None:
            n = min(n, len([x for x in v.split(',') if x.strip() != '']))
    return n


def launch_ranks(args):
    """Plain `python bench.py --gpus N`: start torch.distributed.run as a child process (this process has not touched
    the GPU and never replaces itself), relay rank 0's JSON line, fail if any rank failed."""
    n_dev = count_gpus()                    # from sysfs: the parent never loads the HIP runtime
    env = dict(os.environ)
    env.setdefault('HSA_ENABLE_IPC_MODE_LEGACY', '0')
    if n_dev < args.gpus and 'CM_BENCH_BACKEND' not in env:
        # fewer GPUs than ranks (a 1-GPU test box): rehearse the same launch path with gloo, ranks sharing the devices;
        # the line then says so ("rccl_ranks": null, "rehearsal": ...), it is not a scaling measurement
        env['CM_BENCH_BACKEND'] = 'gloo'
    with socket.socket() as s:
        s.bind(('127.0.0.1', 0))
        port = s.getsockname()[1]
    cmd = [sys.executable, '-m', 'torch.distributed.run', '--nnodes=1', '--nproc-per-node', str(args.gpus),
           '--master-addr', '127.0.0.1', '--master-port', str(port), os.path.abspath(__file__),
           '--gpus', str(args.gpus), '--steps', str(args.steps), '--warmup', str(args.warmup),
           '--frames', str(args.frames), '--gather-frames', str(args.gather_frames), '--cpu-sample', str(args.cpu_sample)]
    proc = subprocess.run(cmd, env=env, stdout=subprocess.PIPE, universal_newlines=True)
    line = None
    for ln in proc.stdout.splitlines():
        if ln.startswith('{') and '"metric"' in ln:
            line = ln
    if proc.returncode != 0 or line is None:
        sys.stderr.write(proc.stdout)
        raise SystemExit('bench.py: the %d-rank launch failed (exit code %d)' % (args.gpus, proc.returncode or 1))
    print(line)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument('--gpus', type=int, default=1)
    ap.add_argument('--steps', type=int, default=10)
    ap.add_argument('--warmup', type=int, default=2)
    ap.add_argument('--frames', type=int, default=1000, help='frames per GPU per step')
    ap.add_argument('--cpu-sample', type=int, default=-1,
                    help='frames of the CPU baseline sample (0 = skip; default: 4 per host thread, 32 ... 256)')
    ap.add_argument('--other-configs', type=int, default=1,
                    help='N = 1: also time BASELINE configs 3 and 4 after the headline (other_configs in the line; 0 = skip)')
    ap.add_argument('--gather-frames', type=int, default=16, help='N > 1: frames per rank of the separately timed output gather')
    args = ap.parse_args()

    world = int(os.environ.get('WORLD_SIZE', '1'))
    rank = int(os.environ.get('RANK', '0'))
    local_rank = int(os.environ.get('LOCAL_RANK', '0'))
    if args.gpus > 1 and 'WORLD_SIZE' not in os.environ:
        return launch_ranks(args)
    if args.gpus != world:
        raise SystemExit('--gpus %d but WORLD_SIZE=%d' % (args.gpus, world))

    import torch
    backend = os.environ.get('CM_BENCH_BACKEND', 'nccl')     # 'gloo': rehearsal of the N > 1 path on a box with fewer GPUs than ranks
    n_dev = torch.cuda.device_count()                        # (counting does not initialise the runtime on this image)
    if n_dev < 1:
        raise SystemExit('bench.py needs a GPU: the product path has no CPU implementation')
    if backend == 'nccl' and world > n_dev:
        backend = 'gloo'      # more ranks than GPUs on this box (every rank sees the same count): the launch-path rehearsal, reported as such
    dev_index = local_rank if backend == 'nccl' else local_rank % n_dev
    torch.cuda.set_device(dev_index)                         # the rank binds ITS device before the first HIP call of the process
    if not torch.cuda.is_available():
        raise SystemExit('bench.py needs a GPU: the product path has no CPU implementation')
    device = torch.device('cuda', dev_index)
    dist = None
    if world > 1:
        import torch.distributed as dist
        if backend == 'nccl':
            dist.init_process_group('nccl', device_id=device)
        else:
            dist.init_process_group(backend)

    from color_modem_amd import image, line, parallel
    from color_modem_amd.color import pal
    modem = pal.PalDModem(line.LineConfig((WIDTH, HEIGHT)))
    eng = image.ImageModem(modem)._engine()
    frames = args.frames
    first_frame = rank * frames  # every rank continues the frame numbering: all four PAL phases are exercised
    comp = synthetic_stream(torch, eng, frames, 1234 + 7919 * rank, first_frame, device)
    out = torch.empty((frames, 3, HEIGHT, WIDTH), dtype=torch.float32, device=device)

    def barrier():
        torch.cuda.synchronize()
        if dist is not None:
            dist.barrier()
        torch.cuda.synchronize()

    for _ in range(args.warmup):
        eng.demodulate_frames(comp, first_frame, out=out)
    barrier()
    ev = [(torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)) for _ in range(args.steps)]
    props0 = torch.cuda.get_device_properties(device) if device.type == 'cuda' else None
    pci0 = '%04x:%02x:%02x' % (getattr(props0, 'pci_domain_id', 0), getattr(props0, 'pci_bus_id', 0), getattr(props0, 'pci_device_id', 0)) if props0 is not None else ''
    sampler = ClockSampler(pci0) if rank == 0 else None      # reads sysfs from a host thread: nothing is added to the stream
    if sampler:
        sampler.start()
    t0 = time.perf_counter()
    for a, b in ev:
        a.record()                      # same stream the library launches on (torch's current stream)
        eng.demodulate_frames(comp, first_frame, out=out)
        b.record()
    barrier()
    elapsed = time.perf_counter() - t0
    clock_stats = sampler.stop() if sampler else None
    kernel_ms = [a.elapsed_time(b) for a, b in ev]

    # ---- after the timed region -------------------------------------------------------------------------------------
    picks = sorted(set([0, frames // 2 - 1 if frames > 1 else 0, frames - 1])) if rank == 0 else [frames - 1]
    worst, bad = check_frames(torch, modem, comp, out, first_frame, picks)
    rccl_ranks, gather, rank_devices = None, None, None
    if dist is not None:
        props = torch.cuda.get_device_properties(device)
        me = {'rank': rank, 'local_rank': local_rank, 'device_index': dev_index, 'name': props.name,
              'uuid': str(getattr(props, 'uuid', '')), 'pci_bus_id': '%04x:%02x:%02x' % (getattr(props, 'pci_domain_id', 0), getattr(props, 'pci_bus_id', 0), getattr(props, 'pci_device_id', 0)),
              'pid': os.getpid()}
        rank_devices = [None] * world
        dist.all_gather_object(rank_devices, me)      # which GPU every rank really ran on: a mis-bound rank shows as a repeated uuid
        t = torch.tensor([elapsed, worst, float(bad)], dtype=torch.float64, device=device if backend == 'nccl' else 'cpu')
        tmin = t[0:1].clone()
        dist.all_reduce(tmin, op=dist.ReduceOp.MIN)       # the fastest rank's wall time beside the slowest (the reported one)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        elapsed, worst, bad = float(t[0]), float(t[1]), int(t[2])
        rank_ms = {'min': round(float(tmin[0]) / args.steps * 1e3, 4), 'max': round(elapsed / args.steps * 1e3, 4)}
        ones = torch.ones(1, dtype=torch.int32, device=device if backend == 'nccl' else 'cpu')
        dist.all_reduce(ones, op=dist.ReduceOp.SUM)      # one element per rank through the collective library itself
        if backend == 'nccl':
            rccl_ranks = int(ones.item())
        # the output gather, timed on its own: every rank holds the same G * N frame batch, demodulates its share
        # (parallel.demodulate_frames_sharded) and the shares are exchanged with one all_gather
        g = max(1, min(args.gather_frames, frames))
        batch = synthetic_stream(torch, eng, g * world, 4321, 0, device)
        gdev = (lambda x: x) if backend == 'nccl' else (lambda x: x.cpu())
        for timed in (False, True):
            barrier()
            t1 = time.perf_counter()
            local = parallel.demodulate_frames_sharded(eng.demodulate_frames, batch, 0, gather=False)
            torch.cuda.synchronize()
            t2 = time.perf_counter()
            full = parallel.gather_frames(gdev(local), g * world)
            barrier()
            t3 = time.perf_counter()
        lo, hi = parallel.frame_range(g * world, world, rank)
        same = bool(torch.equal(full[lo:hi].to(local.device), local)) and full.shape[0] == g * world
        tg = torch.tensor([t3 - t2, 0.0 if same else 1.0], dtype=torch.float64, device=device if backend == 'nccl' else 'cpu')
        dist.all_reduce(tg, op=dist.ReduceOp.MAX)
        gbytes = g * world * 3 * HEIGHT * WIDTH * 4
        gather = {'gather_ms': round(float(tg[0]) * 1e3, 3), 'frames': g * world, 'bytes_per_rank_received': gbytes,
                  'GBps_per_rank': round(gbytes / max(float(tg[0]), 1e-9) / 1e9, 2),
                  'demodulate_ms': round((t2 - t1) * 1e3, 3), 'own_share_intact': float(tg[1]) == 0.0,
                  'what': 'all_gather of rgb[%d, 3, 576, 720] float32 (every rank receives the whole batch), '
                          'parallel.gather_frames; outside the timed steps' % (g * world)}

    if rank == 0:
        px_step = frames * WIDTH * HEIGHT
        value = world * px_step * args.steps / elapsed / 1e6
        mean_kernel_ms = sum(kernel_ms) / len(kernel_ms)
        achieved = BYTES_PER_PIXEL * px_step / (mean_kernel_ms * 1e-3) / 1e9
        valu_tflops = 2.0 * FMA_EQ_PER_PIXEL * px_step / (mean_kernel_ms * 1e-3) / 1e12
        traffic, traffic_source = None, None
        tpath = os.path.join(ROOT, 'profiles', 'traffic.json')
        lib_sha = library_sha16()
        if os.path.exists(tpath):
            with open(tpath) as fh:
                tj = json.load(fh)
            # a PMC measurement of THIS kernel on THIS workload by THIS library build only: same frame count, same kernel description,
            # same hash of the shared object (tools/profile_bench.sh records it) - anything else would be a stale figure
            # same hash of the shared object (tools/profile_bench.sh records it) or, for a rebuild of the same tree in another directory, the
            # same hash of the sources + compile flags - anything else would be a stale figure
            src_sha = sources_sha16()
            same_build = tj.get('lib_sha16') == lib_sha or (tj.get('src_sha16') is not None and tj.get('src_sha16') == src_sha)
            if tj.get('frames') == frames and tj.get('kernel') == eng.describe() and same_build:
                traffic = tj.get('hbm_bytes_per_launch')
                traffic_source = 'profiles/traffic.json: builder PMC run %s (rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE passes of this command, tools/profile_bench.sh), head %s, library %s (%s), sources %s' \
                                 % (tj.get('date', '?'), tj.get('head', '?'), lib_sha, 'the measured binary' if tj.get('lib_sha16') == lib_sha
                                    else 'a rebuild of the measured sources; measured: %s' % tj.get('lib_sha16'), src_sha)
            else:
                traffic_source = 'none: profiles/traffic.json was measured on another build / workload (library %s, sources %s there; %s, %s here)' \
                                 % (tj.get('lib_sha16'), tj.get('src_sha16'), lib_sha, src_sha)
        res = {
            'metric': 'Mpixels/s demodulated (720x576 PAL, 2D comb)',
            'value': round(value, 1), 'unit': 'Mpixels/s', 'n_gpus': world, 'steps': args.steps, 'warmup': args.warmup,
            'ms_per_step': round(elapsed / args.steps * 1e3, 4), 'higher_is_better': True, 'scaling': 'weak',
            'vs_baseline': None, 'dtype': 'f32', 'data': 'synthetic',
            'config': {'workload': 'PAL-BG 2D comb (PalDModem) demodulate, 720x576, %d-frame synthetic stream per GPU: smoothed '
                                   'random RGB, PAL-encoded on the device by the library\'s own modulator; float32 planar in HBM'
                                   % frames,
                       'frames_per_gpu': frames, 'parallelism': 'frames sharded, one stream per GPU, no data-path collective',
                       'kernel': eng.describe()},
            'roofline': {'bound': 'hbm', 'achieved': round(achieved, 1), 'peak': HBM_PEAK_GBPS, 'unit': 'GB/s',
                         'frac': round(achieved / HBM_PEAK_GBPS, 4), 'traffic': traffic, 'traffic_source': traffic_source,
                         'kernel_ms': round(mean_kernel_ms, 4), 'algorithmic_bytes_per_launch': BYTES_PER_PIXEL * px_step,
                         'note': 'the HBM roofline is the one the metric names; the kernel is limited by the vector pipe and the '
                                 'power cap, see roofline_valu (DESIGN.md section 4)'},
            'roofline_valu': {'bound': 'valu', 'achieved': round(valu_tflops, 1), 'peak': VALU_PEAK_TFLOPS, 'unit': 'TFLOP/s',
                              'frac': round(valu_tflops / VALU_PEAK_TFLOPS, 4),
                              'sustained_peak': VALU_SUSTAINED_TFLOPS, 'frac_of_sustained': round(valu_tflops / VALU_SUSTAINED_TFLOPS, 4),
                              'fma_equivalents_per_pixel': FMA_EQ_PER_PIXEL,
                              # the power give-back: at the 2.4 GHz the peak is quoted at, this arithmetic alone would take valu_limited_ms;
                              # the clock the board held while the timed steps ran (sysfs, sampled from a thread) is beside it
                              'valu_limited_ms_at_2400MHz': round(2.0 * FMA_EQ_PER_PIXEL * px_step / (VALU_PEAK_TFLOPS * 1e12) * 1e3, 4),
                              'measured_kernel_ms': round(mean_kernel_ms, 4),
                              'clock_MHz_if_valu_bound': round(2400.0 * valu_tflops / VALU_PEAK_TFLOPS, 1),
                              'sclk_MHz_during_timed_steps': clock_stats,
                              'note': 'float32 FMA-equivalents of the PAL-D interior bodies per pixel '
                                      '(counted from the ISA of the interior bodies, tools/isa_fma_census.py, profiles/r04_headline_bound.txt) x 2 '
                                      'flop; sustained_peak = a bare v_fma_f32 loop under the 1400 W cap (profiles/r01_ubench_valu.txt)'},
            'check': {'max_rel_err': float('%.3g' % worst), 'allclose_violations': bad, 'tolerance': 1e-5,
                      'frames': picks if world == 1 else 'rank 0: %s, other ranks: their last frame' % picks,
                      'what': 'timed output vs float64 CPU oracle: max |out - ref| / max |ref| per plane, and the count of '
                              'samples outside numpy.allclose(rtol=1e-5, atol=1e-6)'},
        }
        if world > 1:
            res['rccl_ranks'] = rccl_ranks
            res['ms_per_step_ranks'] = rank_ms
            res['rank_devices'] = rank_devices
            res['distinct_devices'] = len(set((d['uuid'] or d['pci_bus_id']) for d in rank_devices))
            res['gather'] = gather
            if backend != 'nccl':
                res['rehearsal'] = 'backend %s, %d ranks on %d GPU(s): launch-path rehearsal, not a scaling measurement' \
                                   % (backend, world, n_dev)
        if world == 1 and args.cpu_sample != 0:
            n_cpu = args.cpu_sample if args.cpu_sample > 0 else max(32, min(256, 4 * host_threads()))
            n_cpu = min(n_cpu, frames)
            res['cpu_baseline'] = cpu_baseline(modem, comp[:n_cpu].cpu().numpy(), first_frame)
        if world == 1 and args.other_configs:
            del comp, out
            torch.cuda.empty_cache()
            try:
                res['other_configs'] = other_configs(torch, device, frames)
                worst = max([worst] + [c['check']['max_rel_err'] for c in res['other_configs']])
            except Exception as e:      # the headline line must not be lost to a failure in the additional configurations: say so instead
                res['other_configs'] = []
                res['other_configs_error'] = '%s: %s' % (type(e).__name__, e)
        print(json.dumps(res))
        if worst > 1e-5:
            sys.stdout.flush()
            raise SystemExit('bench.py: a timed output misses the oracle by %.3g (> 1e-5)' % worst)
        if res.get('other_configs_error'):       # the headline line is out; a broken additional configuration still fails the run
            sys.stdout.flush()
            raise SystemExit('bench.py: other_configs failed: %s' % res['other_configs_error'])
    if dist is not None:
        dist.barrier()
        dist.destroy_process_group()


if __name__ == '__main__':
    main()
