# -*- coding: utf-8 -*-
"""Frame sharding across the GPUs of one node (one process per GPU, torch.distributed; nccl = RCCL).

Frames are independent units of work - every stateful class of the reference resets at a frame
change, including the "3D" combs - so the data path needs no exchange step: rank r
demodulates a contiguous frame range with `first_frame` advanced accordingly, and results stay
sharded in each GPU's HBM.  A gather over xGMI is offered for callers that want the whole batch on
every rank; it is not part of the hot path and is timed on its own by bench.py (one root's inbound
links are slower than one GPU's output rate).
"""


def frame_range(n_frames, world_size, rank):
    """Contiguous, balanced split: the first (n_frames % world_size) ranks take one extra frame."""
    if world_size < 1 or not 0 <= rank < world_size:
        raise ValueError('bad rank %r of %r' % (rank, world_size))
    base, extra = divmod(int(n_frames), world_size)
    lo = rank * base + min(rank, extra)
    return lo, lo + base + (1 if rank < extra else 0)


def _world(group):
    import torch.distributed as dist
    if dist.is_available() and dist.is_initialized():
        return dist.get_world_size(group), dist.get_rank(group)
    return 1, 0


def gather_frames(local, n_frames, group=None):
    """all_gather of the per-rank results of a batch of `n_frames` frames split by frame_range: every rank receives
    the complete [n_frames, ...] tensor.  One collective (ranks may differ by one frame: shares are padded to the
    largest one)."""
    import torch
    import torch.distributed as dist
    world, rank = _world(group)
    if world == 1:
        return local
    lo, hi = frame_range(n_frames, world, rank)
    if local.shape[0] != hi - lo:
        raise ValueError('rank %d holds %d frames, its share of %d is %d' % (rank, local.shape[0], n_frames, hi - lo))
    per = -(-n_frames // world)
    if (hi - lo) == per:
        padded = local.contiguous()
    else:
        padded = torch.zeros((per,) + tuple(local.shape[1:]), dtype=local.dtype, device=local.device)
        padded[:hi - lo] = local
    full = torch.empty((world * per,) + tuple(local.shape[1:]), dtype=local.dtype, device=local.device)
    dist.all_gather_into_tensor(full, padded, group=group)
    if n_frames == world * per:
        return full
    parts = []
    for r in range(world):
        rlo, rhi = frame_range(n_frames, world, r)
        parts.append(full[r * per:r * per + (rhi - rlo)])
    return torch.cat(parts, dim=0)


def demodulate_frames_sharded(demodulate, composite, first_frame=0, group=None, gather=False):
    """Run `demodulate(composite[lo:hi], first_frame + lo)` on this rank's share of a batch that every
    rank holds (or can address) in full.  Returns the local rgb[hi-lo, 3, H, W], or with gather=True
    the complete rgb[F, 3, H, W] assembled with all_gather (torch tensors only)."""
    import torch
    world, rank = _world(group)
    n = composite.shape[0]
    lo, hi = frame_range(n, world, rank)
    local = demodulate(composite[lo:hi], first_frame + lo)
    if not gather or world == 1:
        return local
    if not torch.is_tensor(local):
        local = torch.as_tensor(local)
    return gather_frames(local, n, group)


def demodulate_streams(engine, streams, devices, first_frames=None):
    """ONE process, several devices ("one process per GPU or one process with 8 streams"): stream i - composite
    [F_i, H, W] float32, a numpy array or a tensor anywhere - is demodulated on devices[i] under the frame numbers first_frames[i] ..
    The launches of all streams are enqueued before any result is waited for (each on its device's current stream), so the GPUs run
    side by side; the engine keeps one plan per device (engine._DevicePlans).  Returns the rgb [F_i, 3, H, W] tensors, each left on
    its device.  A fallback for nodes where torch.distributed.run cannot be used - bench.py's ranks are the measured path."""
    import torch
    if len(streams) != len(devices):
        raise ValueError('%d streams for %d devices' % (len(streams), len(devices)))
    first_frames = [0] * len(streams) if first_frames is None else list(first_frames)
    outs = []
    for comp, dev, first in zip(streams, devices, first_frames):
        device = torch.device('cuda', int(dev)) if not isinstance(dev, torch.device) else dev
        t = torch.as_tensor(comp)
        if t.dtype != torch.float32:
            raise ValueError('streams are float32 [F, H, W]')
        t = t.to(device, non_blocking=True).contiguous()
        with torch.cuda.device(device):
            outs.append(engine.demodulate_frames(t, int(first)))      # asynchronous: the launch returns as soon as it is enqueued
    for dev in set(int(getattr(d, 'index', d)) for d in devices):
        torch.cuda.synchronize(dev)
    return outs
