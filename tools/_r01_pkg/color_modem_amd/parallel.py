# -*- coding: utf-8 -*-
"""Frame sharding across the GPUs of one node (one process per GPU, torch.distributed; nccl = RCCL).

Frames are independent units of work - every stateful class of the reference resets at a frame
change, including the "3D" combs - so the data path needs no exchange step: rank r
demodulates a contiguous frame range with `first_frame` advanced accordingly, and results stay
sharded in each GPU's HBM.  A gather over xGMI is offered for callers that want the whole batch on
every rank; it is not part of the timed hot path (one root's inbound links are slower than one
GPU's output rate).
"""


def frame_range(n_frames, world_size, rank):
    """Contiguous, balanced split: the first (n_frames % world_size) ranks take one extra frame."""
    if world_size < 1 or not 0 <= rank < world_size:
        raise ValueError('bad rank %r of %r' % (rank, world_size))
    base, extra = divmod(int(n_frames), world_size)
    lo = rank * base + min(rank, extra)
    return lo, lo + base + (1 if rank < extra else 0)


def demodulate_frames_sharded(demodulate, composite, first_frame=0, group=None, gather=False):
    """Run `demodulate(composite[lo:hi], first_frame + lo)` on this rank's share of a batch that every
    rank holds (or can address) in full.  Returns the local rgb[hi-lo, 3, H, W], or with gather=True
    the complete rgb[F, 3, H, W] assembled with all_gather (torch tensors only)."""
    import torch
    import torch.distributed as dist
    world = dist.get_world_size(group) if dist.is_initialized() else 1
    rank = dist.get_rank(group) if dist.is_initialized() else 0
    n = composite.shape[0]
    lo, hi = frame_range(n, world, rank)
    local = demodulate(composite[lo:hi], first_frame + lo)
    if not gather or world == 1:
        return local
    if not torch.is_tensor(local):
        local = torch.as_tensor(local)
    per = -(-n // world)  # ranks may differ by one frame: pad to the largest share
    padded = torch.zeros((per,) + tuple(local.shape[1:]), dtype=local.dtype, device=local.device)
    padded[:hi - lo] = local
    parts = [torch.empty_like(padded) for _ in range(world)]
    dist.all_gather(parts, padded, group=group)
    out = []
    for r in range(world):
        rlo, rhi = frame_range(n, world, r)
        out.append(parts[r][:rhi - rlo])
    return torch.cat(out, dim=0)
