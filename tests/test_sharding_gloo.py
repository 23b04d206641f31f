# -*- coding: utf-8 -*-
"""The N > 1 path on CPU: two gloo ranks shard a batch of frames, each demodulates its range (here with
the oracle standing in for the device engine), and the gathered result equals the single-process one."""
import os
import socket

import numpy
import pytest

from color_modem_amd import parallel


def test_frame_range_partitions():
    for n in (0, 1, 7, 8, 1000):
        for world in (1, 2, 3, 8):
            spans = [parallel.frame_range(n, world, r) for r in range(world)]
            assert spans[0][0] == 0 and spans[-1][1] == n
            assert all(a[1] == b[0] for a, b in zip(spans, spans[1:]))
            sizes = [hi - lo for lo, hi in spans]
            assert max(sizes) - min(sizes) <= 1
    with pytest.raises(ValueError):
        parallel.frame_range(4, 2, 2)


def _worker(rank, world, port, result_path, n_frames=5):
    import sys
    here = os.path.dirname(os.path.abspath(__file__))
    sys.path.insert(0, os.path.dirname(here))
    sys.path.insert(0, here)
    import torch
    import torch.distributed as dist
    import stacks
    from color_modem_amd import testing
    from oracle import cm_oracle
    os.environ['MASTER_ADDR'] = '127.0.0.1'
    os.environ['MASTER_PORT'] = str(port)
    dist.init_process_group('gloo', rank=rank, world_size=world)
    modem = stacks.make('pal_d', (720, 8))
    comp = testing.synthetic_composite(n_frames, 8, 720, seed=77)

    def demod(x, first):
        return torch.from_numpy(cm_oracle.demodulate_frames_f32(modem, numpy.asarray(x), first_frame=first))

    local = parallel.demodulate_frames_sharded(demod, torch.from_numpy(comp), first_frame=2, gather=False)
    lo, hi = parallel.frame_range(n_frames, world, rank)
    assert local.shape[0] == hi - lo
    full = parallel.demodulate_frames_sharded(demod, torch.from_numpy(comp), first_frame=2, gather=True)
    if rank == 0:
        numpy.save(result_path, full.numpy())
    dist.barrier()
    dist.destroy_process_group()


def test_gather_frames_uneven_shares():
    """one process: gather_frames is the identity; its share check refuses a tensor of the wrong length"""
    import torch
    x = torch.arange(12.0).reshape(3, 4)
    assert parallel.gather_frames(x, 3) is x


def test_two_rank_gloo_sharding(tmp_path):
    import torch.multiprocessing as mp
    import stacks
    from color_modem_amd import testing
    from oracle import cm_oracle
    with socket.socket() as s:
        s.bind(('127.0.0.1', 0))
        port = s.getsockname()[1]
    result = str(tmp_path / 'full.npy')
    mp.spawn(_worker, args=(2, port, result), nprocs=2, join=True)
    got = numpy.load(result)
    modem = stacks.make('pal_d', (720, 8))
    comp = testing.synthetic_composite(5, 8, 720, seed=77)
    want = cm_oracle.demodulate_frames_f32(modem, comp, first_frame=2)
    assert numpy.array_equal(got, want)


def test_eight_rank_gloo_sharding(tmp_path):
    """BASELINE configs[4] is eight ranks: the same path at world size 8 (uneven shares: 19 frames), CPU, the oracle as the compute
    function - frame ranges, first_frame offsets and the padded all_gather of parallel.gather_frames at the driver's rank count."""
    import torch.multiprocessing as mp
    import stacks
    from color_modem_amd import testing
    from oracle import cm_oracle
    with socket.socket() as s:
        s.bind(('127.0.0.1', 0))
        port = s.getsockname()[1]
    result = str(tmp_path / 'full8.npy')
    mp.spawn(_worker, args=(8, port, result, 19), nprocs=8, join=True)
    got = numpy.load(result)
    modem = stacks.make('pal_d', (720, 8))
    comp = testing.synthetic_composite(19, 8, 720, seed=77)
    want = cm_oracle.demodulate_frames_f32(modem, comp, first_frame=2)
    assert got.shape == want.shape and numpy.array_equal(got, want)
