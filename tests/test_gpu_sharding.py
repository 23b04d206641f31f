# -*- coding: utf-8 -*-
"""The N > 1 path with the HIP engine as the compute function: two ranks (gloo rendezvous, both on the box's one GPU)
shard a batch with parallel.demodulate_frames_sharded, gather it, and the result equals the single-process HIP result
bit for bit (frames are independent and `first_frame` carries the phase).  Plus bench.py's own
plain `--gpus 2` launch path on a tiny workload."""
import json
import os
import socket
import subprocess
import sys

import numpy
import pytest

import stacks
from color_modem_amd import image, testing

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _worker(rank, world, port, stack, size, n_frames, first, result_path):
    here = os.path.dirname(os.path.abspath(__file__))
    sys.path.insert(0, os.path.dirname(here))
    sys.path.insert(0, here)
    import torch
    import torch.distributed as dist
    import stacks as st
    from color_modem_amd import image as im, parallel, testing as tg
    os.environ['MASTER_ADDR'] = '127.0.0.1'
    os.environ['MASTER_PORT'] = str(port)
    dist.init_process_group('gloo', rank=rank, world_size=world)
    torch.cuda.set_device(rank % torch.cuda.device_count())
    eng = im.ImageModem(st.make(stack, size))._engine()
    comp = torch.from_numpy(tg.synthetic_composite(n_frames, size[1], size[0], seed=77)).cuda()

    def demod(x, first_frame):
        return eng.demodulate_frames(x, first_frame).cpu()     # gloo gathers host tensors

    local = parallel.demodulate_frames_sharded(demod, comp, first_frame=first, gather=False)
    lo, hi = parallel.frame_range(n_frames, world, rank)
    assert local.shape[0] == hi - lo
    full = parallel.demodulate_frames_sharded(demod, comp, first_frame=first, gather=True)
    assert full.shape[0] == n_frames
    if rank == 0:
        numpy.save(result_path, full.numpy())
    dist.barrier()
    dist.destroy_process_group()


@pytest.mark.parametrize('stack,size,n_frames,first', [('pal_d', (720, 64), 5, 2), ('ntsc_comb_3d', (720, 30), 4, 1)])
def test_two_ranks_hip_path_equals_single_process(tmp_path, stack, size, n_frames, first):
    import torch.multiprocessing as mp
    with socket.socket() as s:
        s.bind(('127.0.0.1', 0))
        port = s.getsockname()[1]
    result = str(tmp_path / 'full.npy')
    mp.spawn(_worker, args=(2, port, stack, size, n_frames, first, result), nprocs=2, join=True)
    got = numpy.load(result)
    comp = testing.synthetic_composite(n_frames, size[1], size[0], seed=77)
    want = image.ImageModem(stacks.make(stack, size)).demodulate_frames(comp, first_frame=first)
    # the ranks' shares equal the single-process result to float32 resolution (bit for bit on batches large enough to run
    # unsegmented; these few frames run in row segments whose number depends on the batch size: cm_api.hip: segment_geometry)
    assert numpy.abs(got - want).max() < 2e-7 * numpy.abs(want).max()


def _bench(*args):
    env = dict(os.environ)
    env.pop('WORLD_SIZE', None)
    env.pop('RANK', None)
    env.pop('LOCAL_RANK', None)
    proc = subprocess.run([sys.executable, os.path.join(ROOT, 'bench.py')] + list(args), env=env, stdout=subprocess.PIPE,
                          stderr=subprocess.PIPE, universal_newlines=True, timeout=900)
    assert proc.returncode == 0, proc.stdout + proc.stderr
    lines = [ln for ln in proc.stdout.splitlines() if ln.startswith('{')]
    assert len(lines) == 1, proc.stdout
    return json.loads(lines[0])


def test_bench_plain_invocations():
    """`python bench.py` and plain `python bench.py --gpus 2` (self-launching; gloo rehearsal when the box has one GPU)
    each print one self-verified line."""
    one = _bench('--frames', '24', '--steps', '2', '--warmup', '1', '--cpu-sample', '4')
    assert one['n_gpus'] == 1 and one['check']['max_rel_err'] < 1e-5 and one['check']['allclose_violations'] == 0
    assert 'PAL-encoded' in one['config']['workload'] and one['cpu_baseline']['cores'] >= 1
    assert one['roofline']['bound'] == 'hbm' and one['roofline_valu']['bound'] == 'valu'
    two = _bench('--gpus', '2', '--frames', '24', '--steps', '2', '--warmup', '1', '--gather-frames', '3')
    assert two['n_gpus'] == 2 and two['check']['max_rel_err'] < 1e-5
    assert two['gather']['frames'] == 6 and two['gather']['own_share_intact'] and two['gather']['gather_ms'] > 0
    # the CPU baseline and the additional configurations belong to the one-GPU line: rank 0 of an N > 1 run does not spend its time there
    assert 'cpu_baseline' not in two and 'other_configs' not in two
    # which device every rank ran on, as the ranks themselves report it
    assert [d['rank'] for d in two['rank_devices']] == [0, 1] and all(d['name'] for d in two['rank_devices'])
    assert one['roofline']['traffic_source'] and 'sclk_MHz_during_timed_steps' in one['roofline_valu']
    import torch
    if torch.cuda.device_count() >= 2:
        assert two['rccl_ranks'] == 2 and 'rehearsal' not in two
    else:
        assert two['rccl_ranks'] is None and 'rehearsal' in two


def test_streams_on_several_devices_from_one_process():
    """parallel.demodulate_streams ("or one process with 8 streams"): the fallback for a node where torch.distributed.run is
    not usable.  Two streams on devices [0, 0] (the box has one GPU) equal two single calls bit for bit, frame numbers carried."""
    import torch
    from color_modem_amd import parallel
    eng = image.ImageModem(stacks.make('pal_d', (720, 48)), batch_invariant=True)._engine()
    a = testing.synthetic_composite(12, 48, 720, seed=5)
    b = testing.synthetic_composite(9, 48, 720, seed=6)
    devs = [0, min(1, torch.cuda.device_count() - 1)]
    outs = parallel.demodulate_streams(eng, [a, b], devs, first_frames=[3, 15])
    assert [o.device.index for o in outs] == devs and outs[0].shape == (12, 3, 48, 720) and outs[1].shape == (9, 3, 48, 720)
    assert numpy.array_equal(outs[0].cpu().numpy(), eng.demodulate_frames(a, 3))
    assert numpy.array_equal(outs[1].cpu().numpy(), eng.demodulate_frames(b, 15))
