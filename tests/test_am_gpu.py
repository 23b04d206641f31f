# -*- coding: utf-8 -*-
"""GPU parity of the amplitude-modulated line-sequential standards (Proto-SECAM 1957, NIIR / SECAM-IV) through the C ABI
(cm_am_*): reference-generated goldens (tests/golden/am_*.npz) and the numpy oracle on seeded inputs.  Tolerance:
max |out - ref| <= 1e-5 max |ref| per frame, as for the other standards."""
import numpy
import pytest

import am_stacks
import stacks
from color_modem_amd import image, line, testing

pytestmark = pytest.mark.gpu
TOL = 1e-5

PROTO = ['proto', 'proto_avg', 'proto_nofilter', 'proto_625']


@pytest.mark.parametrize('stack', PROTO)
def test_modulate_frames_golden(stack):
    z = am_stacks.load('am_mod_' + stack)
    im = image.ImageModem(am_stacks.make(stack, z))
    for i, f in enumerate(z['frames']):
        out = im.modulate_frames(z['inp'][i:i + 1], first_frame=int(f))[0]
        assert stacks.rel_err(out, z['out'][i]) < TOL, (stack, int(f))


@pytest.mark.parametrize('stack', PROTO)
def test_demodulate_frames_golden(stack):
    z = am_stacks.load('am_demod_' + stack)
    im = image.ImageModem(am_stacks.make(am_stacks.DECODER_OF.get(stack, stack), z))
    for i, f in enumerate(z['frames']):
        out = im.demodulate_frames(z['inp'][i:i + 1], first_frame=int(f))[0]
        assert stacks.rel_err(out, z['out'][i]) < TOL, (stack, int(f))
    frames = [int(f) for f in z['frames']]
    if frames == list(range(frames[0], frames[0] + len(frames))):        # consecutive frames as one batch
        out = im.demodulate_frames(z['inp'], first_frame=frames[0])
        for i in range(len(frames)):
            assert stacks.rel_err(out[i], z['out'][i]) < TOL


@pytest.mark.parametrize('stack', ['proto'])
def test_row_sequences_golden(stack):
    """The stateful per-row protocol with a break in the run and a frame change, at full-height line numbers."""
    z = am_stacks.load('am_rows_' + stack)
    modem = am_stacks.make(stack, z)
    for i, (f, y) in enumerate(z['seq']):
        out = numpy.stack(modem.demodulate(int(f), int(y), z['inp'][i]))
        assert stacks.rel_err(out, z['out'][i]) < TOL, (stack, int(f), int(y))


@pytest.mark.parametrize('stack,size,std,first', [('proto', (720, 64), 'FRENCH_819', 2), ('proto_avg', (720, 33), 'BELGIAN_819', 1),
                                                 ('proto', (1000, 9), 'FRENCH_819', 0), ('proto', (718, 12), 'FRENCH_819', 5)])
def test_proto_round_trip_vs_oracle(stack, size, std, first):
    """encode and decode on the device against the oracle's round trip, on frame sizes the goldens do not cover (many
    workgroups per field, a width that is not a multiple of 4, an odd height)."""
    from oracle import cm_oracle_am as oa
    import test_am_oracle
    lc = line.LineConfig(size, getattr(line.LineStandard, std))
    modem = am_stacks.STACKS[stack](lc)
    inner = modem.backend if stack == 'proto_avg' else modem
    rgb = testing.synthetic_rgb(2, size[1], size[0], seed=77 + size[1])
    im = image.ImageModem(modem)
    comp = im.modulate_frames(rgb, first_frame=first)
    if stack == 'proto_avg':
        comp_ref = test_am_oracle._averaging_frames(modem, rgb.astype(numpy.float64), first)
    else:
        comp_ref = oa.modulate_frames(modem, rgb.astype(numpy.float64), first)
    assert stacks.rel_err(comp, comp_ref) < TOL
    comp32 = comp_ref.astype(numpy.float32)
    back = image.ImageModem(inner).demodulate_frames(comp32, first_frame=first)
    back_ref = oa.demodulate_frames(inner, comp32.astype(numpy.float64), first)
    for i in range(2):
        assert stacks.rel_err(back[i], back_ref[i]) < TOL, (stack, i)
