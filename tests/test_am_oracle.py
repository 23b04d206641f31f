# -*- coding: utf-8 -*-
"""The numpy restatement of the reference's Proto-SECAM and NIIR modems (oracle/cm_oracle_am.py) against vectors the
reference produced (tests/golden/am_*.npz, tests/golden/make_golden_am.py).  CPU only."""
import numpy
import pytest

import am_stacks
from oracle import cm_oracle_am as oa

MOD = ['proto', 'proto_avg', 'proto_nofilter', 'proto_625', 'niir', 'niir_hue', 'niir_525']


def _averaging_frames(modem, rgb, first_frame):
    """ColorAveragingModem(ProtoSecamModem) (comb.py:130-167) on top of the oracle object"""
    from oracle import cm_oracle_am
    n, _, height, width = rgb.shape
    out = numpy.zeros((n, height, width))
    inner = modem.backend
    for i in range(n):
        orc = cm_oracle_am.make(inner)
        frame = first_frame + i
        state = {'f': -1, 'l': -1, 'y': None, 'u': None, 'v': None}

        def modulate(fr, ln, r, g, b):
            y, u, v = inner.encode_components(r, g, b)
            if fr != state['f'] or ln != state['l'] + 2 or state['u'] is None:
                state['y'], state['u'], state['v'] = y, u, v
            state['y'], y = y, state['y']
            state['u'], u = u, 0.5 * (u + state['u'])
            state['v'], v = v, 0.5 * (v + state['v'])
            state['f'], state['l'] = fr, ln
            return orc.modulate_components(fr, ln - 2, y, u, v)
        for field in range(2):
            for y in range(field, 2, 2):
                modulate(frame, y, rgb[i, 0, y], rgb[i, 1, y], rgb[i, 2, y])
            for y in range(field, height, 2):
                iy = y + 2
                while iy >= height:
                    iy -= 2
                out[i, y] = modulate(frame, y + 2, rgb[i, 0, iy], rgb[i, 1, iy], rgb[i, 2, iy])
    return out


@pytest.mark.parametrize('stack', MOD + ['niir_grey', 'niir_hue_grey'])
def test_modulate_frames_golden(stack):
    z = am_stacks.load('am_mod_' + stack)
    modem = am_stacks.make(stack, z)
    for i, f in enumerate(z['frames']):
        rgb = z['inp'][i:i + 1].astype(numpy.float64)
        got = _averaging_frames(modem, rgb, int(f))[0] if stack == 'proto_avg' else oa.modulate_frames(modem, rgb, int(f))[0]
        assert numpy.abs(got - z['out'][i]).max() < 1e-11, (stack, int(f))


@pytest.mark.parametrize('stack', ['niir_noise', 'niir_hue_noise'])
def test_modulate_frames_with_noise_golden(stack):
    """NiirModem(noise_level != 0): the reference draws numpy.random.random_sample for db, then dr, per modulate() call
    (niir.py:45-46, 193-194); the goldens were made under numpy.random.seed(seeds[i]) per frame."""
    z = am_stacks.load('am_mod_' + stack)
    modem = am_stacks.make(stack, z)
    for i, f in enumerate(z['frames']):
        numpy.random.seed(int(z['seeds'][i]))
        got = oa.modulate_frames(modem, z['inp'][i:i + 1].astype(numpy.float64), int(f))[0]
        assert numpy.abs(got - z['out'][i]).max() < 1e-11, (stack, int(f))


@pytest.mark.parametrize('stack', MOD)
def test_demodulate_frames_golden(stack):
    z = am_stacks.load('am_demod_' + stack)
    modem = am_stacks.make(am_stacks.DECODER_OF.get(stack, stack), z)
    for i, f in enumerate(z['frames']):
        got = oa.demodulate_frames(modem, z['inp'][i:i + 1].astype(numpy.float64), int(f))[0]
        ref = z['out'][i]
        ok = numpy.isfinite(ref)
        assert numpy.array_equal(ok, numpy.isfinite(got)), stack
        assert numpy.abs(got[ok] - ref[ok]).max() < 1e-9 * max(1.0, numpy.abs(ref[ok]).max()), (stack, int(f))


@pytest.mark.parametrize('stack', ['proto', 'niir'])
def test_row_sequences_golden(stack):
    z = am_stacks.load('am_rows_' + stack)
    orc = oa.make(am_stacks.make(stack, z))
    for i, (f, y) in enumerate(z['seq']):
        got = numpy.stack(orc.demodulate(int(f), int(y), z['inp'][i].astype(numpy.float64)))
        ref = z['out'][i]
        ok = numpy.isfinite(ref)
        assert numpy.abs(got[ok] - ref[ok]).max() < 1e-9 * max(1.0, numpy.abs(ref[ok]).max()), (stack, int(f), int(y))


def test_niir_components_unstripped_noise():
    z = am_stacks.load('am_niir_components_noise')
    modem = am_stacks.make('niir', z)
    orc = oa.make(modem)
    k = 0
    for i, f in enumerate(z['frames']):
        for field in range(2):
            for y in range(field, 6, 2):
                got = numpy.stack(orc.demodulate_components(int(f), y, z['inp'][i, y].astype(numpy.float64), strip_chroma=False))
                ref = z['out'][k]
                ok = numpy.isfinite(ref)
                assert numpy.abs(got[ok] - ref[ok]).max() < 1e-8 * max(1.0, numpy.abs(ref[ok]).max())
                k += 1
