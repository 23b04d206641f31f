# -*- coding: utf-8 -*-
"""The numpy restatement of the reference's MAC modem (oracle/cm_oracle_mac.py) against vectors the reference produced
(tests/golden/mac_*.npz, tests/golden/make_golden_mac.py).  CPU only."""
import os

import numpy
import pytest

from color_modem_amd import line
from oracle import cm_oracle_mac as om

GOLDEN = os.path.join(os.path.dirname(os.path.abspath(__file__)), 'golden')
STD = line.LineStandard.GERBER_625


def _load(name):
    return numpy.load(os.path.join(GOLDEN, name + '.npz'))


def test_resampling_fir_known_answers():
    h = om.firwin41()
    assert abs(h[20] - 0.5002587352980301) < 1e-15
    assert abs(h[21] - 0.3167003457) < 1e-9 and abs(h[39] + 0.0010514588) < 1e-9
    assert numpy.abs(h[0:20:2]).max() < 1e-16 and abs(h.sum() - 1.0) < 1e-15
    x = numpy.arange(8.0)
    assert abs(om.resample_up2(x)[6] - 2 * h[20] * 3.0) < 1e-15       # even outputs are the input times 2 h[20]


@pytest.mark.parametrize('name,averaging', [('mac_mod_plain', False), ('mac_mod_avg', True), ('mac_mod_avg_h7', True)])
def test_modulate_frames_golden(name, averaging):
    z = _load(name)
    lc = line.LineConfig((720, int(z['height'])), STD)
    for i, f in enumerate(z['frames']):
        got = om.modulate_frames(lc, z['inp'][i:i + 1].astype(numpy.float64), int(f), averaging)[0]
        assert numpy.abs(got - z['out'][i]).max() < 1e-12


@pytest.mark.parametrize('name', ['mac_demod_plain', 'mac_demod_noise'])
def test_demodulate_frames_golden(name):
    z = _load(name)
    lc = line.LineConfig((720, int(z['height'])), STD)
    for i, f in enumerate(z['frames']):
        got = om.demodulate_frames(lc, z['inp'][i:i + 1].astype(numpy.float64), int(f))[0]
        assert numpy.abs(got - z['out'][i]).max() < 1e-12


@pytest.mark.parametrize('name', ['w768_7mhz', 'w768_7mhz_avg', 'w640_12mhz', 'w720_7mhz', 'w1000_900'])
def test_resampled_golden(name):
    z = _load('mac_resampled_' + name)
    H, W, cw, avg = int(z['height']), int(z['width']), int(z['line_width']), bool(z['averaging'])
    lc = line.LineConfig((W, H), STD)
    for i, f in enumerate(z['frames']):
        got = om.modulate_frames(lc, z['rgb'][i:i + 1].astype(numpy.float64), int(f), avg, cw)[0]
        assert numpy.abs(got - z['comp'][i]).max() < 1e-12
        back = om.demodulate_frames(lc, z['comp'][i:i + 1].astype(numpy.float32).astype(numpy.float64), int(f))[0]
        assert numpy.abs(back - z['back'][i]).max() < 1e-12


def test_image_round_trip_golden():
    """uint8 through the oracle with ImageModem's level mapping against the reference's own ImageModem."""
    from color_modem_amd.image import ImageModem, _as_bytes
    z = _load('mac_image_avg')
    rgb8, frame = z['rgb8'], int(z['frame'])
    H = rgb8.shape[0]
    lc = line.LineConfig((720, H), STD)
    rgb = (rgb8.astype(numpy.float64) / 255.0).transpose(2, 0, 1)[None]
    comp = om.modulate_frames(lc, rgb, frame, averaging=True)
    comp8 = _as_bytes(ImageModem.encode_composite_level(comp[0]))
    assert numpy.array_equal(comp8, z['comp8'])
    back = om.demodulate_frames(lc, ImageModem.decode_composite_level(comp8.astype(numpy.float64) / 255.0)[None], frame)
    assert numpy.array_equal(_as_bytes(back[0]).transpose(1, 2, 0), z['back8'])
