# -*- coding: utf-8 -*-
"""GPU parity of the MAC path (cm_mac_* behind MacModem / ColorAveragingModem(MacModem) / ImageModem) against the
reference-generated goldens (tests/golden/mac_*.npz) and the float64 oracle (oracle/cm_oracle_mac.py).
Tolerance: max|out - ref| <= 1e-5 * max|ref| per frame, uint8 images within 1 LSB."""
import numpy
import pytest

import stacks
from color_modem_amd import comb, image, line, testing
from color_modem_amd.color import mac
from oracle import cm_oracle_mac as om

pytestmark = pytest.mark.gpu
TOL = 1e-5
STD = line.LineStandard.GERBER_625


def make(height, averaging=False, std=STD):
    m = mac.MacModem(line.LineConfig((720, height), std))
    return comb.ColorAveragingModem(m) if averaging else m


@pytest.mark.parametrize('name,averaging', [('mac_mod_plain', False), ('mac_mod_avg', True), ('mac_mod_avg_h7', True)])
def test_modulate_frames_golden(name, averaging):
    g = stacks.load(name)
    im = image.ImageModem(make(int(g['height']), averaging))
    for i, f in enumerate(g['frames']):
        out = im.modulate_frames(g['inp'][i:i + 1], first_frame=int(f))[0]
        assert out.dtype == numpy.float32 and out.shape == (int(g['height']), 1080)
        assert stacks.rel_err(out, g['out'][i]) < TOL, (name, int(f))


@pytest.mark.parametrize('name', ['mac_demod_plain', 'mac_demod_noise'])
def test_demodulate_frames_golden(name):
    g = stacks.load(name)
    im = image.ImageModem(make(int(g['height'])))
    for i, f in enumerate(g['frames']):
        out = im.demodulate_frames(g['inp'][i:i + 1], first_frame=int(f))[0]
        assert out.shape == (3, int(g['height']), 720)
        assert stacks.rel_err(out, g['out'][i]) < TOL, (name, int(f))


@pytest.mark.parametrize('height,std_name', [(576, 'GERBER_625'), (480, 'NTSC_525'), (37, 'GERBER_625'), (1, 'GERBER_625')])
@pytest.mark.parametrize('averaging', [False, True])
def test_frames_against_oracle(height, std_name, averaging):
    std = getattr(line.LineStandard, std_name)
    lc = line.LineConfig((720, height), std)
    modem = make(height, averaging, std)
    im = image.ImageModem(modem)
    if averaging and height < 2:     # the reference's row schedule reads row 1 (image.py:49-50): IndexError
        with pytest.raises(IndexError):
            im.modulate_frames(testing.synthetic_rgb(1, height, 720), first_frame=0)
        return
    n = 2 if height > 100 else 5
    rgb = testing.synthetic_rgb(n, height, 720, seed=31 + height)
    want = om.modulate_frames(lc, rgb.astype(numpy.float64), 3, averaging)
    comp = im.modulate_frames(rgb, first_frame=3)
    for i in range(n):
        assert stacks.rel_err(comp[i], want[i]) < TOL
    comp32 = want.astype(numpy.float32)
    back = im.demodulate_frames(comp32, first_frame=3)
    want_back = om.demodulate_frames(lc, comp32.astype(numpy.float64), 3)
    for i in range(n):
        assert stacks.rel_err(back[i], want_back[i]) < TOL


def test_row_protocol_against_oracle():
    """modulate() / demodulate() one row per call, with a reset in the middle (frame change) and a line jump."""
    height = 12
    lc = line.LineConfig((720, height), STD)
    rgb = testing.synthetic_rgb(2, height, 720, seed=5)
    for averaging in (False, True):
        dev, ref = make(height, averaging), om.OracleMac(lc, averaging)
        calls = [(0, 0), (0, 2), (0, 4), (0, 8), (0, 10), (1, 1), (1, 3), (1, 5)]
        for frame, ln in calls:
            r, g, b = (rgb[frame, p, ln].astype(numpy.float64) for p in range(3))
            got, want = dev.modulate(frame, ln, r, g, b), ref.modulate(frame, ln, r, g, b)
            assert got.shape == (1080,) and stacks.rel_err(got, want) < TOL, (averaging, frame, ln)
            comp = want.astype(numpy.float32)
            got3, want3 = dev.demodulate(frame, ln, comp), ref.demodulate(frame, ln, comp.astype(numpy.float64))
            for p in range(3):
                assert numpy.abs(got3[p] - want3[p]).max() < TOL * max(1.0, numpy.abs(numpy.stack(want3)).max()), (averaging, frame, ln, p)


def test_components_protocol():
    lc = line.LineConfig((720, 8), STD)
    modem = make(8)
    rgb = testing.synthetic_rgb(1, 8, 720, seed=9)[0].astype(numpy.float64)
    y, dr, db = mac.MacModem.encode_components(rgb[0, 2], rgb[1, 2], rgb[2, 2])
    got = modem.modulate_components(0, 2, y, dr, db)
    want = om.OracleMac(lc).modulate_components(0, 2, y, dr, db)
    assert stacks.rel_err(got, want) < TOL
    with pytest.raises(AttributeError):
        modem.demodulate_components(0, 2, want)


def test_pil_image_round_trip_golden():
    from PIL import Image
    g = stacks.load('mac_image_avg')
    rgb8, frame = g['rgb8'], int(g['frame'])
    H = rgb8.shape[0]
    im = image.ImageModem(make(H, averaging=True))
    comp_img = im.modulate(Image.frombytes('RGB', (720, H), rgb8.tobytes()), frame)
    assert comp_img.size == (1080, H) and comp_img.mode == 'L'
    comp8 = numpy.frombuffer(comp_img.tobytes(), dtype=numpy.uint8).reshape(H, 1080)
    d = numpy.abs(comp8.astype(int) - g['comp8'].astype(int))
    assert d.max() <= 1 and (d > 0).mean() < 2e-3
    back = im.demodulate(Image.frombytes('L', (1080, H), g['comp8'].tobytes()), frame)
    assert back.size == (720, H) and back.mode == 'RGB'
    back8 = numpy.frombuffer(back.tobytes(), dtype=numpy.uint8).reshape(H, 720, 3)
    d = numpy.abs(back8.astype(int) - g['back8'].astype(int))
    assert d.max() <= 1 and (d > 0).mean() < 2e-3


def test_full_size_properties():
    """576-line frames: batches are independent of how they are cut; the parity of the frame number selects the line
    alternation (period 2); a flat field stays flat; encode -> decode returns the luma exactly where the line carries it."""
    import torch
    height = 576
    im = image.ImageModem(make(height))
    rgb = torch.from_numpy(testing.synthetic_rgb(4, height, 720, seed=77)).cuda()
    comp = im.modulate_frames(rgb, first_frame=10)
    assert torch.equal(comp[2:], im.modulate_frames(rgb[2:], first_frame=12))
    assert torch.equal(im.modulate_frames(rgb[:1], first_frame=10), im.modulate_frames(rgb[:1], first_frame=12))
    back = im.demodulate_frames(comp, first_frame=10)
    assert torch.equal(back[1:3], im.demodulate_frames(comp[1:3], first_frame=11))
    flat = torch.full((1, 3, height, 720), 0.25, device='cuda')
    bf = im.demodulate_frames(im.modulate_frames(flat, 0), 0)
    inner = bf[0, :, 2:, 40:680]          # away from the first line of each field (no previous chroma) and the row ends
    assert float((inner - 0.25).abs().max()) < 2e-3
    y = (0.299 * rgb[:, 0] + 0.587 * rgb[:, 1] + 0.114 * rgb[:, 2])
    assert float((comp[:, :, 372:1071] - y[:, :, 11:710]).abs().max()) < 1e-6


RESAMPLED = ['w768_7mhz', 'w768_7mhz_avg', 'w640_12mhz', 'w720_7mhz', 'w1000_900']


@pytest.mark.parametrize('name', RESAMPLED)
def test_resampled_golden(name):
    """Rows of other lengths / the 720-sample D2MAC_7MHZ line / an arbitrary line length (mac.py:49-55, 71-74, 88-91)."""
    g = stacks.load('mac_resampled_' + name)
    H, W, cw, avg = int(g['height']), int(g['width']), int(g['line_width']), bool(g['averaging'])
    lc = line.LineConfig((W, H), STD)
    variant = {1080: mac.MacVariant.D2MAC_12MHZ, 720: mac.MacVariant.D2MAC_7MHZ}.get(cw, cw)
    enc = mac.MacModem(lc, variant)
    im_enc = image.ImageModem(comb.ColorAveragingModem(enc) if avg else enc)
    im_dec = image.ImageModem(mac.MacModem(lc, variant))
    for i, f in enumerate(g['frames']):
        comp = im_enc.modulate_frames(g['rgb'][i:i + 1], first_frame=int(f))[0]
        assert comp.shape == (H, cw) and stacks.rel_err(comp, g['comp'][i]) < TOL, (name, int(f))
        back = im_dec.demodulate_frames(g['comp'][i:i + 1].astype(numpy.float32), first_frame=int(f))[0]
        assert back.shape == (3, H, 720) and stacks.rel_err(back, g['back'][i]) < TOL, (name, int(f))


@pytest.mark.parametrize('width,line_width', [(720, 720), (1920, 1080), (479, 1081), (704, 720),
                                              (2560, 1080), (3840, 5000), (720, 14000)])      # round 6: beyond 64 KiB of LDS per workgroup
def test_resampled_against_oracle(width, line_width):
    H = 21
    lc = line.LineConfig((width, H), STD)
    enc = comb.ColorAveragingModem(mac.MacModem(lc, line_width))
    rgb = testing.synthetic_rgb(2, H, width, seed=width)
    want = om.modulate_frames(lc, rgb.astype(numpy.float64), 1, True, line_width)
    comp = image.ImageModem(enc).modulate_frames(rgb, first_frame=1)
    comp32 = want.astype(numpy.float32)
    back = image.ImageModem(mac.MacModem(lc, line_width)).demodulate_frames(comp32, first_frame=1)
    want_back = om.demodulate_frames(lc, comp32.astype(numpy.float64), 1)
    for i in range(2):
        assert stacks.rel_err(comp[i], want[i]) < TOL and stacks.rel_err(back[i], want_back[i]) < TOL
    # the per-row protocol on the resampling kernels
    dev, ref = mac.MacModem(lc, line_width), om.OracleMac(lc, False, line_width)
    for ln in (1, 3, 5):
        r, g, b = (rgb[0, p, ln].astype(numpy.float64) for p in range(3))
        got, exp = dev.modulate(4, ln, r, g, b), ref.modulate(4, ln, r, g, b)
        assert got.shape == (line_width,) and stacks.rel_err(got, exp) < TOL
        got3, exp3 = dev.demodulate(4, ln, exp.astype(numpy.float32)), ref.demodulate(4, ln, exp.astype(numpy.float32).astype(numpy.float64))
        assert stacks.rel_err(numpy.stack(got3), numpy.stack(exp3)) < TOL


@pytest.mark.parametrize('width,line_width,averaging', [(720, 1080, False), (720, 1080, True), (768, 720, True), (1000, 900, False)])
def test_fused_byte_boundary(width, line_width, averaging):
    """cm_mac_*_frames_u8 against the float kernels with ImageModem's conversions on the host (image.py:7-8, 20-25, 43-45)."""
    from color_modem_amd.image import _as_bytes
    H = 14
    lc = line.LineConfig((width, H), STD)
    enc = mac.MacModem(lc, line_width)
    im = image.ImageModem(comb.ColorAveragingModem(enc) if averaging else enc)
    rgb8 = _as_bytes(testing.synthetic_rgb(3, H, width, seed=width + 1).astype(numpy.float64)).transpose(0, 2, 3, 1).copy()
    comp8 = im.modulate_frames_u8(rgb8, first_frame=5)
    assert comp8.dtype == numpy.uint8 and comp8.shape == (3, H, line_width)
    rgbf = (rgb8.astype(numpy.float64) / 255.0).astype(numpy.float32).transpose(0, 3, 1, 2)
    want8 = _as_bytes(image.ImageModem.encode_composite_level(im.modulate_frames(numpy.ascontiguousarray(rgbf), first_frame=5).astype(numpy.float64)))
    d = numpy.abs(comp8.astype(int) - want8.astype(int))
    assert d.max() <= 1 and (d > 0).mean() < 2e-3
    back8 = im.demodulate_frames_u8(want8, first_frame=5)
    assert back8.shape == (3, H, 720, 3)
    comp = image.ImageModem.decode_composite_level(want8.astype(numpy.float64) / 255.0).astype(numpy.float32)
    wantb = _as_bytes(im.demodulate_frames(comp, first_frame=5).astype(numpy.float64)).transpose(0, 2, 3, 1)
    d = numpy.abs(back8.astype(int) - wantb.astype(int))
    assert d.max() <= 1 and (d > 0).mean() < 2e-3


def test_limits_fail_loudly():
    with pytest.raises(NotImplementedError):       # a call's rows live in one CU's LDS: 4096 samples per row, 16384 per line
        mac.MacModem(line.LineConfig((4100, 8), STD))
    with pytest.raises(NotImplementedError):
        mac.MacModem(line.LineConfig((720, 8), STD), 20000)
    im = image.ImageModem(make(8))
    with pytest.raises(ValueError):
        im.demodulate_frames(numpy.zeros((1, 8, 720), dtype=numpy.float32))
    assert im.demodulate_frames(numpy.zeros((0, 8, 1080), dtype=numpy.float32)).shape == (0, 3, 8, 720)
