# -*- coding: utf-8 -*-
"""Pin the oracle of the nested stacks (oracle/cm_oracle_generic.py: comb.py:71-167 restated around the leaves' oracles) against vectors the
REFERENCE produced (tests/golden/nested_*.npz, tests/golden/make_golden_nested.py)."""
import glob
import os

import numpy
import pytest

import stacks
from oracle import cm_oracle_generic

TOL = 1e-11
DEMOD = sorted(os.path.basename(p)[len('nested_demod_'):-4] for p in glob.glob(os.path.join(stacks.GOLDEN, 'nested_demod_*.npz')))
MOD = sorted(os.path.basename(p)[len('nested_mod_'):-4] for p in glob.glob(os.path.join(stacks.GOLDEN, 'nested_mod_*.npz')))
ROWS = sorted(os.path.basename(p)[len('nested_rows_'):-4] for p in glob.glob(os.path.join(stacks.GOLDEN, 'nested_rows_*.npz')))


def test_the_reference_made_sets_are_all_there():
    assert len(DEMOD) == 8 and len(MOD) == 4 and len(ROWS) == 4


@pytest.mark.parametrize('name', DEMOD)
def test_nested_frames_demod(name):
    g = stacks.load('nested_demod_' + name)
    modem = stacks.make_nested(name, g['size'])
    for i, f in enumerate(g['frames']):
        out = cm_oracle_generic.demodulate_frames(modem, g['inp'][i][None], int(f))[0]
        assert stacks.rel_err(out, g['out'][i]) < TOL, (name, f)


@pytest.mark.parametrize('name', MOD)
def test_nested_frames_mod(name):
    g = stacks.load('nested_mod_' + name)
    modem = stacks.make_nested(name, g['size'])
    for i, f in enumerate(g['frames']):
        out = cm_oracle_generic.modulate_frames(modem, g['inp'][i][None], int(f))[0]
        assert stacks.rel_err(out, g['out'][i]) < TOL, (name, f)


@pytest.mark.parametrize('name', ROWS)
def test_nested_rows(name):
    """one oracle object fed an explicit (frame, line) sequence with a repeated line in it: the wrapper starts over while a stateful
    backend modulator (ColorAveragingModem, HueCorrectingNiirModem) sees its strip lines continue"""
    g = stacks.load('nested_rows_' + name)
    orc = cm_oracle_generic.make(stacks.make_nested(name, g['size']))
    for i, (f, y) in enumerate(g['seq']):
        out = numpy.stack(orc.demodulate(int(f), int(y), g['inp'][i].astype(numpy.float64)))
        assert stacks.rel_err(out, g['out'][i]) < TOL, (name, f, y)
    first = cm_oracle_generic.make(stacks.make_nested(name, g['size']))
    f, y = g['seq'][0]
    out = numpy.stack(first.demodulate_components(int(f), int(y), g['inp'][0].astype(numpy.float64), strip_chroma=False))
    assert stacks.rel_err(out, g['first_unstripped'][0]) < TOL


def test_nested_image_uint8():
    """image.py:27-84 around the nested oracle: the reference's own bytes (<= 1 LSB on < 0.1 % of the samples: float64 operation order)"""
    g = stacks.load('nested_image_simple_avg_pals')
    h, w = g['comp8'].shape
    modem = stacks.make_nested('simple_avg_pals', (w, h))
    as_bytes = lambda a: numpy.uint8(numpy.rint(255.0 * numpy.clip(a, 0.0, 1.0)))
    rgb = g['rgb8'].astype(numpy.float64).transpose(2, 0, 1) / 255.0
    comp = cm_oracle_generic.modulate_frames(modem, rgb[None], int(g['frame']))[0]
    d = numpy.abs(as_bytes(0.6 * comp + 0.2).astype(int) - g['comp8'].astype(int))
    assert d.max() <= 1 and (d > 0).mean() < 1e-3
    lvl = (5.0 * (g['comp8'].astype(numpy.float64) / 255.0) - 1.0) / 3.0
    back = cm_oracle_generic.demodulate_frames(modem, lvl[None], int(g['frame']))[0]
    d = numpy.abs(as_bytes(back).transpose(1, 2, 0).astype(int) - g['back8'].astype(int))
    assert d.max() <= 1 and (d > 0).mean() < 1e-3
