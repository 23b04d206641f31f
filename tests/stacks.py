# -*- coding: utf-8 -*-
"""Modem stacks named as in tests/golden/make_golden.py, built from color_modem_amd classes."""
import json
import os

import numpy

from color_modem_amd import comb, line
from color_modem_amd.color import ntsc, pal, secam

GOLDEN = os.path.join(os.path.dirname(os.path.abspath(__file__)), 'golden')

STACKS = {
    'pal_s': lambda lc: pal.PalSModem(lc),
    'pal_d': lambda lc: pal.PalDModem(lc),
    'pal_3d': lambda lc: pal.Pal3DModem(lc),
    'ntsc': lambda lc: ntsc.NtscModem(lc),
    'ntsc_comb': lambda lc: ntsc.NtscCombModem(lc),
    'ntsc_comb_simple': lambda lc: comb.SimpleCombModem(ntsc.NtscCombModem(lc)),
    'ntsc_comb_3d': lambda lc: comb.Simple3DCombModem(ntsc.NtscCombModem(lc)),
    'pal_avg': lambda lc: comb.ColorAveragingModem(pal.PalSModem(lc)),
    'ntsc_avg': lambda lc: comb.ColorAveragingModem(ntsc.NtscModem(lc)),
    'secam': lambda lc: secam.SecamModem(lc),
    'secam_avg': lambda lc: comb.ColorAveragingModem(secam.SecamModem(lc)),
}
STANDARD = {'pal': 'GERBER_625', 'ntsc': 'NTSC_525', 'secam': 'GERBER_625'}


def line_config(stack, size, explicit=True):
    if not explicit:
        return line.LineConfig(tuple(int(v) for v in size))
    std = getattr(line.LineStandard, STANDARD[stack.split('_')[0]])
    return line.LineConfig(tuple(int(v) for v in size), std)


def make(stack, size, explicit=True):
    return STACKS[stack](line_config(stack, size, explicit))


def load(name):
    return numpy.load(os.path.join(GOLDEN, name + '.npz'))


def plans():
    with open(os.path.join(GOLDEN, 'plans.json')) as fh:
        return json.load(fh)


def rel_err(out, ref):
    """max |out - ref| / max |ref| (the tolerance convention of SURVEY.md Appendix C)."""
    out = numpy.asarray(out, dtype=numpy.float64)
    ref = numpy.asarray(ref, dtype=numpy.float64)
    return float(numpy.max(numpy.abs(out - ref)) / numpy.max(numpy.abs(ref)))
