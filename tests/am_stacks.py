# -*- coding: utf-8 -*-
"""Stacks of the amplitude-modulated line-sequential standards, named as in tests/golden/make_golden_am.py."""
import os

import numpy

from color_modem_amd import comb, line
from color_modem_amd.color import niir, protosecam

GOLDEN = os.path.join(os.path.dirname(os.path.abspath(__file__)), 'golden')

STACKS = {
    'proto': lambda lc: protosecam.ProtoSecamModem(lc),
    'proto_avg': lambda lc: comb.ColorAveragingModem(protosecam.ProtoSecamModem(lc)),
    'proto_nofilter': lambda lc: protosecam.ProtoSecamModem(lc, premod_luma_filter=False),
    'proto_625': lambda lc: protosecam.ProtoSecamModem(lc),
    'niir': lambda lc: niir.NiirModem(lc),
    'niir_hue': lambda lc: niir.HueCorrectingNiirModem(lc),
    'niir_525': lambda lc: niir.NiirModem(lc),
    'niir_noise': lambda lc: niir.NiirModem(lc, noise_level=0.05),
    'niir_hue_noise': lambda lc: niir.HueCorrectingNiirModem(lc, noise_level=0.08),
}
STACKS['niir_grey'] = STACKS['niir']             # the same modems on pictures with grey / nearly grey areas (goldens am_mod_*_grey)
STACKS['niir_hue_grey'] = STACKS['niir_hue']
STACKS['niir_grey'] = STACKS['niir']             # the same modems on pictures with grey / nearly grey areas (goldens am_mod_*_grey)
STACKS['niir_hue_grey'] = STACKS['niir_hue']
DECODER_OF = {'proto_avg': 'proto'}


def load(name):
    return numpy.load(os.path.join(GOLDEN, name + '.npz'))


def make(stack, z):
    lc = line.LineConfig(tuple(int(v) for v in z['size']), getattr(line.LineStandard, str(z['standard'])))
    return STACKS[stack](lc)
