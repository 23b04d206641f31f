# -*- coding: utf-8 -*-
"""QAM colour sub-carrier core: filter set and glue (host side).

API mirror of /root/reference/color_modem/qam.py:8-72.  ``QamColorModem`` here owns the four
filter designs and the carrier step; the per-sample work (qam.py:20-58 in the reference)
lives in color_modem_amd/csrc (kernels ``qam_demod`` / ``qam_mod``).
"""

import collections

import numpy

from color_modem_amd import utils
from color_modem_amd.rowapi import RowApi

QamConfig = collections.namedtuple('QamConfig', ['fsc', 'bandwidth3db', 'bandwidth20db'])


class QamColorModem(object):
    def __init__(self, wc, wp, ws, gpass, gstop):
        # same four designs as ref qam.py:14-18
        self.carrier_phase_step = 0.5 * numpy.pi * wc
        self._chroma_precorrect_lowpass = utils.iirdesign(wp, ws, gpass, gstop)
        self._extract_chroma2x, self._remove_chroma2x = utils.iirsplitter(0.5 * wc, 0.5 * wp, 0.5 * ws, gpass, gstop)
        self._demod_lowpass = utils.iirfilter(6, wc - 0.5 * ws, rs=48.0, btype='lowpass', ftype='cheby2')

    @property
    def extract_chroma_phase_shift(self):
        return self._extract_chroma2x.phase_shift


class AbstractQamColorModem(utils.ConstantFrequencyCarrier, RowApi):
    """Base of NtscModem / PalSModem (ref qam.py:61-72)."""

    def __init__(self, line_config, config):
        RowApi.__init__(self)
        self.line_config = line_config
        self.config = config
        self.qam = QamColorModem(2.0 * config.fsc / line_config.fs, 2.0 * config.bandwidth3db / line_config.fs,
                                 2.0 * config.bandwidth20db / line_config.fs, 3.0, 20.0)
