# -*- coding: utf-8 -*-
"""Proto-SECAM (1957, 819 lines): AM line-sequential colour (API mirror of
/root/reference/color_modem/color/protosecam.py:9-112).

Host side: the variant preset, the colour matrices and the four filter designs (the same scipy calls as the reference,
through color_modem_amd.utils).  The per-line work - x3 polyphase resampling around the recursive filters, the envelope
detector, the amplitude modulator - runs in ``proto_demod_kernel`` / ``proto_mod_kernel`` (csrc/cm_am_kernels.h) behind
``cm_am_*`` (include/color_modem_hip.h).
"""

import numpy

from color_modem_amd import qam, utils
from color_modem_amd.rowapi import RowApi


class ProtoSecamVariant(qam.QamConfig):
    pass


# ref protosecam.py:13-24: 819 * half the line frequency of the 819-line system
ProtoSecamVariant.SECAM_1957 = ProtoSecamVariant(fsc=8384512.5, bandwidth3db=800000.0, bandwidth20db=2000000.0)

# (luma, dr, db) = ENCODE . (r, g, b)   ref protosecam.py:55-61
ENCODE = numpy.array([[0.3, 0.59, 0.11],
                      [1.001, -0.8437, -0.1573],
                      [-0.336, -0.6608, 0.9968]])
# (r, g, b) = DECODE . (luma, dr, db)   ref protosecam.py:63-69
DECODE = numpy.array([[1.0, 0.6993006993006993, 0.0],
                      [1.0, -0.3555766267630674, -0.1664648910411622],
                      [1.0, 0.0, 0.8928571428571429]])

RESAMPLE_FACTOR = 3


class ProtoSecamModem(utils.ConstantFrequencyCarrier, RowApi):
    def __init__(self, line_config, variant=ProtoSecamVariant.SECAM_1957, premod_luma_filter=True):
        RowApi.__init__(self)
        self.line_config = line_config
        self.config = variant
        self._premod_luma_filter = bool(premod_luma_filter)
        fs = line_config.fs
        self._carrier_phase_step = numpy.pi * variant.fsc / fs
        self._demodulate_resample_factor = RESAMPLE_FACTOR
        # ref protosecam.py:33-48, the same four designs
        self._chroma_precorrect_lowpass = utils.iirdesign(2.0 * variant.bandwidth3db / fs, 2.0 * variant.bandwidth20db / fs,
                                                          3.0, 20.0)
        up = RESAMPLE_FACTOR * fs
        self._extract_chroma_up, self._remove_chroma_up = utils.iirsplitter(
            2.0 * variant.fsc / up, 2.0 * variant.bandwidth3db / up, 2.0 * variant.bandwidth20db / up, 3.0, 20.0)
        post = variant.bandwidth3db if variant.fsc < variant.bandwidth20db else variant.bandwidth20db
        self._chroma_up_post_demod_filter = utils.iirdesign(2.0 * min(post, variant.fsc - post) / up,
                                                            2.0 * max(post, variant.fsc - post) / up, 3.0, 20.0)

    @staticmethod
    def encode_components(r, g, b):
        assert len(r) == len(g) == len(b)
        luma, dr, db = ENCODE @ numpy.stack([numpy.asarray(c, dtype=numpy.float64) for c in (r, g, b)])
        return luma, dr, db

    @staticmethod
    def decode_components(luma, dr, db):
        assert len(luma) == len(dr) == len(db)
        r, g, b = DECODE @ numpy.stack([numpy.asarray(c, dtype=numpy.float64) for c in (luma, dr, db)])
        return r, g, b

    def demodulate_components(self, *args, **kwargs):
        raise AttributeError('ProtoSecamModem has no demodulate_components (ref protosecam.py has none either)')

    def _stack(self):
        return {'kind': 'protosecam', 'backend': self}
