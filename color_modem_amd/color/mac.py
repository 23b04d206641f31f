# -*- coding: utf-8 -*-
"""D2-MAC style time-multiplex modem (API mirror of /root/reference/color_modem/color/mac.py:9-125).

Host side: the variant presets and the colour matrices.  The per-line work - chroma decimation / interpolation with
scipy's resampling FIR and the sample re-arrangement of mac.py:56-69, 93-113 - runs in the HIP kernels ``mac_mod_kernel``
/ ``mac_demod_kernel`` (csrc/cm_mac_kernels.h) behind ``cm_mac_*`` (include/color_modem_hip.h).

Built: 720-sample rows and the 1080-sample line (``MacVariant.D2MAC_12MHZ``, the default), i.e. the cases in which
mac.py:49-52 and 71-74 do not resample; other image widths / ``D2MAC_7MHZ`` need rational resamplers that are not
built and raise NotImplementedError.
"""

import collections

import numpy

from color_modem_amd.rowapi import RowApi

MacVariant = collections.namedtuple('MacVariant', ['width'])

MacVariant.D2MAC_12MHZ = MacVariant(1080)
MacVariant.D2MAC_7MHZ = MacVariant(720)

LUMA_WIDTH, LINE_WIDTH = 720, 1080

# (luma, dr, db) = ENCODE . (r, g, b)   ref mac.py:29-32
ENCODE = numpy.array([[0.299, 0.587, 0.114],
                      [0.649827, -0.544149, -0.105678],
                      [-0.219167, -0.430271, 0.649438]])
# (r, g, b) = DECODE . (luma, dr, db)   ref mac.py:38-41
DECODE = numpy.array([[1.0, 1.0787486515641855, 0.0],
                      [1.0, -0.5494818514781797, -0.2649492993950324],
                      [1.0, 0.0, 1.364256480218281]])


class MacModem(RowApi):
    def __init__(self, line_config, variant_or_width=MacVariant.D2MAC_12MHZ):
        RowApi.__init__(self)
        self.line_config = line_config
        try:
            self._width = int(variant_or_width.width)
        except AttributeError:
            self._width = int(variant_or_width)
        if self._width != LINE_WIDTH:
            raise NotImplementedError('MacModem: only the 1080-sample line (MacVariant.D2MAC_12MHZ) is built; resampling '
                                      'it to %d samples (mac.py:71-74) is not' % self._width)
        if line_config.size[0] != LUMA_WIDTH:
            raise NotImplementedError('MacModem: rows of 720 samples only (mac.py:49-55 resamples other widths to 720 / '
                                      '360; those resamplers are not built), got %d' % line_config.size[0])

    @staticmethod
    def encode_components(r, g, b):
        assert len(r) == len(g) == len(b)
        y, dr, db = ENCODE @ numpy.stack([numpy.asarray(c, dtype=numpy.float64) for c in (r, g, b)])
        return y, dr, db

    @staticmethod
    def decode_components(luma, dr, db):
        assert len(luma) == len(dr) == len(db)
        r, g, b = DECODE @ numpy.stack([numpy.asarray(c, dtype=numpy.float64) for c in (luma, dr, db)])
        return r, g, b

    def demodulate_components(self, *args, **kwargs):
        raise AttributeError('MacModem has no demodulate_components (ref mac.py has none either)')

    def _stack(self):
        return {'kind': 'mac', 'backend': self}
