# -*- coding: utf-8 -*-
"""D2-MAC style time-multiplex modem (API mirror of /root/reference/color_modem/color/mac.py:9-125).

Host side: the variant presets and the colour matrices.  The per-line work - chroma decimation / interpolation with
scipy's resampling FIR and the sample re-arrangement of mac.py:56-69, 93-113 - runs in the HIP kernels ``mac_mod_kernel``
/ ``mac_demod_kernel`` (csrc/cm_mac_kernels.h) behind ``cm_mac_*`` (include/color_modem_hip.h).

720-sample rows <-> the 1080-sample line (``MacVariant.D2MAC_12MHZ``, the default) run on the tuned kernels; every
other row / line length (``D2MAC_7MHZ``, other image widths: the rational resamplers of mac.py:49-55, 71-74, 88-91) on
the resampling kernels.  The decoder always returns rows of 720 samples, like the reference.
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
        if self._width < 1 or self._width > 16384 or line_config.size[0] > 4096:
            raise NotImplementedError('MacModem: lines of 1 .. 16384 samples and rows of up to 4096 samples are built (a call\'s rows live in one CU\'s LDS)')

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
