# -*- coding: utf-8 -*-
"""PAL colour modems (API mirror of /root/reference/color_modem/color/pal.py).

``PalSModem`` (simple PAL, pal.py:28-59), ``PalDModem`` (delay-line PAL, 2-line comb,
pal.py:62-127) and ``Pal3DModem`` (3-line comb, pal.py:130-234).  The classes hold the same
constructor-time state as the reference (filters, constants); rows are processed by the HIP
kernels through :mod:`color_modem_amd.engine`.
"""

import numpy

from color_modem_amd import comb, qam, utils


class PalVariant(qam.QamConfig):
    def __new__(cls, fsc, bandwidth3db=1300000.0, bandwidth20db=4000000.0):
        return super(PalVariant, cls).__new__(cls, fsc, bandwidth3db, bandwidth20db)


PalVariant.PAL = PalVariant(fsc=4433618.75)
PalVariant.PAL_A = PalVariant(fsc=2660343.75)
PalVariant.PAL_M = PalVariant(fsc=227.25 * 15750.0 * 1000.0 / 1001.0, bandwidth20db=3600000.0)
PalVariant.PAL_N = PalVariant(fsc=3582056.25, bandwidth20db=3600000.0)

# (y, u, v) = ENCODE . (r, g, b)   ref pal.py:35-37
ENCODE = numpy.array([[0.299, 0.587, 0.114],
                      [-0.147407, -0.289391, 0.436798],
                      [0.614777, -0.514799, -0.099978]])
# (r, g, b) = DECODE . (y, u, v)   ref pal.py:43-45
DECODE = numpy.array([[1.0, 0.0, 1.140250855188141],
                      [1.0, -0.3939307027516405, -0.5808092090310976],
                      [1.0, 2.028397565922921, 0.0]])


class PalSModem(qam.AbstractQamColorModem):
    system = 'pal'
    v_switch = True
    encode_matrix = ENCODE
    decode_matrix = DECODE

    def __init__(self, line_config, variant=PalVariant.PAL):
        super(PalSModem, self).__init__(line_config, variant)

    @staticmethod
    def encode_components(r, g, b):
        assert len(r) == len(g) == len(b)
        y, u, v = ENCODE.dot(numpy.stack([numpy.asarray(r, float), numpy.asarray(g, float), numpy.asarray(b, float)]))
        return y, u, v

    @staticmethod
    def decode_components(y, u, v):
        assert len(y) == len(u) == len(v)
        r, g, b = DECODE.dot(numpy.stack([numpy.asarray(y, float), numpy.asarray(u, float), numpy.asarray(v, float)]))
        return r, g, b

    def _stack(self):
        return {'kind': 'pal_s', 'backend': self}


class PalDModem(comb.AbstractCombModem):
    def __init__(self, line_config, variant=PalVariant.PAL, *args, **kwargs):
        super(PalDModem, self).__init__(PalSModem(line_config, variant), *args, **kwargs)
        self._sin_factor = numpy.sin(0.5 * self.backend.line_shift)
        self._cos_factor = numpy.cos(0.5 * self.backend.line_shift)
        # same design request as ref pal.py:67-69
        self._filter = utils.iirfilter(
            6, (1.0 - 1300000.0 / self.backend.config.fsc) * self.backend.qam.carrier_phase_step / numpy.pi,
            rs=48.0, btype='lowpass', ftype='cheby2')

    def _stack(self):
        return {'kind': 'pal_d', 'backend': self.backend, 'comb': self, 'comb_notch': self.notch}


class Pal3DModem(PalDModem):
    def __init__(self, *args, **kwargs):
        use_sin = kwargs.pop('use_sin', True)
        use_cos = kwargs.pop('use_cos', True)
        avg = kwargs.pop('avg', None)
        super(Pal3DModem, self).__init__(*args, **kwargs)
        lssin = numpy.sin(self.backend.line_shift)
        lscos = numpy.cos(self.backend.line_shift)
        if abs(lssin) < 0.1:  # ref pal.py:154-156
            use_sin = False
        if abs(lscos) > 0.9:  # ref pal.py:158-160
            use_cos = False
        self.demodulation_delay = 1 if (use_cos or use_sin) else 0
        self._use_sin = use_sin
        self._use_cos = use_cos
        if use_sin:
            self._sin_sum_factor = 0.5 / lssin
        if use_cos:
            self._cos_u_factor = -0.5 / (1.0 - lscos)
            self._cos_v_factor = -0.5 / (1.0 + lscos)
        self._avg = avg if avg is not None else comb.avg

    def _stack(self):
        if not (self._use_sin or self._use_cos):
            return {'kind': 'pal_d', 'backend': self.backend, 'comb': self, 'comb_notch': self.notch}
        return {'kind': 'pal_3d', 'backend': self.backend, 'comb': self, 'comb_notch': self.notch}
