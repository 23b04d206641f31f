# -*- coding: utf-8 -*-
"""NTSC colour modems (API mirror of /root/reference/color_modem/color/ntsc.py:8-82)."""

import numpy

from color_modem_amd import comb, qam


class NtscVariant(qam.QamConfig):
    def __new__(cls, fsc, bandwidth3db=1300000.0, bandwidth20db=3600000.0):
        return super(NtscVariant, cls).__new__(cls, fsc, bandwidth3db, bandwidth20db)


NtscVariant.NTSC = NtscVariant(fsc=227.5 * 15750.0 * 1000.0 / 1001.0)
NtscVariant.NTSC_A = NtscVariant(fsc=2657812.5, bandwidth3db=1000000.0, bandwidth20db=2500000.0)
NtscVariant.NTSC_I = NtscVariant(fsc=4429687.5)
NtscVariant.NTSC443 = NtscVariant(fsc=4433618.75)
NtscVariant.NTSC_N = NtscVariant(fsc=3585937.5)
NtscVariant.NTSC361 = NtscVariant(fsc=229.5 * 15750.0 * 1000.0 / 1001.0)

# ref ntsc.py:30-32
ENCODE = numpy.array([[0.3, 0.59, 0.11],
                      [-0.1476019510016258, -0.2893575108184752, 0.436959461820101],
                      [0.6183717846575098, -0.5185533057776567, -0.099818478879853]])
# ref ntsc.py:38-40, columns ordered (y, u, v)
DECODE = numpy.array([[0.9999999999999998, 0.007249535771601484, 1.133735501874552],
                      [1.0, -0.3834753199055935, -0.5766784873222262],
                      [1.0, 2.037050709207452, 0.001087790524980047]])


class NtscModem(qam.AbstractQamColorModem):
    system = 'ntsc'
    v_switch = False
    encode_matrix = ENCODE
    decode_matrix = DECODE

    def __init__(self, line_config, variant=NtscVariant.NTSC):
        super(NtscModem, self).__init__(line_config, variant)

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
        return {'kind': 'ntsc', 'backend': self}


class NtscCombModem(comb.AbstractCombModem):
    def __init__(self, line_config, variant=NtscVariant.NTSC, *args, **kwargs):
        super(NtscCombModem, self).__init__(NtscModem(line_config, variant), *args, **kwargs)
        sine = numpy.sin(self.backend.line_shift * 0.5)
        # ref ntsc.py:55-59: the comb is switched off when consecutive lines are nearly in phase
        self._factor = 0.5 / sine if abs(sine) > 0.05 else numpy.inf

    def _stack(self):
        return {'kind': 'ntsc_comb', 'backend': self.backend, 'comb': self, 'comb_notch': self.notch}
