# -*- coding: utf-8 -*-
"""SECAM FM colour modem (API mirror of /root/reference/color_modem/color/secam.py:10-304).

Host side: the eight variant presets, the six filter designs and the normalised FM constants.
The per-sample encode/decode (secam.py:127-149, 240-304) runs in the HIP kernels
``secam_demod`` / ``secam_mod``.
"""

import collections

import numpy
from color_modem_amd import design

from color_modem_amd import utils
from color_modem_amd.rowapi import RowApi

SecamVariant = collections.namedtuple(
    'SecamVariant', ['fsc_dr', 'fsc_db', 'fdev_dr', 'fdev_db', 'flimit_minbell', 'flimit_maxbell',
                     'm0', 'bell_f0', 'bell_kn', 'bell_kd', 'lf_precorrect_f1', 'lf_precorrect_k'])


def _variant(fsc_dr, fsc_db, fdev_dr, fdev_db, fmin, fmax, m0, f0, kn, kd, f1, k):
    return SecamVariant(fsc_dr=fsc_dr, fsc_db=fsc_db, fdev_dr=fdev_dr, fdev_db=fdev_db, flimit_minbell=fmin,
                        flimit_maxbell=fmax, m0=m0, bell_f0=f0, bell_kn=kn, bell_kd=kd, lf_precorrect_f1=f1,
                        lf_precorrect_k=k)


_NTSC_FSC = 227.5 * 15750.0 * 1000.0 / 1001.0
# presets: ref secam.py:15-124 (I, II: early proposals; III: as proposed; SECAM: IIIb as broadcast; ...)
SecamVariant.SECAM_I = _variant(4437500.0, 4437500.0, 250000.0, 250000.0, -250000.0, 250000.0, 0.2, 4437500.0,
                                1.0, 1.0, 0.0, 1.0)
SecamVariant.SECAM_II = _variant(4437500.0, 4437500.0, 250000.0, 250000.0, -250000.0, 250000.0, 0.1, 4437500.0,
                                 16.0, 1.26, 0.0, 1.0)
SecamVariant.SECAM_III = _variant(4437500.0, 4437500.0, 230000.0, 230000.0, -450000.0, 350000.0, 0.1, 4437500.0,
                                  16.0, 1.26, 70000.0, 5.6)
SecamVariant.SECAM = _variant(4406250.0, 4250000.0, 280000.0, 230000.0, -386000.0, 470250.0, 0.115, 4286000.0,
                              16.0, 1.26, 85000.0, 3.0)
SecamVariant.SECAM_A = _variant(2660000.0, 2660000.0, 250000.0, 250000.0, -250000.0, 250000.0, 0.2, 2660000.0,
                                1.0, 1.0, 0.0, 1.0)
SecamVariant.SECAM_E = _variant(8370000.0, 8370000.0, 250000.0, 250000.0, -250000.0, 250000.0, 0.2, 8370000.0,
                                1.0, 1.0, 0.0, 1.0)
SecamVariant.SECAM_M = _variant(_NTSC_FSC, _NTSC_FSC, 230000.0, 230000.0, -500000.0, 500000.0, 0.1, _NTSC_FSC,
                                16.0, 1.26, 70000.0, 5.6)
SecamVariant.SECAM_N = _variant(3578125.0, 3578125.0, 230000.0, 230000.0, -500000.0, 500000.0, 0.1, 3578125.0,
                                16.0, 1.26, 70000.0, 5.6)

# (luma, dr, db) = ENCODE . (r, g, b)   ref secam.py:195-197
ENCODE = numpy.array([[0.299, 0.587, 0.114],
                      [-1.333302, 1.116474, 0.216828],
                      [-0.449995, -0.883435, 1.33343]])
# (r, g, b) = DECODE . (luma, dr, db)   ref secam.py:205-207
DECODE = numpy.array([[1.0, -0.5257623554153522, 0.0],
                      [1.0, 0.2678074007993021, -0.1290417517983779],
                      [1.0, 0.0, 0.6644518272425249]])


class FmDecoder(object):
    """Design record of the quadrature FM discriminator (ref secam.py:127-132)."""

    def __init__(self, fc, dev, resample_rate=2):
        if resample_rate != 2:
            raise NotImplementedError('only the 2x discriminator used by SecamModem is built')
        self._fc = fc
        self._resample_rate = resample_rate
        self._lowpass = utils.iirfilter(6, (2.0 * fc - dev) / resample_rate, rs=48.0, btype='lowpass',
                                        ftype='cheby2')


class SecamModem(RowApi):
    system = 'secam'
    encode_matrix = ENCODE
    decode_matrix = DECODE
    # True (set it before the first call): the decoder's chroma front end in float64 whatever the shape - the library selects
    # it by itself where float32 rounding noise would come near 1e-5 (cm_secam_desc.present | CM_SECAM_FLOAT64)
    float64_front_end = False

    def __init__(self, line_config, variant=SecamVariant.SECAM, alternate_phases=False):
        RowApi.__init__(self)
        self._line_config = line_config
        self._variant = variant
        self._fsc_dr = 2.0 * variant.fsc_dr / line_config.fs
        self._fsc_db = 2.0 * variant.fsc_db / line_config.fs
        self._fdev_dr = 2.0 * variant.fdev_dr / line_config.fs
        self._fdev_db = 2.0 * variant.fdev_db / line_config.fs
        self._flimit_min = 2.0 * (variant.bell_f0 + variant.flimit_minbell) / line_config.fs
        self._flimit_max = 2.0 * (variant.bell_f0 + variant.flimit_maxbell) / line_config.fs
        self._bell_f0 = 2.0 * variant.bell_f0 / line_config.fs
        self._alternate_phases = bool(alternate_phases)
        # colour sync phase sequence over 6 lines, ref secam.py:163-166
        self._start_phase_inversions = ([False, False, False, True, True, True] if alternate_phases
                                        else [False, False, True, False, False, True])
        self._chroma_demod_bell = None
        if variant.bell_kn != variant.bell_kd:
            self._chroma_demod_bell = self._chroma_demod_bell_design(self._bell_f0, self._flimit_max,
                                                                     variant.bell_kn, variant.bell_kd)
        self._chroma_precorrect_lowpass = utils.iirdesign(wp=2.0 * 1300000.0 / line_config.fs,
                                                          ws=2.0 * 3500000.0 / line_config.fs, gpass=3.0, gstop=30.0)
        self._chroma_precorrect = None
        self._reverse_chroma_precorrect = None
        if variant.lf_precorrect_k != 1.0:
            self._chroma_precorrect, self._reverse_chroma_precorrect = self._chroma_precorrect_design(
                2.0 * variant.lf_precorrect_f1 / line_config.fs, variant.lf_precorrect_k)
        center = 0.5 * (self._flimit_min + self._flimit_max)
        dev = 0.5 * (self._flimit_max - self._flimit_min)
        self._chroma_demod_filter_order = 3
        self._chroma_demod_chroma_filter = utils.iirfilter(3, [center - dev, center + dev],
                                                           rp=0.1, btype='bandpass', ftype='cheby1')
        self._chroma_demod_luma_filter = utils.iirfilter(3, [center - dev * numpy.e, center + dev * numpy.e],
                                                         btype='bandstop', ftype='bessel')
        self._chroma_demod = FmDecoder(center, dev)

    @property
    def line_config(self):
        return self._line_config

    @staticmethod
    def encode_components(r, g, b):
        assert len(r) == len(g) == len(b)
        luma, dr, db = ENCODE.dot(numpy.stack([numpy.asarray(r, float), numpy.asarray(g, float),
                                               numpy.asarray(b, float)]))
        return luma, dr, db

    @staticmethod
    def decode_components(luma, dr, db):
        assert len(luma) == len(dr) == len(db)
        r, g, b = DECODE.dot(numpy.stack([numpy.asarray(luma, float), numpy.asarray(dr, float),
                                          numpy.asarray(db, float)]))
        return r, g, b

    @staticmethod
    def _chroma_precorrect_design(wc, k):
        """LF pre-emphasis 1 + (k - 1) * highpass and its exact inverse (ref secam.py:211-221)."""
        assert k != 1.0
        hp_b, hp_a = design.iirfilter(1, k * wc, btype='highpass', ftype='butter')
        assert hp_a[0] == 1.0
        fwd_b = numpy.array([(k - 1.0) * hp_b[0] + 1.0, (k - 1.0) * hp_b[1] + hp_a[1]])
        fwd_a = numpy.array(hp_a, dtype=numpy.float64)
        back_b = numpy.array([1.0, fwd_a[1]]) / fwd_b[0]
        back_a = numpy.array([1.0, fwd_b[1] / fwd_b[0]])
        forward = utils.FilterFunction(fwd_b, fwd_a, k * wc, btype='highpass', shift=False)
        backward = utils.FilterFunction(back_b, back_a, k * wc, btype='lowpass', shift=False)
        return forward, backward

    @staticmethod
    def _chroma_demod_bell_design(f0, f_max, kn, kd):
        """Receiver 'anti-bell' as a Butterworth band-pass matched at two gains (ref secam.py:224-238)."""
        def gain_db(f):
            num = kd * kd * f0 ** 4 + (1 - 2 * kd * kd) * f * f * f0 * f0 + kd * kd * f ** 4
            den = kn * kn * f0 ** 4 + (1 - 2 * kn * kn) * f * f * f0 * f0 + kn * kn * f ** 4
            return 10.0 * numpy.log10(numpy.sqrt(num / den))

        assert kn != kd
        wp2 = f0 + 1 / 256.0
        wp1 = f0 * f0 / wp2
        ws2 = f_max
        ws1 = f0 * f0 / ws2
        return utils.iirdesign([wp1, wp2], [ws1, ws2], -gain_db(wp2), -gain_db(ws2), shift=False)

    def _start_phase_inverted(self, frame, line):
        frame %= 6
        line_in_field = (23 if line % 2 == 0 else 336) + line // 2
        return self._start_phase_inversions[(frame * 625 + line_in_field) % 6] ^ (frame % 2 == 1)

    def _stack(self):
        return {'kind': 'secam', 'backend': self}
