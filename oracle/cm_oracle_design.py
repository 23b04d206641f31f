# -*- coding: utf-8 -*-
"""The reference's filter designs, restated for the ORACLE with scipy.signal - the library the reference itself designs with.

TEST INFRASTRUCTURE (only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline may import oracle/).  The product designs its filters
with its own code (color_modem_amd/design.py); if the oracle took its coefficients from the product's objects, a slip there would move
product and oracle together.  So the oracle asks scipy, call for call as the reference does:

    FilterFunction.__init__                      /root/reference/color_modem/utils.py:9-26
    iirfilter / iirdesign / iirdesign_wc /       /root/reference/color_modem/utils.py:39-64
      iirsplitter
    QamColorModem.__init__                       /root/reference/color_modem/qam.py:14-18, 61-66
    PalDModem.__init__ (the AM low-pass)         /root/reference/color_modem/color/pal.py:67-69
    _notch                                       /root/reference/color_modem/comb.py:18-20
    SecamModem.__init__ and its two designs,     /root/reference/color_modem/color/secam.py:131-132, 153-186, 211-238
      FmDecoder.__init__

scipy.signal.iirdesign: releases >= 1.12 validate the band edges; the reference's NTSC set-up hands over an edge below zero,
so the harness behaviour recorded there is used - buttord + iirfilter, what iirdesign does behind its validation (identical coefficients
wherever current scipy accepts the request).  What the oracle takes from the product's modem objects are the CONSTANTS of the variant
presets (fsc, bandwidths, deviations ...: reference tables, pinned by tests/golden/plans.json) and the line geometry - not one filter
coefficient.
"""

import warnings

import numpy
import scipy.signal

_BANDSTOP_NAMES = frozenset(('bs', 'bandstop', 'bands', 'stop'))


class Designed(object):
    """What the oracle needs of a reference FilterFunction: (b, a), the delay compensation, the phase at the band centre."""

    def __init__(self, b, a, wp, btype, shift):                 # utils.py:9-26
        self.b = numpy.array(b, dtype=numpy.float64)
        self.a = numpy.array(a, dtype=numpy.float64)
        wp = numpy.atleast_1d(wp)
        if len(wp) > 1 and btype.lower() not in _BANDSTOP_NAMES:
            shiftfreq = float(numpy.average(wp))
        else:
            shiftfreq = 0.0
        with warnings.catch_warnings():
            warnings.simplefilter('ignore')
            if shift:
                self.shift = int(numpy.round(scipy.signal.group_delay((self.b, self.a), [shiftfreq], fs=2.0)[1][0]))
            else:
                self.shift = 0
            response = scipy.signal.freqz(self.b, self.a, worN=[shiftfreq], fs=2.0)[1][0]
        self.phase_shift = float((numpy.angle(response) + self.shift * numpy.pi * shiftfreq) % (2.0 * numpy.pi))


def _scipy_iirdesign(wp, ws, gpass, gstop):
    """scipy.signal.iirdesign(..., ftype='butter') without the band-edge validation of scipy >= 1.12."""
    wp1, ws1 = numpy.atleast_1d(wp), numpy.atleast_1d(ws)
    if len(wp1) == 1:
        btype = 'lowpass' if wp1[0] < ws1[0] else 'highpass'
    else:
        btype = 'bandstop' if wp1[0] < ws1[0] else 'bandpass'
    with warnings.catch_warnings():
        warnings.simplefilter('ignore')
        order, wn = scipy.signal.buttord(wp, ws, gpass, gstop, analog=False)
    return scipy.signal.iirfilter(order, wn, rp=gpass, rs=gstop, btype=btype, ftype='butter', output='ba')


def iirfilter(N, Wn, rp=None, rs=None, btype='band', ftype='butter', shift=True):      # utils.py:39-41
    b, a = scipy.signal.iirfilter(N, Wn, rp, rs, btype, ftype=ftype)
    return Designed(b, a, Wn, btype, shift)


def iirdesign(wp, ws, gpass, gstop, shift=True):                                          # utils.py:44-51
    smallest = numpy.nextafter(0.0, 1.0)
    largest = numpy.nextafter(1.0, 0.0)
    b, a = _scipy_iirdesign(numpy.maximum(wp, smallest), numpy.minimum(ws, largest), gpass, gstop)
    btype = 'band'
    if len(numpy.atleast_1d(wp)) > 1 and len(numpy.atleast_1d(ws)) > 1 and ws[0] > wp[0]:
        btype = 'bandstop'
    return Designed(b, a, wp, btype, shift)


def iirdesign_wc(wc, wp, ws, gpass, gstop, shift=True):                                   # utils.py:54-55
    return iirdesign([wc - wp, wc + wp], [wc - ws, wc + ws], gpass, gstop, shift)


def iirsplitter(wc, wp, ws, gpass, gstop, shift=True):                                    # utils.py:58-64
    def invert_db(db):
        return -(20.0 * numpy.log10(1.0 - 10.0 ** (-db / 20.0)))

    bpass = iirdesign_wc(wc, wp, ws, gpass, gstop, shift)
    bstop = iirdesign_wc(wc, ws, wp, invert_db(gstop), invert_db(gpass), shift)
    return bpass, bstop


def qam_filters(fs, config):
    """(pre-correction low-pass, extract band-pass at 2x, remove band-stop at 2x, detector low-pass, carrier phase step) of
    AbstractQamColorModem(line_config, config): qam.py:64-66 -> qam.py:14-18."""
    wc, wp, ws = 2.0 * config.fsc / fs, 2.0 * config.bandwidth3db / fs, 2.0 * config.bandwidth20db / fs
    gpass, gstop = 3.0, 20.0
    carrier_phase_step = 0.5 * numpy.pi * wc
    precorrect = iirdesign(wp, ws, gpass, gstop)
    extract2x, remove2x = iirsplitter(0.5 * wc, 0.5 * wp, 0.5 * ws, gpass, gstop)
    demod_lowpass = iirfilter(6, wc - 0.5 * ws, rs=48.0, btype='lowpass', ftype='cheby2')
    return precorrect, extract2x, remove2x, demod_lowpass, carrier_phase_step


def pald_filter(fsc, carrier_phase_step):                                                  # pal.py:67-69
    return iirfilter(6, (1.0 - 1300000.0 / fsc) * carrier_phase_step / numpy.pi, rs=48.0, btype='lowpass', ftype='cheby2')


def notch(fsc, fs, q):                                                                     # comb.py:18-20
    b, a = scipy.signal.iirnotch(2.0 * fsc / fs, q)
    return Designed(b, a, wp=0.0, btype='bandstop', shift=True)


def secam_precorrect(wc, k):                                                               # secam.py:211-221
    assert k != 1.0
    forward_b, forward_a = scipy.signal.iirfilter(1, k * wc, btype='highpass', ftype='butter')
    assert forward_a[0] == 1.0
    forward_b[0] = (k - 1.0) * forward_b[0] + 1.0
    forward_b[1] = (k - 1.0) * forward_b[1] + forward_a[1]
    backward_b = numpy.array([1.0, forward_a[1]]) / forward_b[0]
    backward_a = numpy.array([1.0, forward_b[1] / forward_b[0]])
    return (Designed(forward_b, forward_a, k * wc, btype='highpass', shift=False),
            Designed(backward_b, backward_a, k * wc, btype='lowpass', shift=False))


def secam_bell(f0, f_max, kn, kd):                                                         # secam.py:224-238
    def gain(f):
        return numpy.sqrt(
            (kd * kd * f0 * f0 * f0 * f0 + (1 - 2 * kd * kd) * f * f * f0 * f0 + kd * kd * f * f * f * f) / (
                kn * kn * f0 * f0 * f0 * f0 + (1 - 2 * kn * kn) * f * f * f0 * f0 + kn * kn * f * f * f * f))

    def gain_db(f):
        return 10.0 * numpy.log10(gain(f))

    assert kn != kd
    wp2 = f0 + 1 / 256.0
    wp1 = f0 * f0 / wp2
    ws2 = f_max
    ws1 = f0 * f0 / ws2
    return iirdesign([wp1, wp2], [ws1, ws2], -gain_db(wp2), -gain_db(ws2), shift=False)


def secam_filters(fs, variant):
    """The seven filters of SecamModem(line_config, variant) in the order the oracle's descriptor takes them (secam.py:153-186, 131-132):
    pre-correction low-pass, LF pre-emphasis and its inverse (or None), receiver bell (or None), chroma band-pass, luma band-stop, the FM
    discriminator's low-pass; and the discriminator centre."""
    flimit_min = 2.0 * (variant.bell_f0 + variant.flimit_minbell) / fs
    flimit_max = 2.0 * (variant.bell_f0 + variant.flimit_maxbell) / fs
    bell_f0 = 2.0 * variant.bell_f0 / fs
    bell = secam_bell(bell_f0, flimit_max, variant.bell_kn, variant.bell_kd) if variant.bell_kn != variant.bell_kd else None
    pre_lowpass = iirdesign(wp=2.0 * 1300000.0 / fs, ws=2.0 * 3500000.0 / fs, gpass=3.0, gstop=30.0)
    forward = backward = None
    if variant.lf_precorrect_k != 1.0:
        forward, backward = secam_precorrect(2.0 * variant.lf_precorrect_f1 / fs, variant.lf_precorrect_k)
    center = 0.5 * (flimit_min + flimit_max)
    dev = 0.5 * (flimit_max - flimit_min)
    chroma = iirfilter(3, [center - dev, center + dev], rp=0.1, btype='bandpass', ftype='cheby1')
    luma = iirfilter(3, [center - dev * numpy.e, center + dev * numpy.e], btype='bandstop', ftype='bessel')
    fm_lowpass = iirfilter(6, (2.0 * center - dev) / 2, rs=48.0, btype='lowpass', ftype='cheby2')      # FmDecoder(center, dev): secam.py:131-132
    return [pre_lowpass, forward, backward, bell, chroma, luma, fm_lowpass], center


def proto_filters(fs, variant):
    """ProtoSecamModem.__init__ (protosecam.py:32-50): carrier phase step, pre-correction low-pass, the band-pass / band-stop pair and the
    post-detection low-pass at three times the sampling rate."""
    factor = 3
    out = {'carrier_phase_step': numpy.pi * variant.fsc / fs,
           'pre': iirdesign(2.0 * variant.bandwidth3db / fs, 2.0 * variant.bandwidth20db / fs, 3.0, 20.0)}
    out['extract_up'], out['remove_up'] = iirsplitter(2.0 * variant.fsc / (factor * fs), 2.0 * variant.bandwidth3db / (factor * fs),
                                                      2.0 * variant.bandwidth20db / (factor * fs), 3.0, 20.0)
    post = variant.bandwidth3db if variant.fsc < variant.bandwidth20db else variant.bandwidth20db
    out['post_demod'] = iirdesign(2.0 * min(post, variant.fsc - post) / (factor * fs), 2.0 * max(post, variant.fsc - post) / (factor * fs), 3.0, 20.0)
    return out


def niir_filters(fs, config):
    """NiirModem.__init__ (niir.py:11-23) with _demodulate_am_design (niir.py:90-93)."""
    factor = 3
    wc, wp, ws = 2.0 * config.fsc / fs, 2.0 * config.bandwidth3db / fs, 2.0 * config.bandwidth20db / fs
    return {'carrier_phase_step': 2.0 * numpy.pi * config.fsc / fs,
            'pre': iirdesign(wp, ws, 3.0, 20.0),
            'baseband_up': iirdesign(wp / factor, ws / factor, 3.0, 20.0),
            'bandpass_up': iirdesign_wc(wc / factor, wp / factor, ws / factor, 3.0, 20.0)}
