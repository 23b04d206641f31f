#!/usr/bin/env python3
"""VERDICT r03 item 4, the bounded host experiment (no GPU): would merging stages of the PAL-D line take arithmetic out?

(1) E(x) = dn2(BPF(up2(x)))  (qam.py:34-37) as ONE 1x-rate FIR + an all-pole recursion with squared poles:
    B(z)/A(z) = B(z) A(-z) / A2(z^2); 1/A2(z^2) passes the decimator as 1/A2(z) (noble identity), and
    dn2 . FIR[h * B A(-) * h] . up2 is the even polyphase branch of one 2x-rate FIR.
(2) dn2(F_palD(z)) (pal.py:71-77) the same way: recursion at 1x on the decimated stream, FIR at the 2x rate.

Prints, per form: multiply-adds per pixel against the section-by-section form the kernel runs, and the float32 error of the
merged form on the interior of a row (the reference pads / truncates between the stages, so the row ends need their own
bodies either way).  Result recorded in profiles/r04_headline_bound.txt."""
import os
import sys

import numpy
import scipy.signal

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from color_modem_amd import line  # noqa: E402
from color_modem_amd.color import pal  # noqa: E402


def halfband():
    # resample_poly's default design for up = 2 or down = 2: firwin(2 * 10 * 2 + 1, 1 / 2, window=('kaiser', 5.0))
    return scipy.signal.firwin(41, 0.5, window=('kaiser', 5.0))


def ff(f, x):
    """FilterFunction.__call__ of the reference (utils.py:28-36), float64."""
    pad = numpy.concatenate([x, numpy.full(f.shift, x[-1])])
    return scipy.signal.lfilter(f.b, f.a, pad)[f.shift:]


def lfilter32(b, a, x):
    """Direct-form recursion in float32 arithmetic (worst case for the merged forms: no sections)."""
    return scipy.signal.lfilter(numpy.asarray(b, numpy.float32), numpy.asarray(a, numpy.float32), numpy.asarray(x, numpy.float32))


def sos32(sos, x):
    y = numpy.asarray(x, numpy.float32)
    for s in numpy.asarray(sos, numpy.float32):
        y = scipy.signal.lfilter(s[:3], s[3:], y)
    return y


def main():
    lc = line.LineConfig((720, 576))
    m = pal.PalDModem(lc)
    bpf = m.backend.qam._extract_chroma2x
    lpf = m._filter
    h = halfband()
    rng = numpy.random.default_rng(7)
    W = 720
    x = numpy.convolve(rng.uniform(0, 1, W + 3), numpy.ones(4) / 4, 'valid')
    k = numpy.arange(W)
    x = 0.5 * x + 0.3 * numpy.sin(k * 2.0 * m.backend.qam.carrier_phase_step + 0.7) * x   # luma + chroma at the sub-carrier
    inner = slice(120, W - 120)

    print('PAL-D 720 wide: band-pass order %d (shift %d), detector low-pass order %d (shift %d), half-band %d taps (%d non-zero)'
          % (bpf.order, bpf.shift, lpf.order, lpf.shift, len(h), int(numpy.sum(numpy.abs(h) > 1e-14))))

    # ---- (1) E(x) ----------------------------------------------------------------------------------------------------------
    up = scipy.signal.resample_poly(x, 2, 1)
    e_ref = scipy.signal.resample_poly(ff(bpf, up), 1, 2)
    a_neg = bpf.a * (-1.0) ** numpy.arange(len(bpf.a))
    a2 = numpy.polymul(bpf.a, a_neg)            # A(z) A(-z): only even powers
    assert numpy.max(numpy.abs(a2[1::2])) < 1e-12
    a2 = a2[0::2]
    num = numpy.polymul(bpf.b, a_neg)           # B(z) A(-z) at the 2x rate
    g = numpy.convolve(numpy.convolve(2.0 * h, num), h)   # up2 = 2 h on the zero-stuffed row, dn2 = h then every second sample
    # output sample n = sum_j g[j] xs[2n + d - j], xs the zero-stuffed row: the even / odd branch of g by the parity of the delay
    d = 20 + 20 + bpf.shift
    branch = g[(d % 2)::2]
    lead = (d - (d % 2)) // 2
    for dtype, name in ((numpy.float64, 'float64'), (numpy.float32, 'float32')):
        fir = numpy.convolve(numpy.asarray(x, dtype), numpy.asarray(branch, dtype))[lead:lead + W]
        if dtype is numpy.float64:
            e_m = scipy.signal.lfilter([1.0], a2, fir)
            e_m2 = e_m
        else:
            e_m = lfilter32([1.0], a2, fir)                                   # direct form, order 8
            e_m2 = sos32(scipy.signal.tf2sos([1.0], a2), fir)                  # all-pole sections
        s = numpy.max(numpy.abs(e_ref))
        print('(1) merged E, %s: interior error %.2e (direct form) %.2e (all-pole sections) of the band-passed row\'s peak'
              % (name, numpy.max(numpy.abs(e_m - e_ref)[inner]) / s, numpy.max(numpy.abs(e_m2 - e_ref)[inner]) / s))
    taps = int(numpy.sum(numpy.abs(branch) > 1e-9 * numpy.max(numpy.abs(branch))))
    sect = bpf.order // 2
    now = 20 + 3 * sect * 2 + 21
    merged = taps + 2 * len(scipy.signal.tf2sos([1.0], a2))
    print('(1) multiply-adds per pixel: now up2 20 + band-pass %d sections x 3 x 2 samples + dn2 21 = %d; merged %d FIR taps '
          '(|tap| > 1e-9 of the largest; dynamic range %.1e) + %d recursion = %d  ->  saves %d'
          % (sect, now, taps, numpy.max(numpy.abs(branch)) / numpy.min(numpy.abs(branch[numpy.abs(branch) > 0])),
             merged - taps, merged, now - merged))
    print('    pole radius: band-pass %.4f, squared %.4f' % (numpy.max(numpy.abs(numpy.roots(bpf.a))), numpy.max(numpy.abs(numpy.roots(a2)))))

    # ---- (2) dn2(F_palD(z)) --------------------------------------------------------------------------------------------------
    la_neg = lpf.a * (-1.0) ** numpy.arange(len(lpf.a))
    la2 = numpy.polymul(lpf.a, la_neg)[0::2]
    lnum = numpy.polymul(lpf.b, la_neg)
    g2 = numpy.convolve(lnum, h)
    taps2 = int(numpy.sum(numpy.abs(g2) > 1e-9 * numpy.max(numpy.abs(g2))))
    lsect = (lpf.order + 1) // 2
    now2 = 4 * lsect * 2 + 21
    merged2 = taps2 + 2 * len(scipy.signal.tf2sos([1.0], la2))
    print('(2) per detector channel and pixel: now low-pass %d sections x 4 x 2 samples + dn2 21 = %d; merged: %d-tap FIR at the 2x rate, '
          'every tap per OUTPUT pixel, + %d recursion = %d  ->  saves %d' % (lsect, now2, taps2, merged2 - taps2, merged2, now2 - merged2))
    z = up * numpy.sin(numpy.arange(2 * W) * m.backend.qam.carrier_phase_step)
    d_ref = scipy.signal.resample_poly(ff(lpf, z), 1, 2)
    d2 = 20 + lpf.shift
    full = scipy.signal.lfilter([1.0], numpy.polymul(lpf.a, la_neg), numpy.convolve(z, g2))[d2:d2 + 2 * W:2]
    print('(2) merged form, float64: interior error %.2e' % (numpy.max(numpy.abs(full - d_ref)[inner]) / numpy.max(numpy.abs(d_ref))))


main()
