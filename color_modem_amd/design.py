# -*- coding: utf-8 -*-
# Ported from scipy.signal (scipy 1.15: _filter_design.py - buttap, cheb1ap, cheb2ap, besselap, lp2lp_zpk / lp2hp_zpk / lp2bp_zpk /
# lp2bs_zpk, bilinear_zpk, zpk2tf, zpk2sos with _cplxreal / _nearest_real_complex_idx, iirfilter, buttord, iirnotch, group_delay, freqz;
# _fir_filter_design.py - firwin with the Kaiser window; scipy.optimize's bounded scalar minimiser _minimize_scalar_bounded, as buttord
# uses it).  Helper names, variable names and control flow follow scipy's; what this package adds is the numpy-only packaging of the
# subset the modems request.  scipy is distributed under the BSD 3-clause licence, reproduced here as it requires:
#
#   Copyright (c) 2001-2002 Enthought, Inc. 2003-2024, SciPy Developers.
#   All rights reserved.
#
#   Redistribution and use in source and binary forms, with or without modification, are permitted provided that the following
#   conditions are met:
#   1. Redistributions of source code must retain the above copyright notice, this list of conditions and the following disclaimer.
#   2. Redistributions in binary form must reproduce the above copyright notice, this list of conditions and the following disclaimer
#      in the documentation and/or other materials provided with the distribution.
#   3. Neither the name of the copyright holder nor the names of its contributors may be used to endorse or promote products derived
#      from this software without specific prior written permission.
#
#   THIS SOFTWARE IS PROVIDED BY THE COPYRIGHT HOLDERS AND CONTRIBUTORS "AS IS" AND ANY EXPRESS OR IMPLIED WARRANTIES, INCLUDING, BUT
#   NOT LIMITED TO, THE IMPLIED WARRANTIES OF MERCHANTABILITY AND FITNESS FOR A PARTICULAR PURPOSE ARE DISCLAIMED. IN NO EVENT SHALL
#   THE COPYRIGHT OWNER OR CONTRIBUTORS BE LIABLE FOR ANY DIRECT, INDIRECT, INCIDENTAL, SPECIAL, EXEMPLARY, OR CONSEQUENTIAL DAMAGES
#   (INCLUDING, BUT NOT LIMITED TO, PROCUREMENT OF SUBSTITUTE GOODS OR SERVICES; LOSS OF USE, DATA, OR PROFITS; OR BUSINESS
#   INTERRUPTION) HOWEVER CAUSED AND ON ANY THEORY OF LIABILITY, WHETHER IN CONTRACT, STRICT LIABILITY, OR TORT (INCLUDING NEGLIGENCE
#   OR OTHERWISE) ARISING IN ANY WAY OUT OF THE USE OF THIS SOFTWARE, EVEN IF ADVISED OF THE POSSIBILITY OF SUCH DAMAGE.
"""Filter design without scipy at run time (host side, numpy only) - a port of the scipy.signal routines the path requests (see the
licence notice above).

The reference designs every filter of the path with ``scipy.signal`` at construction time (/root/reference/color_modem/
utils.py:9-64 ``FilterFunction`` + ``iirfilter`` / ``iirdesign``; ``comb.py:18-20`` ``iirnotch``; the FIR that
``resample_poly`` builds behind ``qam.py:35-57``, ``secam.py:136-149``, ``niir.py:109-145``, ``protosecam.py:83-102``,
``mac.py:49-91``).  scipy is a third-party dependency with no pinned version there; this module carries
scipy's own implementations of what those calls run - analog prototypes, frequency transformations, the bilinear transform,
Butterworth order selection (with the bounded scalar minimiser its band-stop case uses), zero / pole pairing into
second-order sections, the Kaiser-windowed low-pass - so that the product's plan constants do not move with the installed
scipy.  ``tests/test_design.py`` holds every function to scipy's result (<= 1e-13) over every design the modems request,
and ``tests/test_host_constants.py`` to the constants generated from the reference (``tests/golden/plans.json``).

Only what the path requests is built: Butterworth / Chebyshev I / Chebyshev II / Bessel (phase-normalised) prototypes,
digital designs (fs = 2, i.e. frequencies as fractions of Nyquist), low / high / band-pass / band-stop.
"""

import math

import numpy

_BTYPES = {'lowpass': 'lowpass', 'low': 'lowpass', 'lp': 'lowpass', 'l': 'lowpass',
           'highpass': 'highpass', 'high': 'highpass', 'hp': 'highpass', 'h': 'highpass',
           'bandpass': 'bandpass', 'band': 'bandpass', 'pass': 'bandpass', 'bp': 'bandpass',
           'bandstop': 'bandstop', 'bands': 'bandstop', 'stop': 'bandstop', 'bs': 'bandstop'}


# ---- analog low-pass prototypes (zeros, poles, gain) at cut-off 1 rad/s -----------------------------------------------------
def buttap(order):
    m = numpy.arange(-order + 1, order, 2)
    return numpy.array([]), -numpy.exp(1j * numpy.pi * m / (2 * order)), 1.0


def cheb1ap(order, rp):
    if order == 0:
        return numpy.array([]), numpy.array([]), 10.0 ** (-rp / 20.0)
    eps = numpy.sqrt(10.0 ** (0.1 * rp) - 1.0)
    mu = 1.0 / order * numpy.arcsinh(1.0 / eps)
    m = numpy.arange(-order + 1, order, 2)
    theta = numpy.pi * m / (2 * order)
    p = -numpy.sinh(mu + 1j * theta)
    k = numpy.prod(-p, axis=0).real
    if order % 2 == 0:
        k = k / numpy.sqrt(1.0 + eps * eps)
    return numpy.array([]), p, k


def cheb2ap(order, rs):
    if order == 0:
        return numpy.array([]), numpy.array([]), 1.0
    de = 1.0 / numpy.sqrt(10.0 ** (0.1 * rs) - 1.0)
    mu = numpy.arcsinh(1.0 / de) / order
    if order % 2:
        m = numpy.concatenate((numpy.arange(-order + 1, 0, 2), numpy.arange(2, order, 2)))
    else:
        m = numpy.arange(-order + 1, order, 2)
    z = -numpy.conjugate(1j / numpy.sin(m * numpy.pi / (2.0 * order)))
    p = -numpy.exp(1j * numpy.pi * numpy.arange(-order + 1, order, 2) / (2 * order))
    p = numpy.sinh(mu) * p.real + 1j * numpy.cosh(mu) * p.imag
    p = 1.0 / p
    k = (numpy.prod(-p, axis=0) / numpy.prod(-z, axis=0)).real
    return z, p, k


def _reverse_bessel_coefficients(order):
    """theta_N(s) = sum a_k s^k, a_k = (2N - k)! / (2^(N - k) k! (N - k)!), highest power first (integers)."""
    out = []
    for k in range(order, -1, -1):
        out.append(math.factorial(2 * order - k) // (2 ** (order - k) * math.factorial(k) * math.factorial(order - k)))
    return out


def besselap(order, norm='phase'):
    """Bessel (Thomson) prototype.  Poles = roots of the reverse Bessel polynomial (delay-normalised), scaled for ``norm``.

    The roots are polished by Newton steps in extended precision on the exact integer coefficients, so they carry float64's full
    resolution (scipy refines them on the modified Bessel function instead; the two agree to the last bits)."""
    if norm != 'phase':
        raise ValueError('only the phase-normalised prototype is requested on this path')
    if order == 0:
        return numpy.array([]), numpy.array([]), 1.0
    coef = _reverse_bessel_coefficients(order)
    roots = numpy.roots(numpy.array(coef, dtype=numpy.float64)).astype(numpy.clongdouble)
    c = [numpy.longdouble(v) for v in coef]
    d = [numpy.longdouble(v * (order - i)) for i, v in enumerate(coef[:-1])]
    for _ in range(8):
        f = numpy.zeros_like(roots)
        for v in c:
            f = f * roots + v
        g = numpy.zeros_like(roots)
        for v in d:
            g = g * roots + v
        roots = roots - f / g
    p = roots.astype(numpy.complex128)
    # conjugate pairs exactly conjugate, a real pole exactly real; order as the root finder scipy uses returns them
    # (by descending imaginary part: upper half-plane first)
    p = p[numpy.argsort(-p.imag, kind='stable')]
    half = order // 2
    for i in range(half):
        j = order - 1 - i
        re = 0.5 * (p[i].real + p[j].real)
        im = 0.5 * (p[i].imag - p[j].imag)
        p[i] = complex(re, im)
        p[j] = complex(re, -im)
    if order % 2:
        p[half] = complex(p[half].real, 0.0)
    a_last = math.factorial(2 * order) // math.factorial(order) // 2 ** order
    p = p * 10.0 ** (-math.log10(a_last) / order)
    return numpy.array([]), p, 1.0


# ---- frequency transformations of a (z, p, k) low-pass prototype ------------------------------------------------------------
def _relative_degree(z, p):
    degree = len(p) - len(z)
    if degree < 0:
        raise ValueError('improper transfer function')
    return degree


def lp2lp_zpk(z, p, k, wo=1.0):
    z, p = numpy.atleast_1d(z), numpy.atleast_1d(p)
    degree = _relative_degree(z, p)
    return wo * z, wo * p, k * wo ** degree


def lp2hp_zpk(z, p, k, wo=1.0):
    z, p = numpy.atleast_1d(z), numpy.atleast_1d(p)
    degree = _relative_degree(z, p)
    z_hp = wo / z
    p_hp = wo / p
    z_hp = numpy.append(z_hp, numpy.zeros(degree))
    return z_hp, p_hp, k * numpy.real(numpy.prod(-z) / numpy.prod(-p))


def lp2bp_zpk(z, p, k, wo=1.0, bw=1.0):
    z, p = numpy.atleast_1d(z), numpy.atleast_1d(p)
    degree = _relative_degree(z, p)
    z_lp = (z * bw / 2).astype(complex)
    p_lp = (p * bw / 2).astype(complex)
    z_bp = numpy.concatenate((z_lp + numpy.sqrt(z_lp ** 2 - wo ** 2), z_lp - numpy.sqrt(z_lp ** 2 - wo ** 2)))
    p_bp = numpy.concatenate((p_lp + numpy.sqrt(p_lp ** 2 - wo ** 2), p_lp - numpy.sqrt(p_lp ** 2 - wo ** 2)))
    z_bp = numpy.append(z_bp, numpy.zeros(degree))
    return z_bp, p_bp, k * bw ** degree


def lp2bs_zpk(z, p, k, wo=1.0, bw=1.0):
    z, p = numpy.atleast_1d(z), numpy.atleast_1d(p)
    degree = _relative_degree(z, p)
    z_hp = ((bw / 2) / z).astype(complex)
    p_hp = ((bw / 2) / p).astype(complex)
    z_bs = numpy.concatenate((z_hp + numpy.sqrt(z_hp ** 2 - wo ** 2), z_hp - numpy.sqrt(z_hp ** 2 - wo ** 2)))
    p_bs = numpy.concatenate((p_hp + numpy.sqrt(p_hp ** 2 - wo ** 2), p_hp - numpy.sqrt(p_hp ** 2 - wo ** 2)))
    z_bs = numpy.append(z_bs, numpy.full(degree, +1j * wo))
    z_bs = numpy.append(z_bs, numpy.full(degree, -1j * wo))
    return z_bs, p_bs, k * numpy.real(numpy.prod(-z) / numpy.prod(-p))


def bilinear_zpk(z, p, k, fs):
    z, p = numpy.atleast_1d(z), numpy.atleast_1d(p)
    degree = _relative_degree(z, p)
    fs2 = 2.0 * fs
    z_z = (fs2 + z) / (fs2 - z)
    p_z = (fs2 + p) / (fs2 - p)
    z_z = numpy.append(z_z, -numpy.ones(degree))
    return z_z, p_z, k * numpy.real(numpy.prod(fs2 - z) / numpy.prod(fs2 - p))


# ---- representations ---------------------------------------------------------------------------------------------------------
def _real_if_conjugate(coef, roots):
    if numpy.iscomplexobj(coef):
        roots = numpy.asarray(roots, complex)
        pos = roots[roots.imag > 0]
        neg = numpy.conjugate(roots[roots.imag < 0])
        if len(pos) == len(neg) and numpy.all(numpy.sort_complex(neg) == numpy.sort_complex(pos)):
            coef = coef.real.copy()
    return coef


def zpk2tf(z, p, k):
    z, p = numpy.atleast_1d(z), numpy.atleast_1d(p)
    b = k * numpy.poly(z)
    a = numpy.atleast_1d(numpy.poly(p))
    return _real_if_conjugate(numpy.atleast_1d(b), z), _real_if_conjugate(a, p)


def _cplxreal(z):
    """One of every conjugate pair (positive imaginary part) and the real elements, each sorted."""
    z = numpy.atleast_1d(z)
    if z.size == 0:
        return z, z
    tol = 100 * numpy.finfo((1.0 * z).dtype).eps
    z = z[numpy.lexsort((abs(z.imag), z.real))]
    real_indices = abs(z.imag) <= tol * abs(z)
    zr = z[real_indices].real
    if len(zr) == len(z):
        return numpy.array([]), zr
    z = z[~real_indices]
    zp = z[z.imag > 0]
    zn = z[z.imag < 0]
    if len(zp) != len(zn):
        raise ValueError('array contains a complex value with no matching conjugate')
    same_real = numpy.diff(zp.real) <= tol * abs(zp[:-1])
    diffs = numpy.diff(numpy.concatenate(([0], same_real, [0])))
    run_starts = numpy.nonzero(diffs > 0)[0]
    run_stops = numpy.nonzero(diffs < 0)[0]
    for start, stop in zip(run_starts, run_stops):
        for chunk in (zp[start:stop + 1], zn[start:stop + 1]):
            chunk[...] = chunk[numpy.lexsort([abs(chunk.imag)])]
    if numpy.any(abs(zp - zn.conj()) > tol * abs(zn)):
        raise ValueError('array contains a complex value with no matching conjugate')
    return (zp + zn.conj()) / 2, zr


def _nearest_real_complex_idx(fro, to, which):
    order = numpy.argsort(numpy.abs(fro - to))
    mask = numpy.isreal(fro[order])
    if which == 'complex':
        mask = ~mask
    return order[numpy.nonzero(mask)[0][0]]


def _single_zpksos(z, p, k):
    sos = numpy.zeros(6)
    b, a = zpk2tf(z, p, k)
    sos[3 - len(b):3] = b
    sos[6 - len(a):6] = a
    return sos


def zpk2sos(z, p, k):
    """Second-order sections of a digital filter, poles paired with their nearest zeros, the section whose poles lie
    closest to the unit circle last."""
    z, p = numpy.atleast_1d(z), numpy.atleast_1d(p)
    if len(z) == len(p) == 0:
        return numpy.array([[k, 0.0, 0.0, 1.0, 0.0, 0.0]])
    p = numpy.concatenate((p, numpy.zeros(max(len(z) - len(p), 0))))
    z = numpy.concatenate((z, numpy.zeros(max(len(p) - len(z), 0))))
    n_sections = (max(len(p), len(z)) + 1) // 2
    if len(p) % 2 == 1:
        p = numpy.append(p, 0)
        z = numpy.append(z, 0)
    z = numpy.concatenate(_cplxreal(z))
    p = numpy.concatenate(_cplxreal(p))
    if not numpy.isreal(k):
        raise ValueError('k must be real')
    k = k.real if hasattr(k, 'real') else k

    def idx_worst(q):
        return numpy.argmin(numpy.abs(1 - numpy.abs(q)))

    sos = numpy.zeros((n_sections, 6))
    for si in range(n_sections - 1, -1, -1):
        p1_idx = idx_worst(p)
        p1 = p[p1_idx]
        p = numpy.delete(p, p1_idx)
        if numpy.isreal(p1) and numpy.isreal(p).sum() == 0:
            # the last real pole on its own
            z1_idx = _nearest_real_complex_idx(z, p1, 'real')
            z1 = z[z1_idx]
            z = numpy.delete(z, z1_idx)
            sos[si] = _single_zpksos([z1, 0], [p1, 0], 1)
        elif len(p) + 1 == len(z) and not numpy.isreal(p1) and numpy.isreal(p).sum() == 1 and numpy.isreal(z).sum() == 1:
            # one real pole and one real zero are left besides: this complex pole must take a complex zero
            z1_idx = _nearest_real_complex_idx(z, p1, 'complex')
            z1 = z[z1_idx]
            z = numpy.delete(z, z1_idx)
            sos[si] = _single_zpksos([z1, z1.conj()], [p1, p1.conj()], 1)
        else:
            if numpy.isreal(p1):
                prealidx = numpy.flatnonzero(numpy.isreal(p))
                p2_idx = prealidx[idx_worst(p[prealidx])]
                p2 = p[p2_idx]
                p = numpy.delete(p, p2_idx)
            else:
                p2 = p1.conj()
            if len(z) > 0:
                z1_idx = numpy.argmin(numpy.abs(p1 - z))
                z1 = z[z1_idx]
                z = numpy.delete(z, z1_idx)
                if not numpy.isreal(z1):
                    sos[si] = _single_zpksos([z1, z1.conj()], [p1, p2], 1)
                elif len(z) > 0:
                    z2_idx = _nearest_real_complex_idx(z, p1, 'real')
                    z2 = z[z2_idx]
                    z = numpy.delete(z, z2_idx)
                    sos[si] = _single_zpksos([z1, z2], [p1, p2], 1)
                else:
                    sos[si] = _single_zpksos([z1], [p1, p2], 1)
            else:
                sos[si] = _single_zpksos([], [p1, p2], 1)
    assert len(p) == len(z) == 0
    sos[0][:3] *= k
    return sos


def tf2zpk(b, a):
    b = numpy.atleast_1d(numpy.asarray(b, dtype=numpy.float64))
    a = numpy.atleast_1d(numpy.asarray(a, dtype=numpy.float64))
    b = b / a[0]
    a = a / a[0]
    k = b[0]
    b = b / b[0]
    return numpy.roots(b), numpy.roots(a), k


def tf2sos(b, a):
    z, p, k = tf2zpk(b, a)
    return zpk2sos(z, p, k)


# ---- digital IIR design ------------------------------------------------------------------------------------------------------
def iirfilter(order, wn, rp=None, rs=None, btype='band', ftype='butter', output='ba'):
    """Digital IIR filter of the given order; ``wn`` as fractions of the Nyquist frequency."""
    wn = numpy.asarray(wn, dtype=numpy.float64)
    try:
        btype = _BTYPES[btype.lower()]
    except KeyError:
        raise ValueError("'%s' is an invalid bandtype for filter." % btype)
    if numpy.any(wn <= 0) or numpy.any(wn >= 1):
        raise ValueError('Digital filter critical frequencies must be 0 < Wn < 1')
    ftype = ftype.lower()
    if ftype in ('butter', 'butterworth'):
        z, p, k = buttap(order)
    elif ftype in ('bessel', 'bessel_phase'):
        z, p, k = besselap(order, norm='phase')
    elif ftype in ('cheby1', 'cheby', 'chebyshev1', 'chebyshevi'):
        if rp is None:
            raise ValueError('passband ripple (rp) must be provided to design a Chebyshev I filter.')
        z, p, k = cheb1ap(order, rp)
    elif ftype in ('cheby2', 'chebyshev2', 'chebyshevii'):
        if rs is None:
            raise ValueError('stopband attenuation (rs) must be provided to design an Chebyshev II filter.')
        z, p, k = cheb2ap(order, rs)
    else:
        raise ValueError("'%s' is not a filter type this path requests." % ftype)
    fs = 2.0
    warped = 2 * fs * numpy.tan(numpy.pi * wn / fs)
    if btype in ('lowpass', 'highpass'):
        if numpy.size(wn) != 1:
            raise ValueError('Must specify a single critical frequency Wn for lowpass or highpass filter')
        z, p, k = (lp2lp_zpk if btype == 'lowpass' else lp2hp_zpk)(z, p, k, wo=warped)
    else:
        try:
            bw = warped[1] - warped[0]
            wo = numpy.sqrt(warped[0] * warped[1])
        except IndexError:
            raise ValueError('Wn must specify start and stop frequencies for bandpass or bandstop filter')
        z, p, k = (lp2bp_zpk if btype == 'bandpass' else lp2bs_zpk)(z, p, k, wo=wo, bw=bw)
    z, p, k = bilinear_zpk(z, p, k, fs=fs)
    if output == 'zpk':
        return z, p, k
    if output == 'ba':
        return zpk2tf(z, p, k)
    if output == 'sos':
        return zpk2sos(z, p, k)
    raise ValueError("'%s' is not a valid output form." % output)


def _fminbound(func, x1, x2, xatol=1e-5, maxfun=500):
    """Bounded scalar minimiser (golden section with parabolic interpolation, Forsythe / Malcolm / Moler's FMIN): the band-stop
    case of the Butterworth order selection depends on where exactly it stops, so the step rules are the published ones."""
    sqrt_eps = math.sqrt(2.2e-16)
    golden_mean = 0.5 * (3.0 - math.sqrt(5.0))
    a, b = x1, x2
    fulc = a + golden_mean * (b - a)
    nfc, xf = fulc, fulc
    rat = e = 0.0
    x = xf
    fx = func(x)
    num = 1
    ffulc = fnfc = fx
    xm = 0.5 * (a + b)
    tol1 = sqrt_eps * abs(xf) + xatol / 3.0
    tol2 = 2.0 * tol1
    while abs(xf - xm) > (tol2 - 0.5 * (b - a)):
        golden = True
        if abs(e) > tol1:
            golden = False
            r = (xf - nfc) * (fx - ffulc)
            q = (xf - fulc) * (fx - fnfc)
            p = (xf - fulc) * q - (xf - nfc) * r
            q = 2.0 * (q - r)
            if q > 0.0:
                p = -p
            q = abs(q)
            r = e
            e = rat
            if abs(p) < abs(0.5 * q * r) and p > q * (a - xf) and p < q * (b - xf):
                rat = (p + 0.0) / q
                x = xf + rat
                if (x - a) < tol2 or (b - x) < tol2:
                    si = numpy.sign(xm - xf) + ((xm - xf) == 0)
                    rat = tol1 * si
            else:
                golden = True
        if golden:
            e = a - xf if xf >= xm else b - xf
            rat = golden_mean * e
        si = numpy.sign(rat) + (rat == 0)
        x = xf + si * max(abs(rat), tol1)
        fu = func(x)
        num += 1
        if fu <= fx:
            if x >= xf:
                a = xf
            else:
                b = xf
            fulc, ffulc = nfc, fnfc
            nfc, fnfc = xf, fx
            xf, fx = x, fu
        else:
            if x < xf:
                a = x
            else:
                b = x
            if fu <= fnfc or nfc == xf:
                fulc, ffulc = nfc, fnfc
                nfc, fnfc = x, fu
            elif fu <= ffulc or fulc == xf or fulc == nfc:
                fulc, ffulc = x, fu
        xm = 0.5 * (a + b)
        tol1 = sqrt_eps * abs(xf) + xatol / 3.0
        tol2 = 2.0 * tol1
        if num >= maxfun:
            break
    return xf


def _band_stop_order(wp, ind, passb, stopb, gpass, gstop):
    passb_c = passb.copy()
    passb_c[ind] = wp
    nat = stopb * (passb_c[0] - passb_c[1]) / (stopb ** 2 - passb_c[0] * passb_c[1])
    nat = min(abs(nat))
    g_stop = 10 ** (0.1 * abs(gstop))
    g_pass = 10 ** (0.1 * abs(gpass))
    return numpy.log10((g_stop - 1.0) / (g_pass - 1.0)) / (2 * numpy.log10(nat))


def buttord(wp, ws, gpass, gstop):
    """Lowest order and natural frequency of a digital Butterworth filter losing at most gpass dB in the pass band and at
    least gstop dB in the stop band (band edges as fractions of Nyquist).  No validation of the edges: the reference's NTSC
    set-up hands over a negative stop-band edge, which the scipy release it was written against accepted."""
    wp = numpy.atleast_1d(numpy.asarray(wp, dtype=numpy.float64))
    ws = numpy.atleast_1d(numpy.asarray(ws, dtype=numpy.float64))
    filter_type = 2 * (len(wp) - 1) + 1
    if wp[0] >= ws[0]:
        filter_type += 1
    passb = numpy.tan(numpy.pi * wp / 2.0)
    stopb = numpy.tan(numpy.pi * ws / 2.0)
    if filter_type == 1:
        nat = stopb / passb
    elif filter_type == 2:
        nat = passb / stopb
    elif filter_type == 3:
        passb[0] = _fminbound(lambda w: _band_stop_order(w, 0, passb, stopb, gpass, gstop), passb[0], stopb[0] - 1e-12)
        passb[1] = _fminbound(lambda w: _band_stop_order(w, 1, passb, stopb, gpass, gstop), stopb[1] + 1e-12, passb[1])
        nat = (stopb * (passb[0] - passb[1])) / (stopb ** 2 - passb[0] * passb[1])
    else:
        nat = (stopb ** 2 - passb[0] * passb[1]) / (stopb * (passb[0] - passb[1]))
    nat = min(abs(nat))
    g_stop = 10 ** (0.1 * abs(gstop))
    g_pass = 10 ** (0.1 * abs(gpass))
    order = int(numpy.ceil(numpy.log10((g_stop - 1.0) / (g_pass - 1.0)) / (2 * numpy.log10(nat))))
    try:
        w0 = (g_pass - 1.0) ** (-1.0 / (2.0 * order))
    except ZeroDivisionError:
        w0 = 1.0
    if filter_type == 1:
        w_n = w0 * passb
    elif filter_type == 2:
        w_n = passb / w0
    elif filter_type == 3:
        w_n = numpy.zeros(2)
        discr = numpy.sqrt((passb[1] - passb[0]) ** 2 + 4 * w0 ** 2 * passb[0] * passb[1])
        w_n[0] = ((passb[1] - passb[0]) + discr) / (2 * w0)
        w_n[1] = ((passb[1] - passb[0]) - discr) / (2 * w0)
        w_n = numpy.sort(abs(w_n))
    else:
        w0 = numpy.array([-w0, w0])
        w_n = -w0 * (passb[1] - passb[0]) / 2.0 + numpy.sqrt(w0 ** 2 / 4.0 * (passb[1] - passb[0]) ** 2 + passb[0] * passb[1])
        w_n = numpy.sort(abs(w_n))
    wn = (2.0 / numpy.pi) * numpy.arctan(w_n)
    if len(wn) == 1:
        wn = wn[0]
    return order, wn


def iirnotch(w0, q):
    """Second-order notch at w0 (fraction of Nyquist) with quality factor q (comb.py:18-20)."""
    w0 = float(w0)
    if w0 <= 0.0 or w0 >= 1.0:
        raise ValueError('w0 should be such that 0 < w0 < 1')
    bw = w0 / q * numpy.pi
    w0 = w0 * numpy.pi
    gb = 1 / numpy.sqrt(2)
    beta = (numpy.sqrt(1.0 - gb ** 2.0) / gb) * numpy.tan(bw / 2.0)
    gain = 1.0 / (1.0 + beta)
    b = gain * numpy.array([1.0, -2.0 * numpy.cos(w0), 1.0])
    a = numpy.array([1.0, -2.0 * gain * numpy.cos(w0), 2.0 * gain - 1.0])
    return b, a


# ---- responses ---------------------------------------------------------------------------------------------------------------
def freqz_at(b, a, w):
    """H(e^{j pi w}) of b / a at one frequency w (fraction of Nyquist)."""
    zm1 = numpy.exp(-1j * numpy.pi * w)
    return numpy.polynomial.polynomial.polyval(zm1, numpy.asarray(b, dtype=numpy.float64)) / \
        numpy.polynomial.polynomial.polyval(zm1, numpy.asarray(a, dtype=numpy.float64))


def group_delay_at(b, a, w):
    """Group delay in samples of b / a at one frequency w (fraction of Nyquist)."""
    b = numpy.atleast_1d(numpy.asarray(b, dtype=numpy.float64))
    a = numpy.atleast_1d(numpy.asarray(a, dtype=numpy.float64))
    c = numpy.convolve(b, a[::-1])
    cr = c * numpy.arange(c.size)
    z = numpy.exp(-1j * numpy.pi * w)
    num = numpy.polyval(cr[::-1], z)
    den = numpy.polyval(c[::-1], z)
    return float(numpy.real(num / den)) - a.size + 1


# ---- the FIR resample_poly builds ----------------------------------------------------------------------------------------------
def _i0(x):
    """Modified Bessel function of the first kind, order 0, by its power series (arguments of a few units)."""
    x = numpy.asarray(x, dtype=numpy.float64)
    q = 0.25 * x * x
    term = numpy.ones_like(x)
    total = numpy.ones_like(x)
    for k in range(1, 200):
        term = term * q / (k * k)
        total = total + term
        if numpy.all(term <= 1e-18 * total):
            break
    return total


def kaiser(m, beta):
    n = numpy.arange(0, m)
    alpha = (m - 1) / 2.0
    return _i0(beta * numpy.sqrt(1 - ((n - alpha) / alpha) ** 2.0)) / _i0(beta)


def firwin_lowpass_kaiser(numtaps, cutoff, beta):
    """Windowed-sinc low-pass with unit gain at DC: firwin(numtaps, cutoff, window=('kaiser', beta))."""
    alpha = 0.5 * (numtaps - 1)
    m = numpy.arange(0, numtaps) - alpha
    h = cutoff * numpy.sinc(cutoff * m)
    h = h * kaiser(numtaps, beta)
    return h / numpy.sum(h)


def resample_poly_fir(max_rate):
    """The filter resample_poly(x, up, down) designs by default, before its scaling by ``up``: 20 * max(up, down) + 1 taps,
    cut-off 1 / max(up, down), Kaiser window with beta = 5."""
    return firwin_lowpass_kaiser(2 * 10 * max_rate + 1, 1.0 / max_rate, 5.0)
