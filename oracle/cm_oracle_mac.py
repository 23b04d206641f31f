# -*- coding: utf-8 -*-
"""float64 restatement of the reference's D2-MAC style time-multiplex modem - TEST INFRASTRUCTURE.

Follows /root/reference/color_modem/color/mac.py (MacModem: rows and lines of any length) and the
encoder-side wrapper of comb.py:130-167 (ColorAveragingModem) in plain numpy, with scipy's ``resample_poly`` written out
(41-tap Kaiser(5) half-band FIR, zero-extended).  Pinned against vectors the reference itself
produced (tests/golden/mac_*.npz, made by tests/golden/make_golden_mac.py) in tests/test_mac_oracle.py.  May be imported
only by tests/, tools run by hand and bench-style measurement scripts - never by color_modem_amd.
"""

import numpy

LUMA_W, CHROMA_W, LINE_W = 720, 360, 1080



def encode_components(r, g, b):
    """mac.py:28-34, term by term in the reference's operation order"""
    r, g, b = [numpy.asarray(c, dtype=numpy.float64) for c in (r, g, b)]
    luma = 0.299 * r + 0.587 * g + 0.114 * b
    dr = 0.649827 * r - 0.544149 * g - 0.105678 * b
    db = -0.219167 * r - 0.430271 * g + 0.649438 * b
    return luma, dr, db


def decode_components(luma, dr, db):
    """mac.py:36-43"""
    r = luma + 1.0787486515641855 * dr
    g = luma - 0.5494818514781797 * dr - 0.2649492993950324 * db
    b = luma + 1.364256480218281 * db
    return r, g, b


def firwin41():
    """scipy.signal.firwin(41, 0.5, window=('kaiser', 5.0)): windowed sinc, unit gain at DC."""
    n = numpy.arange(41) - 20.0
    h = 0.5 * numpy.sinc(0.5 * n) * numpy.kaiser(41, 5.0)
    return h / h.sum()


def resample_dn2(x):
    """resample_poly(x, 1, 2): y[n] = sum_k h[k] x[2 n + 20 - k], x zero outside (len(x) even)."""
    h = firwin41()
    full = numpy.convolve(x, h)          # full[m] = sum_k h[k] x[m - k]
    return full[20:20 + len(x):2]


def resample_up2(x):
    """resample_poly(x, 2, 1): y[m] = sum_k 2 h[k] xu[m + 20 - k], xu = x with a zero after every sample."""
    h = 2.0 * firwin41()
    xu = numpy.zeros(2 * len(x))
    xu[::2] = x
    return numpy.convolve(xu, h)[20:20 + 2 * len(x)]


def firwin_kaiser(half_len, cutoff):
    """scipy.signal.firwin(2 half_len + 1, cutoff, window=('kaiser', 5.0)): windowed sinc, unit gain at DC."""
    n = numpy.arange(2 * half_len + 1) - float(half_len)
    h = cutoff * numpy.sinc(cutoff * n) * numpy.kaiser(2 * half_len + 1, 5.0)
    return h / h.sum()


def resample_poly(x, up, down):
    """scipy.signal.resample_poly(x, up, down) with its defaults (window ('kaiser', 5.0), zero padding), written out:
    y[n] = sum_j up h[j] xu[n down + half_len - j],  xu = x with up - 1 zeros after every sample, h = firwin(2 half_len + 1,
    1 / max(up, down)), half_len = 10 max(up, down);  ceil(len(x) up / down) outputs."""
    import math
    g = math.gcd(int(up), int(down))
    up, down = int(up) // g, int(down) // g
    x = numpy.asarray(x, dtype=numpy.float64)
    if up == down == 1:
        return x.copy()
    half_len = 10 * max(up, down)
    h = up * firwin_kaiser(half_len, 1.0 / max(up, down))
    n_out = -(-len(x) * up // down)
    # only every up-th sample of xu is non-zero: with t0 = n down + half_len, j = t0 % up + up q meets x[t0 // up - q]
    t0 = numpy.arange(n_out) * down + half_len
    q = numpy.arange(-(-len(h) // up))
    j = (t0 % up)[:, None] + up * q[None, :]
    i = (t0 // up)[:, None] - q[None, :]
    ok = (j < len(h)) & (i >= 0) & (i < len(x))
    return numpy.sum(numpy.where(ok, h[numpy.minimum(j, len(h) - 1)] * x[numpy.clip(i, 0, len(x) - 1)], 0.0), axis=1)


def modulate_components(alternate, luma, dr, db, line_width=LINE_W):
    """mac.py:43-82: components of any length -> the 1080-sample line -> `line_width` samples."""
    assert len(luma) == len(dr) == len(db)
    chroma = resample_poly(db if alternate else dr, CHROMA_W, len(dr)) + 0.5
    luma = resample_poly(luma, LUMA_W, len(luma))
    out = _assemble(luma, chroma)
    return resample_poly(out, line_width, LINE_W)


def _assemble(luma, chroma):
    out = 0.5 * numpy.ones(LINE_W)
    out[15] = 0.4375 + 0.125 * chroma[2]
    out[16] = 0.25 + 0.5 * chroma[3]
    out[17] = 0.0625 + 0.875 * chroma[4]
    out[18:369] = chroma[5:356]
    out[369] = 0.875 * chroma[356] + 0.125 * luma[8]
    out[370] = 0.5 * chroma[357] + 0.5 * luma[9]
    out[371] = 0.125 * chroma[358] + 0.875 * luma[10]
    out[372:1071] = luma[11:710]
    out[1071] = 0.0625 + 0.875 * luma[710]
    out[1072] = 0.25 + 0.5 * luma[711]
    out[1073] = 0.4375 + 0.125 * luma[712]
    return out


def split_line(comp):
    """mac.py:88-118: (luma[720], chroma[360]) of one line (brought to 1080 samples first), before the chroma interpolation."""
    comp = resample_poly(numpy.asarray(comp, dtype=numpy.float64), LINE_W, len(comp))
    assert len(comp) == LINE_W
    luma = 0.5 * numpy.ones(LUMA_W)
    chroma = 0.5 * numpy.ones(CHROMA_W)
    luma[11:710] = comp[372:1071]
    luma[710] = (comp[1071] - 0.0625) / 0.875
    luma[711] = 2.0 * comp[1072] - 0.5
    luma[712] = 8.0 * comp[1073] - 3.5
    chroma[5:356] = comp[18:369]
    chroma[2] = 8.0 * comp[15] - 3.5
    chroma[3] = 2.0 * comp[16] - 0.5
    chroma[4] = (comp[17] - 0.0625) / 0.875
    luma[8] = 8.0 * comp[369] - 7.0 * chroma[355]
    luma[9] = 2.0 * comp[370] - chroma[355]
    luma[10] = (comp[371] - 0.125 * chroma[355]) / 0.875
    luma[0:8] = luma[8]
    luma[713:] = luma[712]
    chroma[0:1] = chroma[2]          # mac.py:113 as written: sample 1 keeps 0.5
    chroma[356] = (comp[369] - 0.125 * luma[11]) / 0.875
    chroma[357] = 2.0 * comp[370] - luma[11]
    chroma[358] = 8.0 * comp[371] - 7.0 * luma[11]
    chroma[359] = chroma[358]
    return luma, chroma


def is_alternate_line(lc, frame, line):
    """line.py:54-65 on a color_modem_amd LineConfig (itself pinned against plans.json)."""
    return lc.is_alternate_line(frame, line)


class OracleMac(object):
    """The stateful per-row protocol of MacModem, optionally inside ColorAveragingModem (averaging=True)."""

    def __init__(self, line_config, averaging=False, line_width=LINE_W):
        self.lc = line_config
        self.line_width = line_width
        self.averaging = averaging
        self.modulation_delay = 1 if averaging else 0
        self.demodulation_delay = 0
        self._last = (-1, -1, None)
        self._mod_last = (-1, -1, None)

    def demodulate(self, frame, line, comp):
        lf, ll, lc_ = self._last
        if frame != lf or line != ll + 2 or lc_ is None:
            lc_ = numpy.zeros(LUMA_W)
        luma, chroma = split_line(comp)
        up = resample_up2(chroma) - 0.5
        if not is_alternate_line(self.lc, frame, line):
            dr, db = up, lc_
        else:
            dr, db = lc_, up
        self._last = (frame, line, up)
        return decode_components(luma, dr, db)

    def modulate(self, frame, line, r, g, b):
        y, u, v = encode_components(r, g, b)
        return self.modulate_components(frame, line, y, u, v)

    def modulate_components(self, frame, line, y, u, v):
        if self.averaging:     # comb.py:141-152
            lf, ll, last = self._mod_last
            if frame != lf or line != ll + 2 or last is None:
                last = (y, u, v)
            self._mod_last = (frame, line, (y, u, v))
            y, u, v = last[0], 0.5 * (u + last[1]), 0.5 * (v + last[2])
            line = line - 2
        return modulate_components(is_alternate_line(self.lc, frame, line), y, u, v, self.line_width)


def modulate_frames(line_config, rgb, first_frame=0, averaging=False, line_width=LINE_W):
    """rgb [F, 3, H, W] -> composite [F, H, line_width] float64 through the row schedule of image.py:47-55."""
    n, _, height, _ = rgb.shape
    out = numpy.zeros((n, height, line_width))
    for f in range(n):
        m = OracleMac(line_config, averaging, line_width)
        for field in range(2):
            for y in range(field, 2 * m.modulation_delay, 2):
                m.modulate(first_frame + f, y, rgb[f, 0, y], rgb[f, 1, y], rgb[f, 2, y])
            for y in range(field, height, 2):
                iy = y + 2 * m.modulation_delay
                while iy >= height:
                    iy -= 2
                out[f, y] = m.modulate(first_frame + f, y + 2 * m.modulation_delay, rgb[f, 0, iy], rgb[f, 1, iy], rgb[f, 2, iy])
    return out


def demodulate_frames(line_config, comp, first_frame=0):
    """composite [F, H, line width] -> rgb [F, 3, H, 720] float64 through the row schedule of image.py:75-83."""
    n, height, _ = comp.shape
    out = numpy.zeros((n, 3, height, LUMA_W))
    for f in range(n):
        m = OracleMac(line_config)
        for field in range(2):
            for y in range(field, height, 2):
                out[f, :, y] = numpy.stack(m.demodulate(first_frame + f, y, comp[f, y]))
    return out
