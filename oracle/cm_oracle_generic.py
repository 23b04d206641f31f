# -*- coding: utf-8 -*-
"""CPU oracle of the NESTED stacks (test infrastructure: only tests/, smoke() and bench.py may load anything under oracle/).

The reference's wrappers work around any backend that has demodulate_components / modulate_components (ref comb.py:90-113, 131-155).
The C++ oracle (cm_oracle.cpp) restates one wrapper around one decoder; this module restates the two wrapper classes THEMSELVES in
Python, float64, one numpy row per call, statement for statement - so they nest the way the reference's do - around oracle objects of
the leaves: cm_oracle.OracleModem (QAM / SECAM families; OraclePal3DCallable for Pal3DModem(avg=f)) and cm_oracle_am.NiirOracle.
Pinned by tests/golden/nested_*.npz (tests/golden/make_golden_nested.py runs the reference on them).
"""

import numpy

from oracle import cm_oracle, cm_oracle_am


# encode / decode_components of the leaves, term by term as the reference writes them
def _pal_encode(r, g, b):                                                            # pal.py:33-38
    r, g, b = [numpy.asarray(c, dtype=numpy.float64) for c in (r, g, b)]
    return (0.299 * r + 0.587 * g + 0.114 * b, -0.147407 * r - 0.289391 * g + 0.436798 * b, 0.614777 * r - 0.514799 * g - 0.099978 * b)


def _ntsc_encode(r, g, b):                                                           # ntsc.py:28-33
    r, g, b = [numpy.asarray(c, dtype=numpy.float64) for c in (r, g, b)]
    return (0.3 * r + 0.59 * g + 0.11 * b, -0.1476019510016258 * r - 0.2893575108184752 * g + 0.436959461820101 * b,
            0.6183717846575098 * r - 0.5185533057776567 * g - 0.099818478879853 * b)


def _secam_encode(r, g, b):                                                          # secam.py:193-200
    r, g, b = [numpy.asarray(c, dtype=numpy.float64) for c in (r, g, b)]
    return (0.299 * r + 0.587 * g + 0.114 * b, -1.333302 * r + 1.116474 * g + 0.216828 * b, -0.449995 * r - 0.883435 * g + 1.33343 * b)


def _secam_decode(luma, dr, db):                                                     # secam.py:202-208
    luma, dr, db = [numpy.asarray(c, dtype=numpy.float64) for c in (luma, dr, db)]
    return luma - 0.5257623554153522 * dr, luma + 0.2678074007993021 * dr - 0.1290417517983779 * db, luma + 0.6644518272425249 * db


class _Leaf(object):
    """An oracle object of a leaf with the members a wrapper calls (comb.py:90-94, 98, 105, 112-122, 152-167)."""

    def __init__(self, modem):
        stack = modem._stack()
        kind = stack['kind']
        self.kind = kind
        if kind in ('niir', 'protosecam'):
            self.o = cm_oracle_am.make(modem)
            self.encode_components, self.decode_components = self.o.encode_components, self.o.decode_components
            self.demodulation_delay = 0
        else:
            self.o = cm_oracle.OracleModem(modem)
            if kind in ('pal_s', 'pal_d', 'pal_3d'):
                self.encode_components, self.decode_components = _pal_encode, cm_oracle.OracleCallableComb._decode_pal
            elif kind in ('ntsc', 'ntsc_comb'):
                self.encode_components, self.decode_components = _ntsc_encode, cm_oracle.OracleCallableComb._decode_ntsc
            elif kind == 'secam':
                self.encode_components, self.decode_components = _secam_encode, _secam_decode
            else:
                raise NotImplementedError(kind)
            self.demodulation_delay = int(self.o.demodulation_delay)
        self.modulation_delay = int(getattr(self.o, 'modulation_delay', 0))

    def demodulate_components(self, frame, line, composite, strip_chroma=True):
        return self.o.demodulate_components(frame, line, composite, strip_chroma)

    def modulate_components(self, frame, line, y, u, v):
        return self.o.modulate_components(frame, line, y, u, v)

    def modulate(self, frame, line, r, g, b):
        return self.o.modulate(frame, line, r, g, b)

    def demodulate(self, frame, line, composite):
        return self.o.demodulate(frame, line, composite)


class SimpleComb(object):
    """comb.py:71-122 (Simple3DCombModem: delay=True, comb.py:125-127)"""

    def __init__(self, backend, own_delay, avg, notch):
        self.backend = backend
        self._own_delay = own_delay                                                   # comb.py:74
        self.modulation_delay = getattr(backend, 'modulation_delay', 0)              # comb.py:75
        self.demodulation_delay = getattr(backend, 'demodulation_delay', 0) + own_delay   # comb.py:76
        self._last_frame = self._last_line = -1
        self._last_demodulated = None
        self._avg = avg
        self._notch = notch                                                           # a FilterFunction record (b, a, shift) or None

    def _apply_notch(self, y):
        return cm_oracle.OracleCallableComb._apply_notch(self, y)                     # utils.py:28-36

    def modulate_components(self, frame, line, y, u, v):
        return self.backend.modulate_components(frame, line, y, u, v)                 # comb.py:90-91

    def modulate(self, frame, line, r, g, b):
        return self.backend.modulate(frame, line, r, g, b)                            # comb.py:93-94

    def demodulate_components(self, frame, line, composite, strip_chroma=True):
        composite = numpy.asarray(composite, dtype=numpy.float64)
        curr = self.backend.demodulate_components(frame, line, composite, strip_chroma=False)   # comb.py:98 / 101
        if frame != self._last_frame or line != self._last_line + 2:                  # comb.py:97
            y, u, v = curr
        else:
            y = self._last_demodulated[0] if self._own_delay else curr[0]             # comb.py:102
            u = numpy.asarray(self._avg(self._last_demodulated[1], curr[1]), dtype=numpy.float64)   # comb.py:103
            v = numpy.asarray(self._avg(self._last_demodulated[2], curr[2]), dtype=numpy.float64)   # comb.py:104
            if strip_chroma:                                                          # comb.py:105-110
                y = y - self.backend.modulate_components(frame, line - 2 * (self._own_delay - self.modulation_delay),
                                                         numpy.zeros(len(composite)), u, v)
                if self._notch is not None:
                    y = self._apply_notch(y)
        self._last_frame, self._last_line, self._last_demodulated = frame, line, curr
        return y, u, v

    def encode_components(self, r, g, b):
        return self.backend.encode_components(r, g, b)                                # comb.py:115-116

    def decode_components(self, y, u, v):
        return self.backend.decode_components(y, u, v)                                # comb.py:118-119

    def demodulate(self, frame, line, composite):
        return self.backend.decode_components(*self.demodulate_components(frame, line, composite))   # comb.py:121-122


class ColorAveraging(object):
    """comb.py:130-167"""

    def __init__(self, backend):
        self.backend = backend
        self.modulation_delay = getattr(backend, 'modulation_delay', 0) + 1          # comb.py:133
        self.demodulation_delay = getattr(backend, 'demodulation_delay', 0)          # comb.py:134
        self._last_modulated_frame = self._last_modulated_line = -1
        self._last_y = self._last_u = self._last_v = None

    def modulate_components(self, frame, line, y, u, v):
        y, u, v = [numpy.asarray(c, dtype=numpy.float64) for c in (y, u, v)]
        if frame != self._last_modulated_frame or line != self._last_modulated_line + 2 \
                or self._last_u is None or self._last_v is None:                      # comb.py:142-146
            self._last_y, self._last_u, self._last_v = y, u, v
        self._last_y, y = y, self._last_y                                             # comb.py:147
        self._last_u, u = u, 0.5 * (u + self._last_u)                                 # comb.py:148
        self._last_v, v = v, 0.5 * (v + self._last_v)                                 # comb.py:149
        self._last_modulated_frame, self._last_modulated_line = frame, line
        return self.backend.modulate_components(frame, line - 2, y, u, v)             # comb.py:152

    def modulate(self, frame, line, r, g, b):
        return self.modulate_components(frame, line, *self.backend.encode_components(r, g, b))   # comb.py:154-155

    def demodulate_components(self, *args, **kwargs):
        return self.backend.demodulate_components(*args, **kwargs)                    # comb.py:157-158

    def demodulate(self, *args, **kwargs):
        return self.backend.demodulate(*args, **kwargs)                               # comb.py:160-161

    def encode_components(self, r, g, b):
        return self.backend.encode_components(r, g, b)

    def decode_components(self, y, u, v):
        return self.backend.decode_components(y, u, v)


def make(modem):
    """oracle object tree of a color_modem_amd modem stack (the wrappers restated here, the leaves by their own oracles)"""
    from color_modem_amd import comb
    if isinstance(modem, comb.SimpleCombModem):
        notch = None
        if modem._notch is not None:      # the oracle's own design (comb.py:18-20 via cm_oracle_design): from the product only q, fsc and fs
            from oracle import cm_oracle_design as design
            leaf = comb._qam_backend(modem.backend)
            notch = design.notch(leaf.config.fsc, leaf.line_config.fs, modem._notch.q)
        return SimpleComb(make(modem.backend), int(modem._own_delay), modem._avg, notch)
    if isinstance(modem, comb.ColorAveragingModem):
        return ColorAveraging(make(modem.backend))
    return _Leaf(modem)


# ---- frames: the row schedule of image.py:47-55, 75-83 ------------------------------------------------------------
def demodulate_frames(modem, comp, first_frame=0):
    comp = numpy.asarray(comp, dtype=numpy.float64)
    n, height, width = comp.shape
    out = numpy.zeros((n, 3, height, width))
    for i in range(n):
        orc = make(modem)
        delay = getattr(orc, 'demodulation_delay', 0)
        frame = first_frame + i
        for field in range(2):
            for y in range(field, 2 * delay, 2):
                orc.demodulate(frame, y, comp[i, y])
            for y in range(field, height, 2):
                iy = y + 2 * delay
                while iy >= height:
                    iy -= 2
                out[i, 0, y], out[i, 1, y], out[i, 2, y] = orc.demodulate(frame, y + 2 * delay, comp[i, iy])
    return out


def modulate_frames(modem, rgb, first_frame=0):
    rgb = numpy.asarray(rgb, dtype=numpy.float64)
    n, _, height, width = rgb.shape
    out = numpy.zeros((n, height, width))
    for i in range(n):
        orc = make(modem)
        delay = getattr(orc, 'modulation_delay', 0)
        frame = first_frame + i
        for field in range(2):
            for y in range(field, 2 * delay, 2):
                orc.modulate(frame, y, rgb[i, 0, y], rgb[i, 1, y], rgb[i, 2, y])
            for y in range(field, height, 2):
                iy = y + 2 * delay
                while iy >= height:
                    iy -= 2
                out[i, y] = orc.modulate(frame, y + 2 * delay, rgb[i, 0, iy], rgb[i, 1, iy], rgb[i, 2, iy])
    return out
