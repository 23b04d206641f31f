# -*- coding: utf-8 -*-
"""Comb-filter wrappers (API mirror of /root/reference/color_modem/comb.py:9-167).

The wrappers carry configuration only; what they do per row is compiled into the per-line
coefficient tables of the device plan (color_modem_amd/plan.py), because every one of them is
a linear combination of the base demodulations of at most three consecutive lines.

Round 6: stackings those tables do not express - a wrapper inside a wrapper, SimpleCombModem around ColorAveragingModem, wrappers
around Pal3DModem(avg=f) or the NIIR modems (color_modem_amd/generic.py: needs_generic) - run level by level: the per-row protocol
as the reference's own statements (ref comb.py:96-113, 141-155) on float64 numpy rows around the backend object's per-row protocol
(the ``_generic`` branches below), the frame entry points through generic.py's engines.
"""

import numpy

from color_modem_amd import utils
from color_modem_amd.rowapi import RowApi


def avg(val1, val2):
    return 0.5 * (val1 + val2)


def minavg(val1, val2):
    sign = (1.0 - numpy.signbit(val1)) - numpy.signbit(val2)
    return sign * numpy.minimum(numpy.abs(val1), numpy.abs(val2))


def _notch(qam_modem, q):
    return utils.notch(qam_modem, q)


def _qam_backend(modem):
    while not hasattr(modem, 'qam'):
        modem = modem.backend
    return modem


class AbstractCombModem(RowApi):
    """2-line comb scaffold around a QAM backend (ref comb.py:23-68)."""

    def __init__(self, backend, notch=0.0):
        RowApi.__init__(self)
        self.backend = backend
        self.notch = _notch(backend, notch) if notch else None     # ref comb.py:29-31

    @property
    def config(self):
        return self.backend.config

    @property
    def line_config(self):
        return self.backend.line_config

    def encode_components(self, r, g, b):
        return self.backend.encode_components(r, g, b)

    def decode_components(self, y, u, v):
        return self.backend.decode_components(y, u, v)


class SimpleCombModem(RowApi):
    """Average the chroma of consecutive lines of a field (ref comb.py:71-122)."""

    def __init__(self, backend, notch=0.0, avg=None, delay=False):
        RowApi.__init__(self)
        self.backend = backend
        self._notch = _notch(_qam_backend(backend), notch) if notch else None   # ref comb.py:86-88
        self._own_delay = 1 if delay else 0
        self.modulation_delay = getattr(backend, 'modulation_delay', 0)
        self.demodulation_delay = getattr(backend, 'demodulation_delay', 0) + self._own_delay
        self._avg = avg if avg is not None else globals()['avg']
        self._is_generic = None
        self._last_frame = self._last_line = -1          # the literal per-row branch (comb.py:77-79)
        self._last_demodulated = None

    def encode_components(self, r, g, b):
        return self.backend.encode_components(r, g, b)

    def decode_components(self, y, u, v):
        return self.backend.decode_components(y, u, v)

    # ---- level-by-level stacks (generic.needs_generic): ref comb.py:90-122 as written, around the backend object -----------------
    def _generic(self):
        if self._is_generic is None:
            from color_modem_amd import generic
            self._is_generic = generic.needs_generic(self)
        return self._is_generic

    def demodulate_components(self, frame, line, composite, strip_chroma=True):
        if not self._generic():
            return RowApi.demodulate_components(self, frame, line, composite, strip_chroma)
        composite = numpy.asarray(composite, dtype=numpy.float64)
        curr = tuple(numpy.asarray(c, dtype=numpy.float64) for c in
                     self.backend.demodulate_components(frame, line, composite, strip_chroma=False))          # comb.py:98 / 101
        if frame != self._last_frame or line != self._last_line + 2 or self._last_demodulated is None:        # comb.py:97
            y, u, v = curr
        else:
            y = self._last_demodulated[0] if self._own_delay else curr[0]                                     # comb.py:102
            u = numpy.asarray(self._avg(self._last_demodulated[1], curr[1]), dtype=numpy.float64)             # comb.py:103
            v = numpy.asarray(self._avg(self._last_demodulated[2], curr[2]), dtype=numpy.float64)             # comb.py:104
            if strip_chroma:                                                                                  # comb.py:105-110
                y = y - numpy.asarray(self.backend.modulate_components(frame, line - 2 * (self._own_delay - self.modulation_delay),
                                                                       numpy.zeros(len(composite)), u, v), dtype=numpy.float64)
                if self._notch:
                    y = numpy.asarray(self._notch(y), dtype=numpy.float64)
        self._last_frame, self._last_line, self._last_demodulated = frame, line, curr
        return y, u, v

    def demodulate(self, frame, line, composite):
        if not self._generic():
            return RowApi.demodulate(self, frame, line, composite)
        return self.backend.decode_components(*self.demodulate_components(frame, line, composite))            # comb.py:121-122

    def demodulate_rows(self, frame, line, composite_rows):
        if not self._generic():
            return RowApi.demodulate_rows(self, frame, line, composite_rows)
        rows = numpy.asarray(composite_rows)
        return numpy.array([numpy.stack(self.demodulate(frame, line + 2 * i, rows[i])) for i in range(len(rows))]).reshape(len(rows), 3, -1)

    def modulate(self, frame, line, r, g, b):
        if not self._generic():
            return RowApi.modulate(self, frame, line, r, g, b)
        return self.backend.modulate(frame, line, r, g, b)                                                    # comb.py:93-94

    def modulate_components(self, frame, line, y, u, v):
        if not self._generic():
            return RowApi.modulate_components(self, frame, line, y, u, v)
        return self.backend.modulate_components(frame, line, y, u, v)                                         # comb.py:90-91

    def modulate_rows(self, frame, line, r, g, b):
        if not self._generic():
            return RowApi.modulate_rows(self, frame, line, r, g, b)
        return self.backend.modulate_rows(frame, line, r, g, b)

    def _stack(self):
        inner = dict(self.backend._stack())
        if 'demod_wrapper' in inner or 'mod_wrapper' in inner:
            # a wrapper inside this one (ref comb.py:105 anticipates it: the strip line carries the backend's modulation_delay): no flattened
            # plan expresses that - engine.make_engine sends such stacks to generic.py before it asks for this description
            raise NotImplementedError('this stack runs level by level (color_modem_amd/generic.py), it has no flattened description')
        # avg= callables other than comb.avg / comb.minavg (ref comb.py:72, 81-84): the composition of wrapped.py applies them to the
        # component planes of consecutive calls between its two kernels (engine.make_engine routes the stack there)
        inner['demod_wrapper'] = 'simple_3d' if self._own_delay else 'simple'
        inner['wrapper_notch'] = self._notch
        inner['wrapper_avg'] = self._avg
        return inner


class Simple3DCombModem(SimpleCombModem):
    def __init__(self, backend, notch=0.0, avg=None):
        super(Simple3DCombModem, self).__init__(backend, notch, avg, True)


class ColorAveragingModem(RowApi):
    """Encoder-side averaging of the chroma of consecutive lines (ref comb.py:130-167)."""

    def __init__(self, backend):
        RowApi.__init__(self)
        self.backend = backend
        self.modulation_delay = getattr(backend, 'modulation_delay', 0) + 1
        self.demodulation_delay = getattr(backend, 'demodulation_delay', 0)
        self._is_generic = None
        self._last_modulated_frame = self._last_modulated_line = -1      # the literal per-row branch (comb.py:134-138)
        self._last_y = self._last_u = self._last_v = None

    def encode_components(self, r, g, b):
        return self.backend.encode_components(r, g, b)

    def decode_components(self, y, u, v):
        return self.backend.decode_components(y, u, v)

    # ---- level-by-level stacks (generic.needs_generic): ref comb.py:141-167 as written, around the backend object ----------------
    def _generic(self):
        if self._is_generic is None:
            from color_modem_amd import generic
            self._is_generic = generic.needs_generic(self)
        return self._is_generic

    def modulate_components(self, frame, line, y, u, v):
        if not self._generic():
            return RowApi.modulate_components(self, frame, line, y, u, v)
        y, u, v = [numpy.asarray(c, dtype=numpy.float64) for c in (y, u, v)]
        if frame != self._last_modulated_frame or line != self._last_modulated_line + 2 \
                or self._last_u is None or self._last_v is None:                                              # comb.py:142-146
            self._last_y, self._last_u, self._last_v = y, u, v
        self._last_y, y = y, self._last_y                                                                     # comb.py:147
        self._last_u, u = u, 0.5 * (u + self._last_u)                                                         # comb.py:148
        self._last_v, v = v, 0.5 * (v + self._last_v)                                                         # comb.py:149
        self._last_modulated_frame, self._last_modulated_line = frame, line
        return self.backend.modulate_components(frame, line - 2, y, u, v)                                     # comb.py:152

    def modulate(self, frame, line, r, g, b):
        if not self._generic():
            return RowApi.modulate(self, frame, line, r, g, b)
        return self.modulate_components(frame, line, *self.backend.encode_components(r, g, b))                # comb.py:154-155

    def modulate_rows(self, frame, line, r, g, b):
        if not self._generic():
            return RowApi.modulate_rows(self, frame, line, r, g, b)
        return numpy.array([self.modulate(frame, line + 2 * i, r[i], g[i], b[i]) for i in range(len(r))]).reshape(len(r), -1)

    def demodulate_components(self, frame, line, composite, strip_chroma=True):
        if not self._generic():
            return RowApi.demodulate_components(self, frame, line, composite, strip_chroma)
        return self.backend.demodulate_components(frame, line, composite, strip_chroma)                       # comb.py:157-158

    def demodulate(self, frame, line, composite):
        if not self._generic():
            return RowApi.demodulate(self, frame, line, composite)
        return self.backend.demodulate(frame, line, composite)                                                # comb.py:160-161

    def demodulate_rows(self, frame, line, composite_rows):
        if not self._generic():
            return RowApi.demodulate_rows(self, frame, line, composite_rows)
        return self.backend.demodulate_rows(frame, line, composite_rows)

    def _stack(self):
        inner = dict(self.backend._stack())
        if 'mod_wrapper' in inner:
            raise NotImplementedError('this stack runs level by level (color_modem_amd/generic.py), it has no flattened description')
        inner['mod_wrapper'] = 'color_averaging'
        return inner
