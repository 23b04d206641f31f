# -*- coding: utf-8 -*-
"""Comb-filter wrappers (API mirror of /root/reference/color_modem/comb.py:9-167).

The wrappers carry configuration only; what they do per row is compiled into the per-line
coefficient tables of the device plan (color_modem_amd/plan.py), because every one of them is
a linear combination of the base demodulations of at most three consecutive lines.
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

    def encode_components(self, r, g, b):
        return self.backend.encode_components(r, g, b)

    def decode_components(self, y, u, v):
        return self.backend.decode_components(y, u, v)

    def _stack(self):
        inner = dict(self.backend._stack())
        if 'demod_wrapper' in inner:
            raise NotImplementedError('nested SimpleCombModem wrappers are not supported')
        if 'mod_wrapper' in inner:
            # ref comb.py:104-106: the luma strip would re-modulate through the stateful averaging encoder at line
            # - 2 (own_delay - 1); the flattened plan has no such path.  ColorAveragingModem(SimpleCombModem(x)) is the
            # supported order (the comb then re-modulates through x itself).
            raise NotImplementedError('SimpleCombModem around ColorAveragingModem is not supported; wrap the other way round')
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

    def encode_components(self, r, g, b):
        return self.backend.encode_components(r, g, b)

    def decode_components(self, y, u, v):
        return self.backend.decode_components(y, u, v)

    def _stack(self):
        inner = dict(self.backend._stack())
        if 'mod_wrapper' in inner:
            raise NotImplementedError('nested ColorAveragingModem wrappers are not supported')
        inner['mod_wrapper'] = 'color_averaging'
        return inner
