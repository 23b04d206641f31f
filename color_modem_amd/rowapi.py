# -*- coding: utf-8 -*-
"""The duck-typed Modem protocol on top of the device engine.

``modulate(frame, line, r, g, b)`` / ``demodulate(frame, line, composite)`` keep the reference's
stateful, one-row-per-call semantics: a call continues the current *run* when
``frame == last_frame and line == last_line + 2`` (ref comb.py:48,97,142; secam.py:279),
otherwise every level of the stack starts over.  Each call is evaluated on the GPU by handing
the engine the last ``depth + 1`` input rows of the run.
"""

import numpy


class _Run(object):
    __slots__ = ('frame', 'line', 'k', 'rows')

    def __init__(self):
        self.frame = -1
        self.line = -1
        self.k = -1
        self.rows = []


class RowApi(object):
    modulation_delay = 0
    demodulation_delay = 0

    def __init__(self):
        self._engine_obj = None
        self._demod_run = _Run()
        self._mod_run = _Run()

    def _engine(self):
        if self._engine_obj is None:
            from color_modem_amd import engine
            self._engine_obj = engine.Engine(self)
        return self._engine_obj

    @staticmethod
    def _advance(run, frame, line, row, depth):
        if frame != run.frame or line != run.line + 2 or run.k < 0:
            run.k = 0
            run.rows = []
        else:
            run.k += 1
        run.frame, run.line = frame, line
        run.rows.append(row)
        del run.rows[:-(depth + 1)]

    def demodulate(self, frame, line, composite):
        eng = self._engine()
        row = numpy.ascontiguousarray(composite, dtype=numpy.float32)
        if row.ndim != 1 or row.shape[0] != eng.width:
            raise ValueError('composite must be one row of %d samples' % eng.width)
        run = self._demod_run
        self._advance(run, frame, line, row, eng.demod_depth)
        n = len(run.rows)
        out = eng.demodulate_run(numpy.stack(run.rows), frame, line - 2 * (n - 1), run.k - (n - 1))
        r, g, b = out[n - 1].astype(numpy.float64)
        return r, g, b

    def modulate(self, frame, line, r, g, b):
        eng = self._engine()
        assert len(r) == len(g) == len(b)
        row = numpy.ascontiguousarray(numpy.stack([r, g, b]), dtype=numpy.float32)
        if row.shape[1] != eng.width:
            raise ValueError('r, g, b must be rows of %d samples' % eng.width)
        run = self._mod_run
        self._advance(run, frame, line, row, eng.mod_depth)
        n = len(run.rows)
        out = eng.modulate_run(numpy.stack(run.rows), frame, line - 2 * (n - 1), run.k - (n - 1))
        return out[n - 1].astype(numpy.float64)
