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
        self._engines = {}
        self._demod_run = _Run()
        self._mod_run = _Run()

    def _engine(self, components=False, strip_chroma=True):
        key = (bool(components), bool(strip_chroma))
        if key not in self._engines:
            from color_modem_amd import engine
            self._engines[key] = engine.make_engine(self, components=key[0], strip_chroma=key[1])
        return self._engines[key]

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
        return self._demodulate(self._engine(), frame, line, composite)

    def demodulate_components(self, frame, line, composite, strip_chroma=True):
        """(y, u, v) of one row (ref qam.py:43-58 behind pal.py:54-59 / ntsc.py:47-49, comb.py:47-59, 96-113,
        pal.py:180-234); shares the run state with demodulate(), which is this followed by decode_components."""
        eng = self._engine(True, strip_chroma)
        if not strip_chroma and eng.built.desc.first_is_plain:
            # the first line of a run is the backend's own unstripped decode (comb.py:48-49); the plain pass of
            # the comb's plan only exists with band-stop luma, so that one call goes to the backend's plan
            run = self._demod_run
            if frame != run.frame or line != run.line + 2 or run.k < 0:
                row = numpy.ascontiguousarray(composite, dtype=numpy.float32)
                self._advance(run, frame, line, row, eng.demod_depth)
                self.backend._demod_run = _Run()
                return self.backend.demodulate_components(frame, line, composite, strip_chroma=False)
        return self._demodulate(eng, frame, line, composite)

    def _demodulate(self, eng, frame, line, composite):
        row = numpy.ascontiguousarray(composite, dtype=numpy.float32)
        if row.ndim != 1 or row.shape[0] != eng.comp_width:
            raise ValueError('composite must be one row of %d samples' % eng.comp_width)
        run = self._demod_run
        self._advance(run, frame, line, row, eng.demod_depth)
        n = len(run.rows)
        out = eng.demodulate_run(numpy.stack(run.rows), frame, line - 2 * (n - 1), run.k - (n - 1))
        r, g, b = out[n - 1].astype(numpy.float64)
        return r, g, b

    def modulate(self, frame, line, r, g, b):
        return self._modulate(self._engine(), frame, line, r, g, b)

    def modulate_components(self, frame, line, y, u, v):
        """Composite row from (y, u, v) / (luma, dr, db) (ref qam.py:28-32 behind pal.py:48-52 / ntsc.py:43-45,
        comb.py:141-152, secam.py:258-276); shares the run state with modulate()."""
        return self._modulate(self._engine(True, True), frame, line, y, u, v)

    def _modulate(self, eng, frame, line, r, g, b):
        assert len(r) == len(g) == len(b)
        row = numpy.ascontiguousarray(numpy.stack([r, g, b]), dtype=numpy.float32)
        if row.shape[1] != eng.in_width:
            raise ValueError('r, g, b must be rows of %d samples' % eng.in_width)
        run = self._mod_run
        self._advance(run, frame, line, row, eng.mod_depth)
        n = len(run.rows)
        out = eng.modulate_run(numpy.stack(run.rows), frame, line - 2 * (n - 1), run.k - (n - 1))
        return out[n - 1].astype(numpy.float64)
