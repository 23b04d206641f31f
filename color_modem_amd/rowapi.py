# -*- coding: utf-8 -*-
"""The duck-typed Modem protocol on top of the device engine.

``modulate(frame, line, r, g, b)`` / ``demodulate(frame, line, composite)`` keep the reference's
stateful, one-row-per-call semantics: a call continues the current *run* when
``frame == last_frame and line == last_line + 2`` (ref comb.py:48,97,142; secam.py:279),
otherwise every level of the stack starts over.  Each call is evaluated on the GPU on the last
``depth + 1`` input rows of the run, which stay on the device between calls (engine.RowSession: one row up, one
result row down per call).
"""

import itertools

import numpy


class _Run(object):
    __slots__ = ('frame', 'line', 'k', 'rows', 'token', 'sessions')
    _tokens = itertools.count(1)        # run identities (next() is atomic under the GIL)

    def __init__(self):
        self.frame = -1
        self.line = -1
        self.k = -1
        self.rows = []          # the run's last depth + 1 input rows (host copies: a session that missed calls needs them)
        self.token = None       # identity of the current run
        self.sessions = {}      # (engine id, direction) -> engine.RowSession holding the run's rows on the device


class RowApi(object):
    modulation_delay = 0
    demodulation_delay = 0

    def __init__(self):
        self._engines = {}
        self._demod_run = _Run()
        self._mod_run = _Run()
        self._small_batch = None

    def set_small_batch(self, mode):
        """Pin the kernel family of the per-row protocol ('auto', 'rows', 'scan': Engine.set_small_batch) on every engine this modem has
        made and will make (a test / diagnosis aid: results differ between the families at float32 resolution only)."""
        self._small_batch = None if mode == 'auto' else mode
        for eng in self._engines.values():
            eng.set_small_batch(mode)

    ABOVE = 8      # lines above the picture an encoder run may start at (a ColorAveragingModem calls its backend at line - 2, comb.py:152)

    def _engine(self, components=False, strip_chroma=True, line=0, above=False):
        """The engine of this stack for the given protocol flavour; its per-line tables are grown (the plan is rebuilt)
        when a call names a line beyond them - the reference takes any line number (line.py:57-65).  above: an encoder whose run starts
        above the picture (a negative line number: this modem sits inside a ColorAveragingModem that runs level by level, generic.py)."""
        key = (bool(components), bool(strip_chroma)) + (('above',) if above else ())
        eng = self._engines.get(key)
        if eng is None or line >= getattr(eng, 'n_lines', 1 << 30):
            from color_modem_amd import engine
            need = 0 if eng is None else max(2 * eng.n_lines, line + 64)
            if eng is not None:      # the replaced engine's device sessions (history buffer, pinned staging, its plans) go with it - also the
                gone = (id(eng), id(getattr(eng, 'encoder', None)))      # one of a comb wrapper's encoder, which _step keys on that engine
                for run in (self._demod_run, self._mod_run):
                    for skey in [k for k in run.sessions if k[0] in gone]:
                        del run.sessions[skey]
            eng = self._engines[key] = engine.make_engine(self, components=key[0], strip_chroma=key[1], min_lines=max(need, line + 1),
                                                          line_offset=self.ABOVE if above else 0)
            if self._small_batch is not None:
                eng.set_small_batch(self._small_batch)
        return eng

    @staticmethod
    def _advance(run, frame, line, row, depth):
        if frame != run.frame or line != run.line + 2 or run.k < 0:
            run.k = 0
            run.rows = []
            run.token = next(_Run._tokens)
        else:
            run.k += 1
        run.frame, run.line = frame, line
        run.rows.append(row)
        del run.rows[:-(depth + 1)]

    @staticmethod
    def _step(run, eng, direction, frame, line):
        from color_modem_amd import engine
        while direction == 'mod' and getattr(eng, 'encoder', None) is not None:
            eng = eng.encoder          # a comb wrapper encodes through its backend (comb.py:90-94): that engine's own session
        if getattr(eng, 'composite', False) or (direction == 'mod' and getattr(eng, 'composite_mod', False)):
            # a composition of kernels (wrapped.py) / an encoder with per-call host input (NIIR noise): the run's last rows go up as they are
            n = len(run.rows)
            fn = eng.demodulate_run if direction == 'demod' else eng.modulate_run
            return numpy.asarray(fn(numpy.stack(run.rows), frame, line - 2 * (n - 1), run.k - (n - 1))[n - 1], dtype=numpy.float64)
        key = (id(eng), direction)
        if key not in run.sessions:
            run.sessions[key] = engine.RowSession(eng, direction)
        return run.sessions[key].step(run.rows, run.token, frame, line, run.k)

    def demodulate(self, frame, line, composite):
        return self._demodulate(self._engine(line=line), frame, line, composite)

    def demodulate_components(self, frame, line, composite, strip_chroma=True):
        """(y, u, v) of one row (ref qam.py:43-58 behind pal.py:54-59 / ntsc.py:47-49, comb.py:47-59, 96-113,
        pal.py:180-234); shares the run state with demodulate(), which is this followed by decode_components."""
        eng = self._engine(True, strip_chroma, line)
        if not strip_chroma and getattr(getattr(eng, 'built', None), 'desc', None) is not None and eng.built.desc.first_is_plain:
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
        r, g, b = self._step(run, eng, 'demod', frame, line)
        return r, g, b

    # ---- several rows of a field per call (an addition to the reference's protocol: one launch and one synchronisation
    # for the whole group instead of one per row; the run state is the same, so single-row calls may precede and follow) ----
    def demodulate_rows(self, frame, line, composite_rows):
        """What ``[demodulate(frame, line + 2 * i, composite_rows[i]) for i in range(n)]`` returns (image.py:75-83 feeds the
        rows of a field in exactly this order), as one float64 array [n, 3, W], from ONE launch."""
        rows = numpy.ascontiguousarray(composite_rows, dtype=numpy.float32)
        eng = self._engine(line=line + 2 * (max(len(rows), 1) - 1))
        if len(rows) == 0:
            return numpy.zeros((0, 3, eng.width))
        if rows.ndim != 2 or rows.shape[1] != eng.comp_width:
            raise ValueError('composite_rows must be [n, %d]' % eng.comp_width)
        return self._rows(self._demod_run, eng, 'demod', eng.demod_depth, frame, line, rows)

    def modulate_rows(self, frame, line, r, g, b):
        """What ``[modulate(frame, line + 2 * i, r[i], g[i], b[i]) for i in range(n)]`` returns, as one float64 array [n, W]."""
        if len(r) == 0 and len(g) == 0 and len(b) == 0:
            return numpy.zeros((0, self._engine(line=line).comp_width))
        rows = numpy.ascontiguousarray(numpy.stack([r, g, b], axis=1), dtype=numpy.float32)       # [n, 3, W]
        eng = self._engine(line=line + 2 * (max(len(rows), 1) - 1))
        if rows.ndim != 3 or rows.shape[2] != eng.in_width:
            raise ValueError('r, g, b must be [n, %d] each' % eng.in_width)
        return self._rows(self._mod_run, eng, 'mod', eng.mod_depth, frame, line, rows)

    def _rows(self, run, eng, direction, depth, frame, line, rows):
        n = len(rows)
        if n == 0:
            return numpy.zeros((0,) + ((3, eng.width) if direction == 'demod' else (eng.comp_width,)))
        while direction == 'mod' and getattr(eng, 'encoder', None) is not None:
            eng = eng.encoder
        if direction == 'mod' and getattr(eng, 'noise_level', 0.0):
            # niir.py:45-46: every modulate() call draws its own numpy.random samples, in call order; the run entry point draws for its newest row only
            raise NotImplementedError('modulate_rows is not built for NiirModem(noise_level != 0): call modulate() row by row')
        continuing = frame == run.frame and line == run.line + 2 and run.k >= 0
        k_first = run.k + 1 if continuing else 0
        n_hist = min(k_first, depth)
        hist = run.rows[len(run.rows) - n_hist:] if n_hist else []
        stacked = numpy.concatenate([numpy.stack(hist), rows]) if hist else rows
        fn = eng.demodulate_run if direction == 'demod' else eng.modulate_run
        out = numpy.asarray(fn(stacked, frame, line - 2 * n_hist, k_first - n_hist), dtype=numpy.float64)[n_hist:]
        if not continuing:
            run.token = next(_Run._tokens)
        run.k = k_first + n - 1
        run.frame, run.line = frame, line + 2 * (n - 1)
        run.rows = (hist + [rows[i] for i in range(max(0, n - depth - 1), n)])[-(depth + 1):]
        return out

    def _starts_above(self, frame, line):
        """does the encoder run this call belongs to start above the picture (at a negative line number)?"""
        run = self._mod_run
        continuing = frame == run.frame and line == run.line + 2 and run.k >= 0
        return (line - 2 * (run.k + 1) if continuing else line) < 0

    def modulate(self, frame, line, r, g, b):
        return self._modulate(self._engine(line=max(line, 0), above=self._starts_above(frame, line)), frame, line, r, g, b)

    def modulate_components(self, frame, line, y, u, v):
        """Composite row from (y, u, v) / (luma, dr, db) (ref qam.py:28-32 behind pal.py:48-52 / ntsc.py:43-45,
        comb.py:141-152, secam.py:258-276); shares the run state with modulate()."""
        return self._modulate(self._engine(True, True, max(line, 0), above=self._starts_above(frame, line)), frame, line, y, u, v)

    def _modulate(self, eng, frame, line, r, g, b):
        assert len(r) == len(g) == len(b)
        row = numpy.ascontiguousarray(numpy.stack([r, g, b]), dtype=numpy.float32)
        if row.shape[1] != eng.in_width:
            raise ValueError('r, g, b must be rows of %d samples' % eng.in_width)
        run = self._mod_run
        self._advance(run, frame, line, row, eng.mod_depth)
        return self._step(run, eng, 'mod', frame, line)
