# -*- coding: utf-8 -*-
"""Pal3DModem(avg=f) with a function of the caller's own (ref pal.py:144-148, 176-179, 209-211).

Pal3DModem forms two estimates of (u, v) per call - one from the sum, one from the difference of consecutive line differences
(pal.py:203-207) - and combines them with ``avg``.  ``comb.avg`` folds into the lane tables, ``comb.minavg`` is kernel code; any
other function cannot be either.  The two estimates are exactly what ``Pal3DModem(use_cos=False)`` and ``Pal3DModem(use_sin=False)``
return for (u, v), so the decode is cut where the reference calls the function:

    engine A (sin estimate only), engine B (cos estimate only)     the fused Pal3D kernels in component mode, strip_chroma=False:
                                                                   (y = the previous composite row, u, v) of every output row
    u = f(uA, uB);  v = s f(s vA, s vB)                            on the device, a whole batch at a time (float32 torch tensors; a
                                                                   function that cannot take them gets float64 numpy arrays on the
                                                                   host).  s = the V-switch sign of line - 2, which the reference
                                                                   applies AFTER f (pal.py:219-220) and the engines before they return
    y -= backend.modulate_components(frame, line - 2, 0, u, v)     the backend's modulator kernel on (0, u, v) (pal.py:225-226)
    y = notch(y); decode_components                                cm_notch_luma_f32 (any FilterFunction shift) / the matrix

The calls k < 2 of a run (pal.py:191-201: the plain decode of the first line, returned once unstripped and once more through the strip)
never reach f: they are engine A's rows as they are.  A fallback for a rare parameter, not a throughput path (three decoder passes'
worth of traffic); bytes at the boundary go through the host conversions of ImageModem.
"""

import copy
import ctypes

import numpy

from color_modem_amd import _native, engine


def custom_avg(stack):
    """the callable of a bare Pal3DModem that combines both estimates with something else than comb.avg / comb.minavg, else None"""
    from color_modem_amd import comb
    if stack.get('kind') != 'pal_3d' or stack.get('demod_wrapper'):
        return None
    c3 = stack['comb']
    if not (c3._use_sin and c3._use_cos):
        return None
    fn = c3._avg
    return fn if (fn is not None and fn is not comb.avg and fn is not comb.minavg) else None


class _OneEstimate(object):
    """The stack of the same Pal3DModem with one of its two estimates switched off (and no notch: it follows the strip, pal.py:227-228)."""

    def __init__(self, stack, use_sin, use_cos):
        from color_modem_amd import comb
        c3 = copy.copy(stack['comb'])
        c3._use_sin, c3._use_cos, c3._avg, c3.notch = use_sin, use_cos, comb.avg, None
        self._one = dict(stack, comb=c3, comb_notch=None)

    def _stack(self):
        return self._one


class Pal3DCallableEngine(object):
    composite = True          # rowapi: runs go through demodulate_run below (the callable sits between native calls)
    CHUNK_BYTES = 1 << 29

    def __init__(self, modem, components=False, strip_chroma=True, min_lines=0):
        stack = modem._stack()
        self.fn = custom_avg(stack)
        assert self.fn is not None
        if stack.get('mod_wrapper'):
            raise NotImplementedError('ColorAveragingModem around Pal3DModem(avg=f): encode through ColorAveragingModem(PalSModem) instead')
        self.backend = stack['backend']
        self.lc = self.backend.line_config
        self.strip = bool(strip_chroma)
        self.notch = stack.get('comb_notch') if self.strip else None
        self.a = engine.Engine(_OneEstimate(stack, True, False), components=True, strip_chroma=False, min_lines=min_lines)
        self.b = engine.Engine(_OneEstimate(stack, False, True), components=True, strip_chroma=False, min_lines=min_lines)
        self.mod = engine.Engine(self.backend, components=True, min_lines=max(min_lines, self.a.n_lines))
        self.encoder = engine.Engine(self.backend, components=components, min_lines=min_lines)
        for name in ('width', 'height', 'comp_width', 'in_width', 'demod_depth', 'demodulation_delay'):
            setattr(self, name, getattr(self.a, name))
        self.mod_depth, self.modulation_delay = 0, 0
        self.n_lines = min(self.a.n_lines, self.b.n_lines, self.mod.n_lines, self.encoder.n_lines)
        m = numpy.eye(3) if components else numpy.asarray(self.backend.decode_matrix, dtype=numpy.float64)
        self._matrix = numpy.ascontiguousarray(m, dtype=numpy.float64).reshape(-1)
        f = self.notch
        self._b = numpy.ascontiguousarray(f.b if f is not None else [1.0], dtype=numpy.float64)
        self._a = numpy.ascontiguousarray(f.a if f is not None else [1.0], dtype=numpy.float64)
        self._shift = int(f.shift) if f is not None else 0

    def _row_parity(self, h):
        """analog_line(row) % 2 for the rows of a picture (line.py:57-62), cached"""
        cached = getattr(self, '_parity', None)
        if cached is None or len(cached) != h:
            cached = self._parity = numpy.array([self.lc.analog_line(int(r)) % 2 for r in range(h)], dtype=numpy.int64)
        return cached

    def describe(self):
        return ('Pal3DModem(avg=f): 2 x %s (one estimate each, components) | f on the device | qam_mod_kernel (strip) | filter_rows_kernel + matrix'
                % self.a.describe().split(';')[0])

    def has_fused_u8(self, direction):
        return direction == 'mod' and self.encoder.has_fused_u8('mod')      # the decoder is a composition of float kernels

    def set_small_batch(self, mode):
        for e in (self.a, self.b, self.mod, self.encoder):
            e.set_small_batch(mode)

    # ---- the function between the kernels -------------------------------------------------------------------------------------
    def _apply(self, x, y):
        from color_modem_amd import avgfn
        return avgfn.apply(self.fn, x, y)

    def _combine(self, ya, yb, sign, plain):
        """(f sees whole batches - the rows of the calls k < 2 of a run included, whose results are discarded below; the reference never calls
        it there, pal.py:191-201: an f that is not defined on every finite pair of arrays should mask by itself.)
        ya, yb [..., 3, rows, W] from the two engines; sign [..., rows, 1] = the V-switch sign of the stripped line; plain [..., rows, 1]
        (bool): rows that never reach f (calls k < 2) -> (y, u, v) with u, v combined"""
        import torch
        u = self._apply(ya[..., 1, :, :], yb[..., 1, :, :])
        v = sign * self._apply(sign * ya[..., 2, :, :], sign * yb[..., 2, :, :])
        u = torch.where(plain, ya[..., 1, :, :], u)
        v = torch.where(plain, ya[..., 2, :, :], v)
        return ya[..., 0, :, :], u, v

    def _finish(self, y, u, v, groups, rows, skip, out):
        """(y, u, v) [groups, rows, W] -> decode_components(notch(y), u, v) in `out` [groups, 3, rows, W] (rows before `skip`: no notch)"""
        import torch
        yuv = torch.stack([y, u, v], dim=1).contiguous()
        dp = ctypes.POINTER(ctypes.c_double)
        with torch.cuda.device(yuv.device):
            stream = torch.cuda.current_stream(yuv.device).cuda_stream
            _native.check(_native.lib().cm_notch_luma_f32(self._b.ctypes.data_as(dp), len(self._b), self._a.ctypes.data_as(dp), len(self._a),
                                                          self._shift, yuv.data_ptr(), out.data_ptr(), int(groups), int(rows), int(self.width),
                                                          int(skip), self._matrix.ctypes.data_as(dp), stream))

    # ---- frames -----------------------------------------------------------------------------------------------------------------
    def demodulate_frames(self, composite, first_frame=0, out=None):
        import torch
        was_numpy = isinstance(composite, numpy.ndarray)
        comp = torch.from_numpy(numpy.ascontiguousarray(composite, dtype=numpy.float32)) if was_numpy else composite
        if not torch.is_tensor(comp) or comp.dtype != torch.float32 or comp.dim() != 3 or tuple(comp.shape[1:]) != (self.height, self.comp_width):
            raise ValueError('composite: expected float32 [n, %d, %d]' % (self.height, self.comp_width))
        comp = (comp if comp.is_cuda else comp.cuda()).contiguous()
        n, h, w = int(comp.shape[0]), self.height, self.width
        shape = (n, 3, h, w)
        if out is None:
            out = torch.empty(shape, dtype=torch.float32, device=comp.device)
        else:
            engine._check_out(out, shape, torch.float32, comp.device)
        # output row r of a frame comes from the call at line r + 2 (demodulation_delay 1): k = r // 2 + 1, stripped at line r
        rows = numpy.arange(h)
        plain = torch.from_numpy((rows // 2 + 1) < 2).to(comp.device).reshape(1, h, 1)
        step = max(1, self.CHUNK_BYTES // (3 * h * w * 4))
        for f0 in range(0, n, step):
            part = comp[f0:f0 + step]
            m = int(part.shape[0])
            frames = numpy.arange(first_frame + f0, first_frame + f0 + m)
            # line.py:64-65: a line is an alternate one when its analog line number has the frame's parity - the rows' parities once per
            # engine, the frames' per chunk (round 6: this was a Python loop over every (frame, row) of the chunk)
            sgn = numpy.where(self._row_parity(h)[None, :] == (frames % 2)[:, None], -1.0, 1.0).astype(numpy.float32)
            sign = torch.from_numpy(sgn).to(comp.device).reshape(m, h, 1)
            ya = self.a.demodulate_frames(part, first_frame + f0)
            yb = self.b.demodulate_frames(part, first_frame + f0)
            y, u, v = self._combine(ya, yb, sign, plain)
            if self.strip:      # pal.py:225-226: every output row is stripped at its own line (= the call's line - 2)
                zuv = torch.stack([torch.zeros_like(u), u, v], dim=1).contiguous()
                y = y - self.mod.modulate_frames(zuv, first_frame + f0)
            self._finish(y, u, v, m, h, 0, out[f0:f0 + step])
        return out.cpu().numpy() if was_numpy else out

    def demodulate_frames_u8(self, composite8, first_frame=0, out=None):
        raise NotImplementedError('Pal3DModem(avg=f) runs on float rows (ImageModem converts on the device around them)')

    def modulate_frames(self, rgb, first_frame=0, out=None):
        return self.encoder.modulate_frames(rgb, first_frame, out=out)

    def modulate_frames_u8(self, rgb8, first_frame=0, out=None):
        return self.encoder.modulate_frames_u8(rgb8, first_frame, out=out)

    # ---- runs (the per-row protocol) --------------------------------------------------------------------------------------------
    def demodulate_run(self, rows, frame, first_line, k0):
        """rows [n, W]: calls k0 .. k0 + n - 1 of one run at lines first_line, first_line + 2, ... -> what each call returns [n, 3, W]"""
        import torch
        was_numpy = isinstance(rows, numpy.ndarray)
        t = torch.from_numpy(numpy.ascontiguousarray(rows, dtype=numpy.float32)).cuda() if was_numpy else rows.contiguous()
        n, w = int(t.shape[0]), self.width
        ya = self.a.demodulate_run(t, frame, first_line, k0).permute(1, 0, 2)        # [3, n, W]
        yb = self.b.demodulate_run(t, frame, first_line, k0).permute(1, 0, 2)
        ks = k0 + numpy.arange(n)
        lines = first_line + 2 * numpy.arange(n)
        plain = torch.from_numpy(ks < 2).to(t.device).reshape(n, 1)
        sgn = numpy.array([-1.0 if self.lc.is_alternate_line(int(frame), int(l) - 2) else 1.0 for l in lines], dtype=numpy.float32)
        sign = torch.from_numpy(sgn).to(t.device).reshape(n, 1)
        y, u, v = self._combine(ya, yb, sign, plain)
        stripped = [i for i in range(n) if ks[i] >= 1]       # pal.py:191-195: call 0 returns before the strip
        if self.strip and stripped:
            i0 = stripped[0]
            zuv = torch.stack([torch.zeros_like(u[i0:]), u[i0:], v[i0:]], dim=1).contiguous()         # [m, 3, W] at lines first_line + 2 i - 2
            y = y.clone()
            y[i0:] = y[i0:] - self.mod.modulate_run(zuv, frame, int(lines[i0]) - 2, 0)
        out = torch.empty((1, 3, n, w), dtype=torch.float32, device=t.device)
        self._finish(y[None], u[None], v[None], 1, n, 1 if k0 == 0 else 0, out)
        res = out[0].permute(1, 0, 2).contiguous()
        return res.cpu().numpy() if was_numpy else res

    def modulate_run(self, rows, frame, first_line, k0):
        return self.encoder.modulate_run(rows, frame, first_line, k0)
