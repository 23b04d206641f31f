# -*- coding: utf-8 -*-
"""Comb decoders whose luma notch has a FilterFunction shift other than 0 (ref comb.py:18-20 over utils.py:9-26).

``notch=q`` designs ``scipy.signal.iirnotch`` at the sub-carrier and wraps it in a FilterFunction, whose delay compensation is
round(group delay at DC): 0 for the usual q >= 2, but 1 for q = 1.0 and other values - negative ones too - below that.  The fused
kernels carry the notch at shift 0 (one more biquad on the luma of a lane, no look-ahead).  For the other values the decoder runs in
two steps behind the same engine interface:

    base engine, component form, strip_chroma as asked, WITHOUT the notch     (y, u, v) of every call - the fused decoder of the stack
    cm_notch_luma_f32                                                         y = notch(y) on every call but the first of a run
                                                                              (comb.py:54-55, 108-110; pal.py:225-228), with the
                                                                              padding / dropping of utils.py:28-36 for either sign of
                                                                              the shift, float64 inside; then decode_components

A fallback for rare parameters, not a throughput path (two more passes over the output).
"""

import ctypes

import numpy

from color_modem_amd import _native, engine


class _BareStack(object):
    """The stack without its notch, in the shape the engines take a modem in."""

    def __init__(self, stack):
        self._bare = dict(stack, comb_notch=None, wrapper_notch=None)

    def _stack(self):
        return self._bare


def shifted_notch(stack, strip_chroma):
    """The notch of this stack that acts on the luma (comb.py:52-55 / 107-110: the wrapper's when there is one - a wrapped comb is called
    with strip_chroma=False - else the comb's) when its FilterFunction shift is not 0, else None."""
    if not strip_chroma or stack.get('kind') not in ('pal_d', 'pal_3d', 'ntsc_comb', 'pal_s', 'ntsc'):
        return None
    f = stack.get('wrapper_notch') if stack.get('demod_wrapper') else stack.get('comb_notch')
    return f if (f is not None and f.shift != 0) else None


class ShiftedNotchEngine(object):
    composite = True          # rowapi: runs go through demodulate_run below

    def __init__(self, modem, components=False, strip_chroma=True, min_lines=0):
        stack = modem._stack()
        self.notch = shifted_notch(stack, strip_chroma)
        assert self.notch is not None
        bare = _BareStack(stack)
        self.base = engine.make_engine(bare, components=True, strip_chroma=strip_chroma, min_lines=min_lines)
        # the wrapper's modulate IS its backend's (comb.py:90-94), and a notch takes no part in it: encode through the leaf engine - a comb
        # wrapper's own `.encoder` - so that the per-row protocol (rowapi._step unwraps one level) opens its session on a plain encoder
        enc = engine.make_engine(bare, components=components, strip_chroma=strip_chroma, min_lines=min_lines)
        while getattr(enc, 'encoder', None) is not None:
            enc = enc.encoder
        self.encoder = enc
        backend = stack['backend']
        m = numpy.eye(3) if components else numpy.asarray(backend.decode_matrix, dtype=numpy.float64)
        self._matrix = numpy.ascontiguousarray(m, dtype=numpy.float64).reshape(-1)
        self._b = numpy.ascontiguousarray(self.notch.b, dtype=numpy.float64)
        self._a = numpy.ascontiguousarray(self.notch.a, dtype=numpy.float64)
        for name in ('width', 'height', 'comp_width', 'in_width', 'demod_depth', 'mod_depth', 'demodulation_delay', 'modulation_delay'):
            setattr(self, name, getattr(self.base, name))
        self.n_lines = min(self.base.n_lines, self.encoder.n_lines)

    def describe(self):
        return '%s (components, no notch) | filter_rows_kernel<float> (notch, shift %d) + matrix_planes_kernel' % (self.base.describe(), self.notch.shift)

    def has_fused_u8(self, direction):
        return direction == 'mod' and self.encoder.has_fused_u8('mod')      # the decoder is a composition of float kernels

    def set_small_batch(self, mode):
        self.base.set_small_batch(mode)
        self.encoder.set_small_batch(mode)

    def _finish(self, yuv, groups, rows, skip, out=None):
        """yuv [groups, 3, rows, W] (a torch tensor on the device) -> decode_components(notch(y), u, v) in `out` (a contiguous tensor of the
        same shape; default: a new one)"""
        import torch
        if out is None:
            out = torch.empty_like(yuv)
        dp = ctypes.POINTER(ctypes.c_double)
        with torch.cuda.device(yuv.device):
            stream = torch.cuda.current_stream(yuv.device).cuda_stream
            _native.check(_native.lib().cm_notch_luma_f32(self._b.ctypes.data_as(dp), len(self._b), self._a.ctypes.data_as(dp), len(self._a),
                                                          int(self.notch.shift), yuv.data_ptr(), out.data_ptr(), int(groups), int(rows),
                                                          int(self.width), int(skip), self._matrix.ctypes.data_as(dp), stream))
        return out

    # ---- frames --------------------------------------------------------------------------------
    CHUNK_BYTES = 1 << 30      # component planes held at a time (the notch and matrix pass is per frame: chunks write straight into the result)

    def demodulate_frames(self, composite, first_frame=0, out=None):
        import torch
        was_numpy = isinstance(composite, numpy.ndarray)
        comp = torch.from_numpy(numpy.ascontiguousarray(composite, dtype=numpy.float32)) if was_numpy else composite
        if not torch.is_tensor(comp) or comp.dtype != torch.float32 or comp.dim() != 3 or tuple(comp.shape[1:]) != (self.height, self.comp_width):
            raise ValueError('composite: expected float32 [n, %d, %d]' % (self.height, self.comp_width))
        if not comp.is_cuda:
            comp = comp.cuda()
        comp = comp.contiguous()
        n = int(comp.shape[0])
        shape = (n, 3, self.height, self.width)
        if out is None:
            out = torch.empty(shape, dtype=torch.float32, device=comp.device)
        else:
            engine._check_out(out, shape, torch.float32, comp.device)
        # image.py:75-83: row r of a field is the result of call r + delay of its run; call 0 is never notched (comb.py:48-49, 97-99)
        skip = 0 if self.demodulation_delay > 0 else min(2, self.height)      # delay 0: rows 0 and 1 of a frame are the two fields' first calls
        step = max(1, self.CHUNK_BYTES // (3 * self.height * self.width * 4))
        for f0 in range(0, n, step):
            yuv = self.base.demodulate_frames(comp[f0:f0 + step], first_frame + f0)
            self._finish(yuv.contiguous(), yuv.shape[0], self.height, skip, out=out[f0:f0 + step])
        return out.cpu().numpy() if was_numpy else out

    def demodulate_frames_u8(self, composite8, first_frame=0, out=None):
        raise NotImplementedError('a notch with a non-zero FilterFunction shift runs on float rows (ImageModem converts on the device around them)')

    def modulate_frames(self, rgb, first_frame=0, out=None):
        return self.encoder.modulate_frames(rgb, first_frame, out=out)

    def modulate_frames_u8(self, rgb8, first_frame=0, out=None):
        return self.encoder.modulate_frames_u8(rgb8, first_frame, out=out)

    # ---- runs (the per-row protocol) --------------------------------------------------------------
    def demodulate_run(self, rows, frame, first_line, k0):
        import torch
        was_numpy = isinstance(rows, numpy.ndarray)
        yuv = self.base.demodulate_run(torch.from_numpy(numpy.ascontiguousarray(rows, dtype=numpy.float32)).cuda() if was_numpy else rows,
                                       frame, first_line, k0)
        n = yuv.shape[0]
        # [n, 3, W] = n groups of one row; the first submitted call is call k0 of its run: unnotched when it is call 0
        res = self._finish(yuv.contiguous().reshape(1, n, 3, self.width).permute(0, 2, 1, 3).contiguous(), 1, n, 1 if k0 == 0 else 0)
        res = res.permute(0, 2, 1, 3).reshape(n, 3, self.width).contiguous()
        return res.cpu().numpy() if was_numpy else res

    def modulate_run(self, rows, frame, first_line, k0):
        return self.encoder.modulate_run(rows, frame, first_line, k0)
