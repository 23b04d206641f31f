# -*- coding: utf-8 -*-
"""SimpleCombModem / Simple3DCombModem around PalDModem or Pal3DModem (ref comb.py:71-127 over pal.py:62-234).

These stacks do not fit the per-line coefficient tables of the fused decoders: on the second line of every run the wrapper
averages the plain first-line decode (QAM front end) with the first delay-line decode (PAL-D front end), and around
Pal3DModem it reaches back three lines.  They run as two streaming kernels per batch behind ONE native call
(`cm_comb_wrap_demodulate_frames` / `_frames_u8` / `_run`, csrc/cm_wrap_kernels.h), call for call what the reference does
(comb.py:96-113):

    backend.demodulate_components(frame, line, composite, strip_chroma=False)   the inner decoder's kernel in component mode,
                                                                                every call of every run of the batch
    u, v = avg(last, curr); y = last / curr luma                                |
    y -= backend.modulate_components(frame, line - 2 own_delay, 0, u, v)        |  comb_wrap_back_kernel: one lane per call,
    y = notch(y)                                                                |  the previous call = the neighbouring lane
    decode_components                                                           |

This module only builds the three plans (inner decoder, plain first-line decoder, backend modulator) and the wrapper's
descriptor.

avg= callables other than comb.avg / comb.minavg (comb.py:72, 81-84) run through the same composition, cut in two
(`cm_comb_wrap_components_*` / `cm_comb_wrap_finish_*`): the caller's function is applied between the kernels to the (u, v) planes of
consecutive calls - as float32 torch tensors ON THE DEVICE, a whole batch at a time ([frames, calls - 1, W] per run), where the reference
hands it one float64 numpy row per call; elementwise functions of two arrays (the only kind that makes sense there) behave the same.
A function that cannot take device tensors (numpy ufuncs raise TypeError on them) is called again with float64 numpy arrays on the host
(color_modem_amd/avgfn.py; any other exception propagates).
That also serves the wrappers around the plain decoders (NtscModem, PalSModem, NtscCombModem), whose comb.avg / comb.minavg forms are
fused into lane tables instead.

Round 5: around Pal3DModem every chroma estimate comes from ONE front end, so long batches run as a two-level comb in a single launch
(`_TwoLevelStack`: Pal3DModem's lane tables + the average of consecutive calls inside the kernel, three halo lanes) - no scratch, no
composition; short batches, avg= callables and the per-row protocol stay on the composition.

Round 4: around PalDModem a fourth plan removes the component scratch from long batches.  From the third call of a run on, both
chroma estimates the wrapper averages are PAL-D decodes - combinations of the PAL-D front end's base pairs of three consecutive
lines - so the whole wrapper is one more line of history in the fused decoder's lane tables (plan.QamTables: fused_main; kernel
instance: PAL-D front end, depth 2).  Only the first two calls of a run (the top four rows of a frame) mix in the plain decode and
still go through the composition (`cm_comb_wrap_demodulate_frames_fused`: 16 instead of 40 bytes per pixel through HBM).
"""

import ctypes

import numpy

from color_modem_amd import _native, avgfn, engine, plan


class CombWrapDesc(ctypes.Structure):
    """cm_comb_wrap_desc (include/color_modem_hip.h)"""
    _fields_ = [('own_delay', ctypes.c_int32), ('minavg', ctypes.c_int32), ('strip_chroma', ctypes.c_int32),
                ('reserved', ctypes.c_int32), ('notch', plan.IirDesc), ('matrix', ctypes.c_double * 9)]


class _FusedStack(object):
    """The wrapper's stack marked for the fused plan (plan.QamTables: fused_main), in the shape engine.Engine takes a modem in."""

    def __init__(self, stack):
        self._marked = dict(stack, fused_main=True)

    def _stack(self):
        return self._marked


class _TwoLevelStack(object):
    """The wrapper's stack around Pal3DModem marked for the two-level comb (plan.QamTables: two_level)."""

    def __init__(self, stack):
        self._marked = dict(stack, two_level=True)

    def _stack(self):
        return self._marked


class WrappedCombEngine(object):
    composite = False       # rowapi: the run lives in an engine.RowSession (pinned zero-copy rows) through row_call below

    def __init__(self, modem, components=False, strip_chroma=True, min_lines=0):
        from color_modem_amd import comb as comb_module
        stack = modem._stack()
        fn = stack.get('wrapper_avg')
        self.custom_avg = fn if (fn is not comb_module.avg and fn is not comb_module.minavg) else None
        if self.custom_avg is not None and not callable(self.custom_avg):
            raise TypeError('avg=%r is not callable' % (fn,))
        if self.custom_avg is not None:
            self.composite = True      # rowapi: runs go through demodulate_run below (the callable sits between two native calls)
        notch = stack.get('wrapper_notch') if strip_chroma else None      # comb.py:105-110: the notch follows the strip
        if notch is not None and notch.shift != 0:      # (engine.make_engine sends such stacks to notched.ShiftedNotchEngine, which builds this engine without the notch)
            raise NotImplementedError('the back end carries the notch at FilterFunction shift 0; shift %d goes through color_modem_amd/notched.py' % notch.shift)
        self.own_delay = 1 if stack['demod_wrapper'] == 'simple_3d' else 0
        self.inner_modem = stack.get('comb') or stack['backend']      # PalDModem / Pal3DModem (avg= callables: any QAM-family decoder)
        self.backend = stack['backend']                     # PalSModem
        lc = self.backend.line_config
        self.width, self.height = int(lc.size[0]), int(lc.size[1])
        self.comp_width = self.in_width = self.width
        d_in = int(getattr(self.inner_modem, 'demodulation_delay', 0))
        self.demodulation_delay = d_in + self.own_delay
        self.modulation_delay = 0
        need = max(self.height + 2 * self.demodulation_delay + 8, int(min_lines))
        self.inner = engine.Engine(self.inner_modem, components=True, strip_chroma=False, min_lines=need)
        self.first = None
        if self.inner.built.desc.first_is_plain:
            # the first line of a run is the backend's own unstripped decode (comb.py:48-49): its plan
            self.first = engine.Engine(self.backend, components=True, strip_chroma=False, min_lines=need)
        self.mod = engine.Engine(self.backend, components=True, min_lines=need)       # the wrapper re-modulates through the backend
        self.encoder = engine.Engine(self.backend, components=components, min_lines=need)     # wrapper.modulate = backend.modulate
        self.fused = None
        if stack['kind'] == 'pal_d' and self.custom_avg is None:
            try:
                self.fused = engine.Engine(_FusedStack(stack), components=components, strip_chroma=strip_chroma, min_lines=need)
            except NotImplementedError:      # no PAL-D depth-2 instance for this filter-set shape: the composition serves every batch
                self.fused = None
        elif stack['kind'] == 'pal_3d' and self.custom_avg is None and getattr(self.inner_modem, 'demodulation_delay', 0) == 1:
            # round 5: around Pal3DModem every estimate comes from the QAM front end, so ONE plan decodes every call of every run - Pal3DModem's
            # lane tables, the wrapper's average of consecutive calls inside the kernel (cm_lane_table::wrap_mode, PassCfg::WRAP)
            try:
                self.fused = engine.Engine(_TwoLevelStack(stack), components=components, strip_chroma=strip_chroma, min_lines=need)
            except NotImplementedError:      # the run-time filter shape (other sampling rates): the composition
                self.fused = None
        self.demod_depth = self.inner.demod_depth + 1
        self.mod_depth = 0
        self.n_lines = min(e.n_lines for e in (self.inner, self.first, self.mod, self.encoder) if e is not None)
        w = CombWrapDesc()
        w.own_delay = self.own_delay
        w.minavg = 2 if self.custom_avg is not None else (1 if fn is comb_module.minavg else 0)
        w.strip_chroma = 1 if strip_chroma else 0
        w.notch = plan.iir_desc(notch)
        eye = numpy.eye(3)
        w.matrix[:] = list(numpy.asarray(eye if components else self.backend.decode_matrix, dtype=numpy.float64).reshape(-1))
        self.desc = w

    def describe(self):
        text = 'composition: %s (components, every call of the batch) | comb_wrap_back_kernel (small batches: wrap_back_scan_kernel)' % self.inner.describe()
        if self.fused is not None and self.fused.built.tables.two_level:
            text = 'long batches: %s; otherwise %s' % (self.fused.describe(), text)
        elif self.fused is not None:
            text = 'long batches: %s + the composition on the top four rows; otherwise %s' % (self.fused.describe(), text)
        return text

    def has_fused_u8(self, direction):
        if direction == 'mod':
            return self.encoder.has_fused_u8('mod')
        return self.width % 4 == 0 and self.custom_avg is None       # (avg= callables sit between two float kernels)

    def set_small_batch(self, mode):
        """Engine.set_small_batch for the plans a wrapped decode runs through (inner decoder, plain first line, back end) and the encoder's."""
        for e in (self.inner, self.first, self.mod, self.encoder):
            if e is not None:
                e.set_small_batch(mode)       # (a pinned mode on the inner plan also keeps long batches on the composition)

    def _plans(self, device):
        return (self.inner._plans.get(device), self.first._plans.get(device) if self.first is not None else None,
                self.mod._plans.get(device))

    def row_call(self, direction):
        """engine.RowSession's entry point for one run: (plans, in, out, n_calls, frame, first_line, k0, stream) -> status."""
        assert direction == 'demod'       # the encoder side is the backend's own engine (rowapi._step)
        fn, desc = _native.lib().cm_comb_wrap_demodulate_run, self.desc
        return lambda plans, src, dst, n, frame, line, k0, stream: fn(plans[0], plans[1], plans[2], ctypes.byref(desc), src, dst, n, frame, line, k0, stream)

    def _call(self, fn, x, out, *args, **kw):
        import torch
        inner, first, mod = self._plans(x.device)
        lead = (self.fused._plans.get(x.device) if self.fused is not None else None,) if kw.get('fused') else ()
        with torch.cuda.device(x.device):
            stream = torch.cuda.current_stream(x.device).cuda_stream
            _native.check(fn(*(lead + (inner, first, mod, ctypes.byref(self.desc), x.data_ptr(), out.data_ptr()) + tuple(args) + (stream,))))

    # ---- one run (rowapi) ----------------------------------------------------------------------
    def demodulate_run(self, rows, frame, first_line, k0):
        """rows [n, W]: calls k0 .. k0 + n - 1 of one run at lines first_line, first_line + 2, ... -> [n, 3, W] (rows whose
        history lies before the buffer are unspecified)."""
        import torch
        was_numpy = isinstance(rows, numpy.ndarray)
        t = torch.from_numpy(numpy.ascontiguousarray(rows, dtype=numpy.float32)).cuda() if was_numpy else rows.contiguous()
        if t.dim() != 2 or t.shape[1] != self.width or t.dtype != torch.float32:
            raise ValueError('rows: expected float32 [n, %d]' % self.width)
        if not t.is_cuda:
            t = t.cuda()
        out = torch.empty((t.shape[0], 3, self.width), dtype=torch.float32, device=t.device)
        args = (int(t.shape[0]), int(frame), int(first_line), int(k0))
        if self.custom_avg is None:
            self._call(_native.lib().cm_comb_wrap_demodulate_run, t, out, *args)
        else:
            self._need_quads()
            buf = torch.empty((t.shape[0], 3, self.width), dtype=torch.float32, device=t.device)
            self._call(_native.lib().cm_comb_wrap_components_run, t, buf, *args)
            self._average(buf[None], [(0, int(t.shape[0]))])
            self._call(_native.lib().cm_comb_wrap_finish_run, buf, out, *args)
        return out.cpu().numpy() if was_numpy else out

    # ---- avg= callables: between the two halves of the composition ------------------------------
    def _need_quads(self):
        if self.width % 4:
            raise NotImplementedError('avg= callables need a width that is a multiple of 4 (the component buffer form of the composition)')

    def _average(self, buf, runs):
        """buf [frames, calls, 3, W] as the inner decoder left it; runs: [lo, hi) call ranges of one frame.  (u, v) of every call but a
        run's first become avg(previous call's, this call's) - comb.py:103-104 - both taken from the buffer as it was."""
        fn = self.custom_avg
        for lo, hi in runs:
            if hi - lo < 2:
                continue
            done = []
            for plane in (1, 2):
                # (avgfn: device tensors first; a function written against numpy - TypeError on them - gets float64 numpy arrays, as in the reference)
                done.append(avgfn.apply(fn, buf[:, lo:hi - 1, plane], buf[:, lo + 1:hi, plane]).clone())
            buf[:, lo + 1:hi, 1] = done[0]
            buf[:, lo + 1:hi, 2] = done[1]

    def _frames_custom(self, comp, out, first_frame):
        import torch
        self._need_quads()
        L = _native.lib()
        inner = self.inner._plans.get(comp.device)
        calls = int(L.cm_comb_wrap_calls_per_frame(inner, ctypes.byref(self.desc)))
        if calls <= 0:
            _native.check(calls)
        run0 = (self.height + 1) // 2 + self.demodulation_delay
        budget = max(1, (1 << 30) // (calls * 3 * self.width * 4))        # frames per pass: the component buffer stays within 1 GiB
        for f0 in range(0, comp.shape[0], budget):
            part = comp[f0:f0 + budget]
            buf = torch.empty((part.shape[0], calls, 3, self.width), dtype=torch.float32, device=comp.device)
            self._call(L.cm_comb_wrap_components_frames, part, buf, int(part.shape[0]), int(first_frame) + f0)
            self._average(buf, [(0, run0), (run0, calls)])
            self._call(L.cm_comb_wrap_finish_frames, buf, out[f0:f0 + budget], int(part.shape[0]), int(first_frame) + f0)

    # ---- frames: the row schedule of image.py:75-83 --------------------------------------------
    def _frames(self, fn, composite, dtype, out, out_shape_of, first_frame):
        import torch
        was_numpy = isinstance(composite, numpy.ndarray)
        np_dtype = numpy.float32 if dtype == torch.float32 else numpy.uint8
        comp = torch.from_numpy(numpy.ascontiguousarray(composite, dtype=np_dtype)).cuda() if was_numpy else composite
        if not torch.is_tensor(comp) or comp.dtype != dtype or comp.dim() != 3 or tuple(comp.shape[1:]) != (self.height, self.width):
            raise ValueError('composite: expected %s [n, %d, %d]' % ('float32' if dtype == torch.float32 else 'uint8', self.height, self.width))
        if not comp.is_cuda:
            comp = comp.cuda()
        comp = comp.contiguous()
        shape = out_shape_of(comp.shape[0])
        if out is None:
            out = torch.empty(shape, dtype=dtype, device=comp.device)
        else:
            engine._check_out(out, shape, dtype, comp.device)
        if self.custom_avg is not None:
            if dtype != torch.float32:
                raise NotImplementedError('avg= callables run on float rows (ImageModem converts on the device around them)')
            self._frames_custom(comp, out, first_frame)
        else:
            self._call(fn, comp, out, int(comp.shape[0]), int(first_frame), fused=True)
        return out.cpu().numpy() if was_numpy else out

    def demodulate_frames(self, composite, first_frame=0, out=None):
        """composite [F, H, W] float32 (numpy or cuda tensor) -> rgb [F, 3, H, W] of the same kind."""
        import torch
        return self._frames(_native.lib().cm_comb_wrap_demodulate_frames_fused, composite, torch.float32, out,
                            lambda n: (n, 3, self.height, self.width), first_frame)

    def demodulate_frames_u8(self, composite8, first_frame=0, out=None):
        """'L' bytes [F, H, W] -> interleaved 'RGB' bytes [F, H, W, 3] (image.py:58-84 fused into the kernels)."""
        import torch
        if self.width % 4:
            raise NotImplementedError('the fused uint8 boundary needs a width that is a multiple of 4')
        return self._frames(_native.lib().cm_comb_wrap_demodulate_frames_fused_u8, composite8, torch.uint8, out,
                            lambda n: (n, self.height, self.width, 3), first_frame)

    # ---- the encoder side is the backend's (comb.py:90-94) ------------------------------------
    def modulate_frames(self, rgb, first_frame=0, out=None):
        return self.encoder.modulate_frames(rgb, first_frame, out=out)

    def modulate_frames_u8(self, rgb8, first_frame=0, out=None):
        return self.encoder.modulate_frames_u8(rgb8, first_frame, out=out)

    def modulate_run(self, rows, frame, first_line, k0):
        return self.encoder.modulate_run(rows, frame, first_line, k0)
