# -*- coding: utf-8 -*-
"""SimpleCombModem / Simple3DCombModem around PalDModem or Pal3DModem (ref comb.py:71-127 over pal.py:62-234).

These stacks do not fit the per-line coefficient tables of the fused decoders: on the second line of every run the wrapper
averages the plain first-line decode (QAM front end) with the first delay-line decode (PAL-D front end), and around
Pal3DModem it reaches back three lines.  They run as a composition of device kernels, call for call what the reference
does (comb.py:96-113):

    backend.demodulate_components(frame, line, composite, strip_chroma=False)   the inner decoder's kernel in component mode
    u, v = avg(last, curr); y = last / curr luma                                cm_comb_combine_run
    backend.modulate_components(frame, line - 2 own_delay, 0, u, v)             the inner modulator's kernel
    y - that; decode_components                                                 cm_comb_finish_run

torch only owns the buffers and gathers the rows of a run.  Not built: a wrapper notch (comb.py:108-110 needs a recursive
filter along the row between the last two steps) and avg= callables other than comb.avg / comb.minavg.
"""

import ctypes

import numpy

from color_modem_amd import _native, engine


class WrappedCombEngine(object):
    composite = True        # rowapi: no device-resident session, the run goes through demodulate_run

    def __init__(self, modem, components=False, strip_chroma=True):
        from color_modem_amd import comb as comb_module
        stack = modem._stack()
        if stack.get('wrapper_notch') is not None:
            raise NotImplementedError('notch= on a comb wrapper around PalDModem / Pal3DModem is not built')
        fn = stack.get('wrapper_avg')
        if fn is not comb_module.avg and fn is not comb_module.minavg:
            raise NotImplementedError('avg=%r: the device path implements comb.avg and comb.minavg' % (fn,))
        if not strip_chroma:
            raise NotImplementedError('demodulate_components(strip_chroma=False) on a wrapper around PalDModem / Pal3DModem is not built')
        self.minavg = fn is comb_module.minavg
        self.own_delay = 1 if stack['demod_wrapper'] == 'simple_3d' else 0
        self.inner_modem = stack['comb']                    # PalDModem / Pal3DModem
        self.backend = stack['backend']                     # PalSModem
        lc = self.backend.line_config
        self.width, self.height = int(lc.size[0]), int(lc.size[1])
        self.comp_width = self.in_width = self.width
        d_in = int(getattr(self.inner_modem, 'demodulation_delay', 0))
        self.demodulation_delay = d_in + self.own_delay
        self.modulation_delay = 0
        need = self.height + 2 * self.demodulation_delay + 8
        self.inner = engine.Engine(self.inner_modem, components=True, strip_chroma=False, min_lines=need)
        self.first = None
        if self.inner.built.desc.first_is_plain:
            # the first line of a run is the backend's own unstripped decode (comb.py:48-49): its plan
            self.first = engine.Engine(self.backend, components=True, strip_chroma=False, min_lines=need)
        self.mod = engine.Engine(self.backend, components=True, min_lines=need)
        self.encoder = engine.Engine(self.backend, components=components, min_lines=need)     # wrapper.modulate = backend.modulate
        self.demod_depth = self.inner.demod_depth + 1
        self.mod_depth = 0
        self.n_lines = 1 << 30
        eye = numpy.eye(3)
        self._matrix = numpy.ascontiguousarray(eye if components else self.backend.decode_matrix, dtype=numpy.float64).reshape(-1)

    def describe(self):
        return 'composition: %s (components) | comb_combine_kernel | qam_mod_kernel | comb_finish_kernel' % self.inner.describe()

    # ---- one run ------------------------------------------------------------------------------
    def _run(self, rows, frame, first_line, k0):
        """rows [n, W] cuda tensor: calls k0 .. k0 + n - 1 of one run at lines first_line, first_line + 2, ...
        -> [n, 3, W] cuda tensor (rows whose history lies before the buffer are unspecified)."""
        import torch
        n, W = rows.shape
        L = _native.lib()
        yuv = self.inner.demodulate_run(rows, frame, first_line, k0)
        if k0 == 0 and self.first is not None:
            yuv[0:1] = self.first.demodulate_run(rows[0:1], frame, first_line, 0)
        uv = torch.empty_like(yuv)
        ysrc = torch.empty((n, W), dtype=torch.float32, device=rows.device)
        remod = torch.zeros((n, W), dtype=torch.float32, device=rows.device)
        out = torch.empty_like(yuv)
        with torch.cuda.device(rows.device):
            stream = torch.cuda.current_stream(rows.device).cuda_stream
            _native.check(L.cm_comb_combine_run(yuv.data_ptr(), uv.data_ptr(), ysrc.data_ptr(), n, W, int(k0), self.own_delay,
                                                1 if self.minavg else 0, stream))
            i0 = 1 if k0 == 0 else 0                      # call 0 of a run is not stripped (comb.py:97-99)
            if n > i0:
                remod[i0:] = self.mod.modulate_run(uv[i0:], frame, first_line + 2 * i0 - 2 * self.own_delay, 0)
            m = (ctypes.c_double * 9)(*self._matrix)
            _native.check(L.cm_comb_finish_run(uv.data_ptr(), ysrc.data_ptr(), remod.data_ptr(), m, out.data_ptr(), n, W, int(k0), stream))
        return out

    def demodulate_run(self, rows, frame, first_line, k0):
        import torch
        was_numpy = isinstance(rows, numpy.ndarray)
        t = torch.from_numpy(numpy.ascontiguousarray(rows, dtype=numpy.float32)).cuda() if was_numpy else rows.contiguous()
        if self.width % 4:
            raise NotImplementedError('wrappers around PalDModem / Pal3DModem need a width that is a multiple of 4')
        out = self._run(t, int(frame), int(first_line), int(k0))
        return out.cpu().numpy() if was_numpy else out

    # ---- frames: the row schedule of image.py:75-83, one run per field -----------------------
    def demodulate_frames(self, composite, first_frame=0, out=None):
        import torch
        was_numpy = isinstance(composite, numpy.ndarray)
        comp = torch.from_numpy(numpy.ascontiguousarray(composite, dtype=numpy.float32)).cuda() if was_numpy else composite
        if comp.dtype != torch.float32 or comp.dim() != 3 or tuple(comp.shape[1:]) != (self.height, self.width):
            raise ValueError('composite: expected float32 [n, %d, %d]' % (self.height, self.width))
        if self.width % 4:
            raise NotImplementedError('wrappers around PalDModem / Pal3DModem need a width that is a multiple of 4')
        n, H, D = comp.shape[0], self.height, self.demodulation_delay
        shape = (n, 3, H, self.width)
        if out is None:
            out = torch.empty(shape, dtype=torch.float32, device=comp.device)
        else:
            engine._check_out(out, shape, torch.float32, comp.device)
        for fi in range(n):
            for field in range(2):
                rows_out = list(range(field, H, 2))
                if not rows_out:
                    continue
                lines = [field + 2 * k for k in range(len(rows_out) + D)]
                src = []
                for ln in lines:
                    while ln >= H:                        # image.py:80-81
                        ln -= 2
                    src.append(ln)
                idx = torch.tensor(src, dtype=torch.long, device=comp.device)
                res = self._run(comp[fi].index_select(0, idx).contiguous(), int(first_frame) + fi, field, 0)
                out[fi, :, field::2] = res[D:].permute(1, 0, 2)
        return out.cpu().numpy() if was_numpy else out

    def demodulate_frames_u8(self, *args, **kwargs):
        raise NotImplementedError('no fused byte boundary for this stack: the PIL entry points convert on the host')

    # ---- the encoder side is the backend's (comb.py:90-94) ------------------------------------
    def modulate_frames(self, rgb, first_frame=0, out=None):
        return self.encoder.modulate_frames(rgb, first_frame, out=out)

    def modulate_frames_u8(self, rgb8, first_frame=0, out=None):
        return self.encoder.modulate_frames_u8(rgb8, first_frame, out=out)

    def modulate_run(self, rows, frame, first_line, k0):
        return self.encoder.modulate_run(rows, frame, first_line, k0)
