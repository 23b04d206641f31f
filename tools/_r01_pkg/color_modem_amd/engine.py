# -*- coding: utf-8 -*-
"""Device engine: owns one plan (cm_plan) per modem stack and moves rows / frames through it.

torch (ROCm build) is used only as the owner of device memory and of the HIP stream; all
arithmetic happens in libcolor_modem_hip.so.
"""

import ctypes

import numpy

from color_modem_amd import _native, plan


def _torch():
    import torch
    if not torch.cuda.is_available():
        raise _native.NativeError('no HIP device visible to torch: color_modem_amd runs on the GPU only')
    return torch


def make_engine(modem, components=False, strip_chroma=True):
    """The engine of a modem stack: a cm_plan for the QAM / SECAM families, the plan-less MAC entry points for MacModem."""
    if modem._stack()['kind'] == 'mac':
        return MacEngine(modem, components)
    return Engine(modem, components, strip_chroma)


class Engine(object):
    def __init__(self, modem, components=False, strip_chroma=True):
        self.built = plan.build_plan(modem, components, strip_chroma)
        d = self.built.desc
        self.width, self.height = d.width, d.height
        self.comp_width = d.width
        self.in_width = d.width
        self.demod_depth = d.depth
        self.mod_depth = 1 if d.modulation_delay else 0
        self.demodulation_delay = d.demodulation_delay
        self.modulation_delay = d.modulation_delay
        self._plan = ctypes.c_void_p()
        _torch()
        _native.check(_native.lib().cm_plan_create(ctypes.byref(d), ctypes.byref(self._plan)))

    def __del__(self):
        p = getattr(self, '_plan', None)
        if p and _native is not None and getattr(_native, '_lib', None) is not None:
            _native._lib.cm_plan_destroy(p)
            self._plan = None

    def describe(self):
        buf = ctypes.create_string_buffer(512)
        _native.lib().cm_plan_describe(self._plan, buf, 512)
        return buf.value.decode()

    # ---- frames -------------------------------------------------------------------------------
    def _as_device(self, x, shape_tail):
        torch = _torch()
        was_numpy = isinstance(x, numpy.ndarray)
        t = torch.from_numpy(numpy.ascontiguousarray(x, dtype=numpy.float32)) if was_numpy else x
        if t.dtype != torch.float32:
            raise ValueError('float32 expected')
        if tuple(t.shape[1:]) != tuple(shape_tail):
            raise ValueError('expected shape [frames, %s], got %s' % (', '.join(map(str, shape_tail)), tuple(t.shape)))
        if not t.is_cuda:
            t = t.cuda()
        return t.contiguous(), was_numpy

    def demodulate_frames(self, composite, first_frame=0, out=None):
        """composite [F, H, W] float32 (numpy or cuda tensor) -> rgb [F, 3, H, W] of the same kind."""
        torch = _torch()
        comp, was_numpy = self._as_device(composite, (self.height, self.width))
        n = comp.shape[0]
        if out is None:
            out = torch.empty((n, 3, self.height, self.width), dtype=torch.float32, device=comp.device)
        stream = torch.cuda.current_stream(comp.device).cuda_stream
        _native.check(_native.lib().cm_demodulate_frames(self._plan, comp.data_ptr(), out.data_ptr(), n,
                                                         int(first_frame), stream))
        return out.cpu().numpy() if was_numpy else out

    def demodulate_frames_u8(self, composite8, first_frame=0, out=None):
        """uint8 composite [F, H, W] -> interleaved uint8 rgb [F, H, W, 3] with ImageModem's level mapping and
        rounding fused into the kernel (every decoder except the notch / minavg instances: NotImplementedError there)."""
        torch = _torch()
        was_numpy = isinstance(composite8, numpy.ndarray)
        t = torch.from_numpy(numpy.ascontiguousarray(composite8, dtype=numpy.uint8)) if was_numpy else composite8
        if t.dtype != torch.uint8 or tuple(t.shape[1:]) != (self.height, self.width):
            raise ValueError('expected uint8 [frames, %d, %d]' % (self.height, self.width))
        t = t.cuda().contiguous() if not t.is_cuda else t.contiguous()
        n = t.shape[0]
        if out is None:
            out = torch.empty((n, self.height, self.width, 3), dtype=torch.uint8, device=t.device)
        stream = torch.cuda.current_stream(t.device).cuda_stream
        _native.check(_native.lib().cm_demodulate_frames_u8(self._plan, t.data_ptr(), out.data_ptr(), n,
                                                            int(first_frame), stream))
        return out.cpu().numpy() if was_numpy else out

    def modulate_frames(self, rgb, first_frame=0, out=None):
        torch = _torch()
        x, was_numpy = self._as_device(rgb, (3, self.height, self.width))
        n = x.shape[0]
        if out is None:
            out = torch.empty((n, self.height, self.width), dtype=torch.float32, device=x.device)
        stream = torch.cuda.current_stream(x.device).cuda_stream
        _native.check(_native.lib().cm_modulate_frames(self._plan, x.data_ptr(), out.data_ptr(), n,
                                                       int(first_frame), stream))
        return out.cpu().numpy() if was_numpy else out

    def modulate_frames_u8(self, rgb8, first_frame=0, out=None):
        """interleaved uint8 rgb [F, H, W, 3] -> uint8 composite [F, H, W]: ImageModem.modulate's byte / 255 on the way in
        and encode_composite_level + clamp + rint on the way out fused into the kernel (widths that are multiples of 16)."""
        torch = _torch()
        was_numpy = isinstance(rgb8, numpy.ndarray)
        t = torch.from_numpy(numpy.ascontiguousarray(rgb8, dtype=numpy.uint8)) if was_numpy else rgb8
        if t.dtype != torch.uint8 or tuple(t.shape[1:]) != (self.height, self.width, 3):
            raise ValueError('expected uint8 [frames, %d, %d, 3]' % (self.height, self.width))
        t = t.cuda().contiguous() if not t.is_cuda else t.contiguous()
        n = t.shape[0]
        if out is None:
            out = torch.empty((n, self.height, self.width), dtype=torch.uint8, device=t.device)
        stream = torch.cuda.current_stream(t.device).cuda_stream
        _native.check(_native.lib().cm_modulate_frames_u8(self._plan, t.data_ptr(), out.data_ptr(), n,
                                                          int(first_frame), stream))
        return out.cpu().numpy() if was_numpy else out

    # ---- runs (the per-row protocol) ----------------------------------------------------------
    def demodulate_run(self, rows, frame, first_line, k0):
        """rows [n, W] float32 numpy -> [n, 3, W] float32 numpy: what calls k0 .. k0+n-1 of a run return."""
        torch = _torch()
        x = torch.from_numpy(numpy.ascontiguousarray(rows, dtype=numpy.float32)).cuda()
        n = x.shape[0]
        out = torch.empty((n, 3, self.width), dtype=torch.float32, device=x.device)
        stream = torch.cuda.current_stream(x.device).cuda_stream
        _native.check(_native.lib().cm_demodulate_run(self._plan, x.data_ptr(), out.data_ptr(), n, int(frame),
                                                      int(first_line), int(k0), stream))
        return out.cpu().numpy()

    def modulate_run(self, rows, frame, first_line, k0):
        torch = _torch()
        x = torch.from_numpy(numpy.ascontiguousarray(rows, dtype=numpy.float32)).cuda()
        n = x.shape[0]
        out = torch.empty((n, self.width), dtype=torch.float32, device=x.device)
        stream = torch.cuda.current_stream(x.device).cuda_stream
        _native.check(_native.lib().cm_modulate_run(self._plan, x.data_ptr(), out.data_ptr(), n, int(frame),
                                                    int(first_line), int(k0), stream))
        return out.cpu().numpy()


class MacEngine(object):
    """MacModem / ColorAveragingModem(MacModem) on the cm_mac_* entry points (rows of 720 samples <-> lines of 1080)."""

    def __init__(self, modem, components=False):
        import fractions
        import scipy.signal
        from color_modem_amd.color import mac
        stack = modem._stack()
        backend = stack['backend']
        lc = backend.line_config
        std = lc.line_standard
        d = _native.MacDesc()
        d.width, d.height = int(lc.size[0]), int(lc.size[1])
        d.line_width = int(backend._width)
        d.line_shift = int(lc._line_shift)
        d.even_first = int(std.even_field_first_active_line)
        d.odd_first = int(std.odd_field_first_active_line)
        d.averaging = 1 if stack.get('mod_wrapper') == 'color_averaging' else 0
        d.resample_fir[:] = list(plan.resample_fir())
        eye = numpy.eye(3)
        d.decode_matrix[:] = list(numpy.asarray(eye if components else mac.DECODE).reshape(-1))
        d.encode_matrix[:] = list(numpy.asarray(eye if components else mac.ENCODE).reshape(-1))
        self._keep = []

        def fir(n_to, n_from):
            """the filter scipy.signal.resample_poly(x, n_to, n_from) designs (its defaults), as mac.py:49-55, 71-74, 88-91 call it"""
            fr = fractions.Fraction(n_to, n_from)
            f = _native.MacFir()
            f.up, f.down = fr.numerator, fr.denominator
            if f.up != f.down:
                max_rate = max(f.up, f.down)
                h = f.up * scipy.signal.firwin(2 * 10 * max_rate + 1, 1.0 / max_rate, window=('kaiser', 5.0))
                h = numpy.ascontiguousarray(h, dtype=numpy.float64)
                self._keep.append(h)
                f.n_taps = len(h)
                f.taps = h.ctypes.data_as(ctypes.POINTER(ctypes.c_double))
            return f

        d.luma_in = fir(mac.LUMA_WIDTH, d.width)
        d.chroma_in = fir(mac.LUMA_WIDTH // 2, d.width)
        d.line_out = fir(d.line_width, mac.LINE_WIDTH)
        d.line_in = fir(mac.LINE_WIDTH, d.line_width)
        self.desc = d
        self.in_width, self.width, self.comp_width, self.height = d.width, mac.LUMA_WIDTH, d.line_width, d.height
        self.demod_depth = 1                     # the other colour-difference signal is the previous call's
        self.mod_depth = d.averaging
        self.demodulation_delay = 0
        self.modulation_delay = d.averaging
        self._plan = ctypes.c_void_p()
        _torch()
        _native.check(_native.lib().cm_mac_plan_create(ctypes.byref(d), ctypes.byref(self._plan)))
        self._keep = []

    def __del__(self):
        p = getattr(self, '_plan', None)
        if p and _native is not None and getattr(_native, '_lib', None) is not None:
            _native._lib.cm_mac_plan_destroy(p)
            self._plan = None

    def describe(self):
        if self.in_width == 720 and self.comp_width == 1080:
            return 'mac_demod_kernel / mac_mod_kernel: one workgroup of 256 threads per 8 rows of a field, threads along the row'
        return 'mac_demod_generic_kernel / mac_mod_generic_kernel (resampling rows / lines): one workgroup per call'

    def _as_device(self, x, shape_tail):
        torch = _torch()
        was_numpy = isinstance(x, numpy.ndarray)
        t = torch.from_numpy(numpy.ascontiguousarray(x, dtype=numpy.float32)) if was_numpy else x
        if t.dtype != torch.float32:
            raise ValueError('float32 expected')
        if tuple(t.shape[1:]) != tuple(shape_tail):
            raise ValueError('expected shape [frames, %s], got %s' % (', '.join(map(str, shape_tail)), tuple(t.shape)))
        if not t.is_cuda:
            t = t.cuda()
        return t.contiguous(), was_numpy

    def demodulate_frames(self, composite, first_frame=0, out=None):
        """composite [F, H, 1080] float32 -> rgb [F, 3, H, 720] (numpy in -> numpy out, cuda tensor in -> cuda tensor out)."""
        torch = _torch()
        comp, was_numpy = self._as_device(composite, (self.height, self.comp_width))
        n = comp.shape[0]
        if out is None:
            out = torch.empty((n, 3, self.height, self.width), dtype=torch.float32, device=comp.device)
        stream = torch.cuda.current_stream(comp.device).cuda_stream
        _native.check(_native.lib().cm_mac_demodulate_frames(self._plan, comp.data_ptr(), out.data_ptr(), n,
                                                             int(first_frame), stream))
        return out.cpu().numpy() if was_numpy else out

    def modulate_frames(self, rgb, first_frame=0, out=None):
        """rgb [F, 3, H, W] float32 -> composite [F, H, line width]."""
        torch = _torch()
        x, was_numpy = self._as_device(rgb, (3, self.height, self.in_width))
        n = x.shape[0]
        if self.height < 2 * self.modulation_delay and n:
            raise IndexError('image.py:49-50 feeds row 1 ahead of a field under modulation_delay 1: the image has %d row(s)' % self.height)
        if out is None:
            out = torch.empty((n, self.height, self.comp_width), dtype=torch.float32, device=x.device)
        stream = torch.cuda.current_stream(x.device).cuda_stream
        _native.check(_native.lib().cm_mac_modulate_frames(self._plan, x.data_ptr(), out.data_ptr(), n,
                                                           int(first_frame), stream))
        return out.cpu().numpy() if was_numpy else out

    def _as_device_u8(self, x, shape_tail):
        torch = _torch()
        was_numpy = isinstance(x, numpy.ndarray)
        t = torch.from_numpy(numpy.ascontiguousarray(x, dtype=numpy.uint8)) if was_numpy else x
        if t.dtype != torch.uint8 or tuple(t.shape[1:]) != tuple(shape_tail):
            raise ValueError('expected uint8 [frames, %s]' % ', '.join(map(str, shape_tail)))
        return (t.cuda() if not t.is_cuda else t).contiguous(), was_numpy

    def demodulate_frames_u8(self, composite8, first_frame=0, out=None):
        """uint8 lines [F, H, line width] -> interleaved uint8 rgb [F, H, 720, 3], ImageModem's level mapping and rounding
        fused into the kernel (the resampling kernels serve every shape here)."""
        torch = _torch()
        t, was_numpy = self._as_device_u8(composite8, (self.height, self.comp_width))
        n = t.shape[0]
        if out is None:
            out = torch.empty((n, self.height, self.width, 3), dtype=torch.uint8, device=t.device)
        stream = torch.cuda.current_stream(t.device).cuda_stream
        _native.check(_native.lib().cm_mac_demodulate_frames_u8(self._plan, t.data_ptr(), out.data_ptr(), n, int(first_frame), stream))
        return out.cpu().numpy() if was_numpy else out

    def modulate_frames_u8(self, rgb8, first_frame=0, out=None):
        """interleaved uint8 rgb [F, H, W, 3] -> uint8 lines [F, H, line width]."""
        torch = _torch()
        t, was_numpy = self._as_device_u8(rgb8, (self.height, self.in_width, 3))
        n = t.shape[0]
        if self.height < 2 * self.modulation_delay and n:
            raise IndexError('image.py:49-50 feeds row 1 ahead of a field under modulation_delay 1: the image has %d row(s)' % self.height)
        if out is None:
            out = torch.empty((n, self.height, self.comp_width), dtype=torch.uint8, device=t.device)
        stream = torch.cuda.current_stream(t.device).cuda_stream
        _native.check(_native.lib().cm_mac_modulate_frames_u8(self._plan, t.data_ptr(), out.data_ptr(), n, int(first_frame), stream))
        return out.cpu().numpy() if was_numpy else out

    def demodulate_run(self, rows, frame, first_line, k0):
        torch = _torch()
        x = torch.from_numpy(numpy.ascontiguousarray(rows, dtype=numpy.float32)).cuda()
        n = x.shape[0]
        out = torch.empty((n, 3, self.width), dtype=torch.float32, device=x.device)
        stream = torch.cuda.current_stream(x.device).cuda_stream
        _native.check(_native.lib().cm_mac_demodulate_run(self._plan, x.data_ptr(), out.data_ptr(), n,
                                                          int(frame), int(first_line), int(k0), stream))
        return out.cpu().numpy()

    def modulate_run(self, rows, frame, first_line, k0):
        torch = _torch()
        x = torch.from_numpy(numpy.ascontiguousarray(rows, dtype=numpy.float32)).cuda()
        n = x.shape[0]
        out = torch.empty((n, self.comp_width), dtype=torch.float32, device=x.device)
        stream = torch.cuda.current_stream(x.device).cuda_stream
        _native.check(_native.lib().cm_mac_modulate_run(self._plan, x.data_ptr(), out.data_ptr(), n,
                                                        int(frame), int(first_line), int(k0), stream))
        return out.cpu().numpy()
