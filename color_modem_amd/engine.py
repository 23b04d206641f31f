# -*- coding: utf-8 -*-
"""Device engine: owns one plan (cm_plan) per modem stack and moves rows / frames through it.

torch (ROCm build) is used only as the owner of device memory and of the HIP stream; all
arithmetic happens in libcolor_modem_hip.so.
"""

import ctypes

import numpy

from color_modem_amd import _native, plan


_TORCH = None


def _torch():
    """torch, once a HIP device has been seen (the check is a driver call: made once, not on every row of the per-row protocol)"""
    global _TORCH
    if _TORCH is None:
        import torch
        if not torch.cuda.is_available():
            raise _native.NativeError('no HIP device visible to torch: color_modem_amd runs on the GPU only')
        _TORCH = torch
    return _TORCH


def _check_out(out, shape, dtype, device):
    """A caller-supplied result tensor goes to the kernels as a raw pointer: refuse anything they would write out of bounds
    or garble (wrong shape / dtype / device, or a non-contiguous view)."""
    torch = _torch()
    if not torch.is_tensor(out):
        raise ValueError('out= must be a torch tensor on the input\'s device')
    if tuple(out.shape) != tuple(shape):
        raise ValueError('out= has shape %s, the result has shape %s' % (tuple(out.shape), tuple(shape)))
    if out.dtype != dtype:
        raise ValueError('out= has dtype %s, the result has dtype %s' % (out.dtype, dtype))
    if out.device != device:
        raise ValueError('out= lives on %s, the input on %s' % (out.device, device))
    if not out.is_contiguous():
        raise ValueError('out= must be contiguous')
    return out


class _DevicePlans(object):
    """One native plan per HIP device, created on first use under that device (a plan's tables live on the device that
    was current when it was created, and the library refuses to run it under another one)."""

    def __init__(self, create, destroy):
        self._create, self._destroy, self._plans = create, destroy, {}
        self.on_create = None     # called with every new plan handle (Engine.set_small_batch re-applies its mode)

    def get(self, device):
        torch = _torch()
        index = device.index if device.index is not None else torch.cuda.current_device()
        handle = self._plans.get(index)
        if handle is None:
            handle = ctypes.c_void_p()
            with torch.cuda.device(index):
                _native.check(self._create(ctypes.byref(handle)))
            self._plans[index] = handle
            if self.on_create is not None:
                self.on_create(handle)
        return handle

    def handles(self):
        return list(self._plans.values())

    def close(self):
        plans, self._plans = self._plans, {}
        if _native is None or getattr(_native, '_lib', None) is None:
            return
        for handle in plans.values():
            if handle:
                self._destroy(handle)


class _NoContext(object):
    def __enter__(self):
        return self

    def __exit__(self, *exc):
        return False


_NO_CONTEXT = _NoContext()


class RowSession(object):
    """The per-row protocol (Modem.demodulate / Modem.modulate, one row per call) without re-sending history: the rows of
    the current run stay on the device in call order, a call uploads ONE row from a pinned staging buffer, launches the
    run entry point on the last depth + 1 rows and downloads the one result row into a pinned buffer - two small
    asynchronous copies, one launch, one stream synchronisation (tools/row_api_bench.py).  A session that missed calls of
    the run (the caller switched between demodulate() and demodulate_components(), which run on different plans) is
    given the run's last rows again."""

    SLOTS = 64     # rows of history buffer; when it is full the last `depth` rows move to its front
    ZERO_COPY = True

    def __init__(self, eng, direction):
        torch = _torch()
        self.eng, self.direction = eng, direction
        dev = torch.device('cuda', torch.cuda.current_device())
        L = _native.lib()
        prefix = getattr(eng, 'abi_prefix', None)
        hook = getattr(eng, 'row_call', None)          # a composition of plans (wrapped.py) brings its own entry point
        if direction == 'demod':
            self.in_shape, self.out_shape, self.depth = (eng.comp_width,), (3, eng.width), eng.demod_depth
            self.fn = hook(direction) if hook else getattr(L, prefix + 'demodulate_run')
        else:
            self.in_shape, self.out_shape, self.depth = (3, eng.in_width), (eng.comp_width,), eng.mod_depth
            self.fn = hook(direction) if hook else getattr(L, prefix + 'modulate_run')
        self.plans_of = eng._plans.get if hasattr(eng._plans, 'get') else eng._plans
        # an encoder whose tables start above the picture (make_engine: line_offset): the line numbers it is passed move with them
        self.line_offset = int(getattr(eng, 'line_offset', 0)) if direction == 'mod' else 0
        # ZERO_COPY: history and result rows live in pinned host memory, which the device reads and writes over the bus -
        # a call is one kernel launch and one synchronisation, no copy is enqueued (a row is 3 - 9 KB: latency, not
        # bandwidth).  Otherwise: device-resident history, one small upload and one download per call.
        self.zero_copy = bool(self.ZERO_COPY)
        if self.zero_copy:
            self.hist = torch.empty((self.SLOTS,) + self.in_shape, dtype=torch.float32).pin_memory()
            self.out = torch.empty((self.depth + 1,) + self.out_shape, dtype=torch.float32).pin_memory()
            self.np_hist, self.np_outs = self.hist.numpy(), self.out.numpy()
            self.hist_ptr, self.out_ptr = self.hist.data_ptr(), self.out.data_ptr()
            self.row_bytes = self.hist[0].numel() * 4
        else:
            self.hist = torch.empty((self.SLOTS,) + self.in_shape, dtype=torch.float32, device=dev)
            self.out = torch.empty((self.depth + 1,) + self.out_shape, dtype=torch.float32, device=dev)
            self.pin_in = torch.empty((self.depth + 1,) + self.in_shape, dtype=torch.float32).pin_memory()
            self.pin_out = torch.empty(self.out_shape, dtype=torch.float32).pin_memory()
            self.np_in, self.np_out = self.pin_in.numpy(), self.pin_out.numpy()
        self.pos = -1                  # slot of the newest row
        self.held = (None, -1)         # (run token, k) of the newest row on the device
        self.device = dev

    def step(self, rows, token, frame, line, k):
        """rows: the run's last rows (numpy float32 of in_shape each, newest last, at least min(k, depth) + 1 of them);
        the newest is call k (k = 0: first after a reset) at `line` of `frame`; token identifies the run.  Returns the
        call's result as a float64 numpy array of out_shape."""
        torch = _torch()
        n = min(k, self.depth) + 1                     # rows the kernels look at
        fresh = n if self.held != (token, k - 1) else 1
        if self.zero_copy:
            if fresh == n:
                self.pos = -1
            elif self.pos + 1 >= self.SLOTS:           # compact: the last n - 1 rows to the front
                keep = n - 1
                self.np_hist[:keep] = self.np_hist[self.pos + 1 - keep:self.pos + 1].copy()
                self.pos = keep - 1
            for j in range(fresh):
                self.np_hist[self.pos + 1 + j] = rows[len(rows) - fresh + j]
            self.pos += fresh
            self.held = (None, -1)
            first = self.pos - (n - 1)
            current = torch.cuda.current_device() == self.device.index
            ctx = _NO_CONTEXT if current else torch.cuda.device(self.device)
            with ctx:
                stream = torch.cuda.current_stream(self.device)
                try:
                    _native.check(self.fn(self.plans_of(self.device), self.hist_ptr + first * self.row_bytes, self.out_ptr, n,
                                          int(frame), int(line) - 2 * (n - 1) + self.line_offset, int(k) - (n - 1), stream.cuda_stream))
                finally:
                    stream.synchronize()
            self.held = (token, k)
            return self.np_outs[n - 1].astype(numpy.float64)
        with torch.cuda.device(self.device):
            stream = torch.cuda.current_stream(self.device)
            if fresh == n:
                self.pos = -1
            elif self.pos + 1 >= self.SLOTS:           # compact: the last n - 1 rows to the front
                keep = n - 1
                self.hist[:keep].copy_(self.hist[self.pos + 1 - keep:self.pos + 1].clone())
                self.pos = keep - 1
            for j in range(fresh):
                self.np_in[j] = rows[len(rows) - fresh + j]
            self.hist[self.pos + 1:self.pos + 1 + fresh].copy_(self.pin_in[:fresh], non_blocking=True)
            self.pos += fresh
            self.held = (None, -1)                     # until the call has gone through: a failed call re-sends its rows
            first = self.pos - (n - 1)
            try:
                _native.check(self.fn(self.plans_of(self.device), self.hist[first].data_ptr(), self.out.data_ptr(), n, int(frame),
                                      int(line) - 2 * (n - 1) + self.line_offset, int(k) - (n - 1), stream.cuda_stream))
                self.pin_out.copy_(self.out[n - 1], non_blocking=True)
            finally:
                stream.synchronize()                   # also when the call is refused: the upload from pin_in must have landed
            self.held = (token, k)
        return self.np_out.astype(numpy.float64)


def stack_backend(modem):
    """the innermost modem of a stack (the one that owns the filters)"""
    return modem._stack().get('backend', modem)


def _custom_avg(stack):
    """a comb wrapper with an avg= callable of the caller's own (ref comb.py:72, 81-84): not one of the two the fused tables express"""
    from color_modem_amd import comb
    fn = stack.get('wrapper_avg')
    return fn is not None and fn is not comb.avg and fn is not comb.minavg


class _OffsetStack(object):
    """A stack whose modulator tables start `line_offset` lines above the picture (plan.QamTables: line_offset)."""

    def __init__(self, stack, line_offset):
        self._shifted = dict(stack, line_offset=int(line_offset))

    def _stack(self):
        return self._shifted


def make_engine(modem, components=False, strip_chroma=True, min_lines=0, line_offset=0):
    """The engine of a modem stack: a cm_plan for the QAM / SECAM families, the plan-less MAC entry points for MacModem.
    line_offset (generic.py's encoders only): modulate_run accepts line numbers down to -line_offset."""
    from color_modem_amd import generic
    if generic.needs_generic(modem):       # a wrapper inside a wrapper, wrappers around Pal3DModem(avg=f) / the NIIR modems: level by level (round 6)
        return generic.make(modem, components, strip_chroma, min_lines, line_offset)
    if line_offset:
        eng = make_engine(modem, components, strip_chroma, min_lines)
        if isinstance(eng, Engine):
            return Engine(_OffsetStack(modem._stack(), line_offset), components, strip_chroma, min_lines)
        if isinstance(eng, (AmEngine, MacEngine)):       # line geometry per lane on the device, any line number (cm_am_stages.h: AmLine; cm_mac_kernels.h)
            return eng
        if getattr(eng, 'encoder', None) is not None and isinstance(eng.encoder, Engine):
            # a composition whose encoder is its leaf's engine (comb.py:90-94): that one gets the offset tables
            eng.encoder = Engine(_OffsetStack(eng.encoder._modem_stack, line_offset), eng.encoder._components, True, min_lines)
            return eng
        raise NotImplementedError('%s has no encoder run above the picture' % type(eng).__name__)
    stack = modem._stack()
    kind = stack['kind']
    if stack.get('demod_wrapper') and _custom_avg(stack) and not stack.get('mod_wrapper') and int(stack['backend'].line_config.size[0]) % 4:
        # avg= callables at widths that are not a multiple of 4: the component buffer of wrapped.py's composition moves 16-byte vectors; the
        # level-by-level engine runs the same statements through the engines' run entry points, which stage such rows by themselves
        return generic.GenericCombEngine(modem, components, strip_chroma, min_lines)
    from color_modem_amd import notched
    if notched.shifted_notch(stack, strip_chroma) is not None:      # notch= values whose FilterFunction shift is not 0: the notch as a pass of its own
        return notched.ShiftedNotchEngine(modem, components, strip_chroma, min_lines)
    from color_modem_amd import pal3d_callable
    if pal3d_callable.custom_avg(stack) is not None:      # Pal3DModem(avg=f): its two estimates from two plans, f applied in between
        return pal3d_callable.Pal3DCallableEngine(modem, components, strip_chroma, min_lines)
    if stack.get('demod_wrapper') and (kind in ('pal_d', 'pal_3d') or (kind in ('pal_s', 'ntsc', 'ntsc_comb') and _custom_avg(stack) and not stack.get('mod_wrapper'))):
        from color_modem_amd import wrapped
        return wrapped.WrappedCombEngine(modem, components, strip_chroma, min_lines)
    if kind == 'mac':
        return MacEngine(modem, components)
    if kind in ('protosecam', 'niir'):
        return AmEngine(modem, components, strip_chroma)
    return Engine(modem, components, strip_chroma, min_lines)


class _EngineBase(object):
    """Shared plumbing: input staging, result validation, and the launch on the input tensor's device and stream."""
    abi_prefix = 'cm_'       # entry point family of include/color_modem_hip.h: cm_*, cm_mac_*, cm_am_*

    def __del__(self):
        plans = getattr(self, '_plans', None)
        if plans is not None:
            plans.close()

    def set_small_batch(self, mode):
        """An engine with one kernel family (MacEngine: its kernels spread a row over the lanes whatever the batch) takes the pins that
        leave it as it is; Engine / AmEngine override this with their plans' setters."""
        if mode not in ('auto', 'rows'):
            raise NotImplementedError('%s has no %r small-batch kernels' % (type(self).__name__, mode))

    @property
    def _plan(self):
        """the plan of the current device (created on first use)"""
        torch = _torch()
        return self._plans.get(torch.device('cuda', torch.cuda.current_device()))

    def _stage(self, x, dtype, shape_tail, what):
        """numpy array or torch tensor -> contiguous tensor on a HIP device; (tensor, came from numpy)"""
        torch = _torch()
        was_numpy = isinstance(x, numpy.ndarray)
        np_dtype = numpy.float32 if dtype == torch.float32 else numpy.uint8
        t = torch.from_numpy(numpy.ascontiguousarray(x, dtype=np_dtype)) if was_numpy else x
        if not torch.is_tensor(t):
            raise ValueError('%s: numpy array or torch tensor expected' % what)
        if t.dtype != dtype:
            raise ValueError('%s: %s expected' % (what, 'float32' if dtype == torch.float32 else 'uint8'))
        if t.dim() != len(shape_tail) + 1 or tuple(t.shape[1:]) != tuple(shape_tail):
            raise ValueError('%s: expected shape [n, %s], got %s' % (what, ', '.join(map(str, shape_tail)), tuple(t.shape)))
        if not t.is_cuda:
            t = t.cuda()
        return t.contiguous(), was_numpy

    def _launch(self, fn, x, out, out_shape, out_dtype, was_numpy, *args):
        torch = _torch()
        if out is None:
            out = torch.empty(out_shape, dtype=out_dtype, device=x.device)
        else:
            _check_out(out, out_shape, out_dtype, x.device)
        plan_handle = self._plans.get(x.device)
        if x.device.index == torch.cuda.current_device():      # (the usual case: no device switch around the launch)
            stream = torch.cuda.current_stream().cuda_stream
            rc = fn(plan_handle, x.data_ptr(), out.data_ptr(), *args, stream)
        else:
            with torch.cuda.device(x.device):
                stream = torch.cuda.current_stream(x.device).cuda_stream
                rc = fn(plan_handle, x.data_ptr(), out.data_ptr(), *args, stream)
        if rc:
            _native.check(rc)
        return out.cpu().numpy() if was_numpy else out


class Engine(_EngineBase):
    def __init__(self, modem, components=False, strip_chroma=True, min_lines=0):
        self.built = plan.build_plan(modem, components, strip_chroma, min_lines)
        self._modem_stack, self._components = modem._stack(), bool(components)
        self.line_offset = int(self._modem_stack.get('line_offset', 0))      # modulator tables start this many lines above the picture (make_engine)
        self.n_lines = int(self.built.desc.demod_main.n_lines or self.built.desc.mod_main.n_lines) - self.line_offset
        d = self.built.desc
        self.width, self.height = d.width, d.height
        self.comp_width = d.width
        self.in_width = d.width
        self.demod_depth = d.depth
        self.mod_depth = 1 if d.modulation_delay else 0
        self.demodulation_delay = d.demodulation_delay
        self.modulation_delay = d.modulation_delay
        L = _native.lib()
        self._plans = _DevicePlans(lambda out: L.cm_plan_create(ctypes.byref(d), out), L.cm_plan_destroy)
        self._plan      # no usable device, or no kernel instance for this stack: fail at construction

    def describe(self):
        buf = ctypes.create_string_buffer(512)
        _native.lib().cm_plan_describe(self._plan, buf, 512)
        return buf.value.decode()

    def has_fused_u8(self, direction):
        """ImageModem's byte boundary inside the kernels (cm_*_frames_u8): every decoder instance at widths that are multiples of 4, every
        encoder at multiples of 16 (the byte tiles)"""
        return self.width % (4 if direction == 'demod' else 16) == 0

    SMALL_BATCH = {'auto': 0, 'rows': 1, 'segments': 2, 'scan': 3}

    def set_small_batch(self, mode):
        """Pin how the decoder runs small batches (cm_plan_set_small_batch): 'auto' (default: the row-parallel scan kernel
        below a few frames where the plan has one, else row segments), 'rows' (the streaming kernel on whole rows),
        'segments', 'scan' (NotImplementedError where the plan's shape does not fit the scan kernel)."""
        code = self.SMALL_BATCH[mode]
        for handle in self._plans.handles():
            _native.check(_native.lib().cm_plan_set_small_batch(handle, code))
        self._plans.on_create = (lambda h: _native.check(_native.lib().cm_plan_set_small_batch(h, code))) if code else None

    # ---- frames -------------------------------------------------------------------------------
    def demodulate_frames(self, composite, first_frame=0, out=None):
        """composite [F, H, W] float32 (numpy or cuda tensor) -> rgb [F, 3, H, W] of the same kind."""
        torch = _torch()
        comp, was_numpy = self._stage(composite, torch.float32, (self.height, self.width), 'composite')
        n = comp.shape[0]
        return self._launch(_native.lib().cm_demodulate_frames, comp, out, (n, 3, self.height, self.width), torch.float32,
                            was_numpy, n, int(first_frame))

    def demodulate_frames_u8(self, composite8, first_frame=0, out=None):
        """uint8 composite [F, H, W] -> interleaved uint8 rgb [F, H, W, 3] with ImageModem's level mapping and
        rounding fused into the kernel (every decoder except the notch / minavg instances: NotImplementedError there)."""
        torch = _torch()
        t, was_numpy = self._stage(composite8, torch.uint8, (self.height, self.width), 'composite8')
        n = t.shape[0]
        return self._launch(_native.lib().cm_demodulate_frames_u8, t, out, (n, self.height, self.width, 3), torch.uint8,
                            was_numpy, n, int(first_frame))

    def modulate_frames(self, rgb, first_frame=0, out=None):
        torch = _torch()
        x, was_numpy = self._stage(rgb, torch.float32, (3, self.height, self.width), 'rgb')
        n = x.shape[0]
        return self._launch(_native.lib().cm_modulate_frames, x, out, (n, self.height, self.width), torch.float32,
                            was_numpy, n, int(first_frame))

    def modulate_frames_u8(self, rgb8, first_frame=0, out=None):
        """interleaved uint8 rgb [F, H, W, 3] -> uint8 composite [F, H, W]: ImageModem.modulate's byte / 255 on the way in
        and encode_composite_level + clamp + rint on the way out fused into the kernel (widths that are multiples of 16)."""
        torch = _torch()
        t, was_numpy = self._stage(rgb8, torch.uint8, (self.height, self.width, 3), 'rgb8')
        n = t.shape[0]
        return self._launch(_native.lib().cm_modulate_frames_u8, t, out, (n, self.height, self.width), torch.uint8,
                            was_numpy, n, int(first_frame))

    # ---- runs (the per-row protocol) ----------------------------------------------------------
    def demodulate_run(self, rows, frame, first_line, k0):
        """rows [n, W] float32 (numpy or cuda tensor) -> [n, 3, W] of the same kind: what calls k0 .. k0+n-1 of a run return."""
        torch = _torch()
        if self.line_offset:
            raise NotImplementedError('an engine with shifted modulator tables (line_offset) encodes only')
        x, was_numpy = self._stage(rows, torch.float32, (self.width,), 'rows')
        n = x.shape[0]
        return self._launch(_native.lib().cm_demodulate_run, x, None, (n, 3, self.width), torch.float32, was_numpy,
                            n, int(frame), int(first_line), int(k0))

    def modulate_run(self, rows, frame, first_line, k0):
        torch = _torch()
        x, was_numpy = self._stage(rows, torch.float32, (3, self.width), 'rows')
        n = x.shape[0]
        return self._launch(_native.lib().cm_modulate_run, x, None, (n, self.width), torch.float32, was_numpy,
                            n, int(frame), int(first_line) + self.line_offset, int(k0))


class MacEngine(_EngineBase):
    """MacModem / ColorAveragingModem(MacModem) on the cm_mac_* entry points (rows of 720 samples <-> lines of 1080)."""
    abi_prefix = 'cm_mac_'

    def __init__(self, modem, components=False):
        import fractions
        from color_modem_amd import design
        from color_modem_amd.color import mac
        stack = modem._stack()
        if stack.get('demod_wrapper'):
            # the reference fails the same way: MacModem has no demodulate_components for a comb wrapper to call (comb.py:98)
            raise AttributeError('MacModem has no demodulate_components: a comb wrapper cannot sit on it (ref comb.py:98, mac.py)')
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
        self._keep = []          # the tap arrays the descriptor points to (a plan per device is created on first use)

        def fir(n_to, n_from):
            """the filter scipy.signal.resample_poly(x, n_to, n_from) designs (its defaults), as mac.py:49-55, 71-74, 88-91 call it"""
            fr = fractions.Fraction(n_to, n_from)
            f = _native.MacFir()
            f.up, f.down = fr.numerator, fr.denominator
            if f.up != f.down:
                max_rate = max(f.up, f.down)
                h = f.up * design.resample_poly_fir(max_rate)
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
        L = _native.lib()
        self._plans = _DevicePlans(lambda out: L.cm_mac_plan_create(ctypes.byref(d), out), L.cm_mac_plan_destroy)
        self._plan

    def describe(self):
        if self.in_width == 720 and self.comp_width == 1080:
            return 'mac_demod_kernel / mac_mod_kernel: one workgroup of 256 threads per 8 rows of a field, threads along the row'
        return 'mac_demod_generic_kernel / mac_mod_generic_kernel (resampling rows / lines): one workgroup per call'

    def has_fused_u8(self, direction):
        return True        # the resampling kernels carry the byte boundary for every shape

    def _needs_rows(self, n):
        if self.height < 2 * self.modulation_delay and n:
            raise IndexError('image.py:49-50 feeds row 1 ahead of a field under modulation_delay 1: the image has %d row(s)' % self.height)

    def demodulate_frames(self, composite, first_frame=0, out=None):
        """composite [F, H, 1080] float32 -> rgb [F, 3, H, 720] (numpy in -> numpy out, cuda tensor in -> cuda tensor out)."""
        torch = _torch()
        comp, was_numpy = self._stage(composite, torch.float32, (self.height, self.comp_width), 'composite')
        n = comp.shape[0]
        return self._launch(_native.lib().cm_mac_demodulate_frames, comp, out, (n, 3, self.height, self.width), torch.float32,
                            was_numpy, n, int(first_frame))

    def modulate_frames(self, rgb, first_frame=0, out=None):
        """rgb [F, 3, H, W] float32 -> composite [F, H, line width]."""
        torch = _torch()
        x, was_numpy = self._stage(rgb, torch.float32, (3, self.height, self.in_width), 'rgb')
        n = x.shape[0]
        self._needs_rows(n)
        return self._launch(_native.lib().cm_mac_modulate_frames, x, out, (n, self.height, self.comp_width), torch.float32,
                            was_numpy, n, int(first_frame))

    def demodulate_frames_u8(self, composite8, first_frame=0, out=None):
        """uint8 lines [F, H, line width] -> interleaved uint8 rgb [F, H, 720, 3], ImageModem's level mapping and rounding
        fused into the kernel (the resampling kernels serve every shape here)."""
        torch = _torch()
        t, was_numpy = self._stage(composite8, torch.uint8, (self.height, self.comp_width), 'composite8')
        n = t.shape[0]
        return self._launch(_native.lib().cm_mac_demodulate_frames_u8, t, out, (n, self.height, self.width, 3), torch.uint8,
                            was_numpy, n, int(first_frame))

    def modulate_frames_u8(self, rgb8, first_frame=0, out=None):
        """interleaved uint8 rgb [F, H, W, 3] -> uint8 lines [F, H, line width]."""
        torch = _torch()
        t, was_numpy = self._stage(rgb8, torch.uint8, (self.height, self.in_width, 3), 'rgb8')
        n = t.shape[0]
        self._needs_rows(n)
        return self._launch(_native.lib().cm_mac_modulate_frames_u8, t, out, (n, self.height, self.comp_width), torch.uint8,
                            was_numpy, n, int(first_frame))

    def demodulate_run(self, rows, frame, first_line, k0):
        torch = _torch()
        x, was_numpy = self._stage(rows, torch.float32, (self.comp_width,), 'rows')
        n = x.shape[0]
        return self._launch(_native.lib().cm_mac_demodulate_run, x, None, (n, 3, self.width), torch.float32, was_numpy,
                            n, int(frame), int(first_line), int(k0))

    def modulate_run(self, rows, frame, first_line, k0):
        torch = _torch()
        x, was_numpy = self._stage(rows, torch.float32, (3, self.in_width), 'rows')
        n = x.shape[0]
        return self._launch(_native.lib().cm_mac_modulate_run, x, None, (n, self.comp_width), torch.float32, was_numpy,
                            n, int(frame), int(first_line), int(k0))


class AmEngine(_EngineBase):
    """ProtoSecamModem / ColorAveragingModem(ProtoSecamModem) / NiirModem / HueCorrectingNiirModem on the cm_am_* entry
    points (the amplitude-modulated line-sequential standards: x3 resampling around recursive filters)."""
    abi_prefix = 'cm_am_'

    def __init__(self, modem, components=False, strip_chroma=True):
        from color_modem_amd import plan_am
        d = plan_am.build_am_desc(modem, components, strip_chroma)
        self.desc = d
        self.width = self.comp_width = self.in_width = d.width
        self.height = d.height
        self.demod_depth = 1                     # the other colour-difference signal / the phase reference is the previous call's
        self.mod_depth = 1 if d.averaging else 0
        self.demodulation_delay = 0
        self.modulation_delay = 1 if d.averaging else 0
        self.n_lines = 1 << 30                   # no per-line tables: the line's phase is computed on the device
        # NiirModem(noise_level != 0): niir.py:45-46 perturbs the hue in modulate() (not in the plain modem's
        # modulate_components, niir.py:82-83); HueCorrectingNiirModem does it in both (niir.py:176-177, 193-194)
        level = float(getattr(stack_backend(modem), '_noise_level', 0.0))
        hue = bool(getattr(stack_backend(modem), 'hue_correcting', False))
        self.noise_level = level if (level != 0.0 and (hue or not components)) else 0.0
        self.composite_mod = self.noise_level != 0.0     # rowapi: the noisy encoder takes its run through modulate_run
        L = _native.lib()
        self._plans = _DevicePlans(lambda out: L.cm_am_plan_create(ctypes.byref(d), out), L.cm_am_plan_destroy)
        self._plan

    def has_fused_u8(self, direction):
        if direction == 'demod':
            return self.width % 4 == 0
        return self.width % 16 == 0 and self.noise_level == 0.0      # (the noisy NIIR encoder has no byte form)

    def _noise(self, calls, newest_only=False):
        """(numpy.random.random_sample(W) - 0.5) * noise_level for db, then dr, per call in call order - exactly the
        reference's draws (niir.py:45-46), so numpy.random.seed() reproduces its output.  newest_only: a run submitted with
        history re-computes older calls whose results are dropped; only the newest call draws."""
        torch = _torch()
        z = numpy.zeros((calls, 2, self.width), dtype=numpy.float32)
        if newest_only:
            z[-1] = (numpy.random.random_sample((2, self.width)) - 0.5) * self.noise_level
        else:
            z[:] = (numpy.random.random_sample((calls, 2, self.width)) - 0.5) * self.noise_level
        return torch.from_numpy(z)

    def set_small_batch(self, mode):
        """Engine.set_small_batch for a Proto-SECAM / NIIR plan (cm_am_plan_set_small_batch): 'auto' (the row-parallel scan kernels of
        csrc/cm_am_scan_kernels.h below a few frames, where the plan's shape fits them), 'rows' (the streaming wave pairs on whole rows),
        'scan' (NotImplementedError where no scan kernel serves the plan: rows beyond ~1000 samples)."""
        code = Engine.SMALL_BATCH[mode]
        for handle in self._plans.handles():
            _native.check(_native.lib().cm_am_plan_set_small_batch(handle, code))
        self._plans.on_create = (lambda h: _native.check(_native.lib().cm_am_plan_set_small_batch(h, code))) if code else None
        self._mode = mode

    def describe(self):
        name = 'proto' if self.desc.kind == 1 else 'niir'
        hue = ', hue path float64' if name == 'niir' else ''
        return ('%s_demod_pair_kernel / %s_mod%s_kernel (wave pairs, one lane per call, x3 polyphase resamplers in registers%s); small batches: '
                '%s_demod_scan_kernel / %s_mod_scan_kernel (one wavefront per call); small-batch mode: %s'
                % (name, name, '' if name == 'niir' else '_pair', hue, name, name, getattr(self, '_mode', 'auto')))

    def demodulate_frames(self, composite, first_frame=0, out=None):
        """composite [F, H, W] float32 -> rgb [F, 3, H, W] (numpy in -> numpy out, cuda tensor in -> cuda tensor out)."""
        torch = _torch()
        comp, was_numpy = self._stage(composite, torch.float32, (self.height, self.width), 'composite')
        n = comp.shape[0]
        return self._launch(_native.lib().cm_am_demodulate_frames, comp, out, (n, 3, self.height, self.width), torch.float32,
                            was_numpy, n, int(first_frame))

    def modulate_frames(self, rgb, first_frame=0, out=None):
        """rgb [F, 3, H, W] float32 -> composite [F, H, W]."""
        torch = _torch()
        x, was_numpy = self._stage(rgb, torch.float32, (3, self.height, self.width), 'rgb')
        n = x.shape[0]
        if self.height < 2 * self.modulation_delay and n:
            raise IndexError('image.py:49-50 feeds row 1 ahead of a field under modulation_delay 1: the image has %d row(s)' % self.height)
        if self.noise_level != 0.0 and n:
            d = self.modulation_delay
            calls = n * (((self.height + 1) // 2 + d) + ((self.height // 2 + d) if self.height > 1 else 0))
            noise = self._noise(calls).to(x.device)
            fn = _native.lib().cm_am_modulate_frames_noise
            return self._launch(lambda plan, a, b, *rest: fn(plan, a, noise.data_ptr(), b, *rest), x, out, (n, self.height, self.width),
                                torch.float32, was_numpy, n, int(first_frame))
        return self._launch(_native.lib().cm_am_modulate_frames, x, out, (n, self.height, self.width), torch.float32,
                            was_numpy, n, int(first_frame))

    def demodulate_frames_u8(self, composite8, first_frame=0, out=None):
        """'L' bytes [F, H, W] -> interleaved 'RGB' bytes [F, H, W, 3] (image.py:58-84 fused into the kernel)."""
        torch = _torch()
        if self.width % 4:
            raise NotImplementedError('the fused uint8 boundary needs a width that is a multiple of 4')
        t, was_numpy = self._stage(composite8, torch.uint8, (self.height, self.width), 'composite8')
        n = t.shape[0]
        return self._launch(_native.lib().cm_am_demodulate_frames_u8, t, out, (n, self.height, self.width, 3), torch.uint8,
                            was_numpy, n, int(first_frame))

    def modulate_frames_u8(self, rgb8, first_frame=0, out=None):
        """interleaved 'RGB' bytes [F, H, W, 3] -> 'L' bytes [F, H, W] (image.py:27-56 fused into the kernel)."""
        torch = _torch()
        if self.width % 16:
            raise NotImplementedError('the fused uint8 boundary of the encoders needs a width that is a multiple of 16')
        if self.noise_level != 0.0:
            raise NotImplementedError('the noisy NIIR encoder has no byte form: ImageModem converts on the device around the float path')
        t, was_numpy = self._stage(rgb8, torch.uint8, (self.height, self.width, 3), 'rgb8')
        n = t.shape[0]
        if self.height < 2 * self.modulation_delay and n:
            raise IndexError('image.py:49-50 feeds row 1 ahead of a field under modulation_delay 1: the image has %d row(s)' % self.height)
        return self._launch(_native.lib().cm_am_modulate_frames_u8, t, out, (n, self.height, self.width), torch.uint8,
                            was_numpy, n, int(first_frame))

    def demodulate_run(self, rows, frame, first_line, k0):
        torch = _torch()
        x, was_numpy = self._stage(rows, torch.float32, (self.width,), 'rows')
        n = x.shape[0]
        return self._launch(_native.lib().cm_am_demodulate_run, x, None, (n, 3, self.width), torch.float32, was_numpy,
                            n, int(frame), int(first_line), int(k0))

    def modulate_run(self, rows, frame, first_line, k0):
        torch = _torch()
        x, was_numpy = self._stage(rows, torch.float32, (3, self.width), 'rows')
        n = x.shape[0]
        if self.noise_level != 0.0 and n:
            noise = self._noise(n, newest_only=True).to(x.device)
            fn = _native.lib().cm_am_modulate_run_noise
            return self._launch(lambda plan, a, b, *rest: fn(plan, a, noise.data_ptr(), b, *rest), x, None, (n, self.width),
                                torch.float32, was_numpy, n, int(frame), int(first_line), int(k0))
        return self._launch(_native.lib().cm_am_modulate_run, x, None, (n, self.width), torch.float32, was_numpy,
                            n, int(frame), int(first_line), int(k0))
