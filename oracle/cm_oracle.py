# -*- coding: utf-8 -*-
"""ctypes binding of oracle/libcm_oracle.so - TEST INFRASTRUCTURE (see oracle/cm_oracle.h).

Builds an ``orc_desc_t`` for a color_modem_amd modem object - variant constants and line geometry from the
object, every filter designed here with scipy, call for call as the reference designs it
(oracle/cm_oracle_design.py; the product designs with its own code, color_modem_amd/design.py) - and
exposes the oracle's per-row, per-frame and batch entry points.  May be imported only by tests/, __graft_entry__.smoke() and the
cpu_baseline leg of bench.py.
"""

import ctypes
import os
import subprocess

import numpy

HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.path.join(HERE, 'libcm_oracle.so')

ORC_MAX_COEF = 12
ORC_MAX_FILTERS = 8
KIND = {'pal_s': 1, 'pal_d': 2, 'pal_3d': 3, 'ntsc': 4, 'ntsc_comb': 5, 'secam': 6}
WRAP = {None: 0, 'simple': 1, 'simple_3d': 2, 'color_averaging': 3}


class Filter(ctypes.Structure):
    _fields_ = [('nb', ctypes.c_int32), ('na', ctypes.c_int32), ('shift', ctypes.c_int32),
                ('present', ctypes.c_int32), ('b', ctypes.c_double * ORC_MAX_COEF),
                ('a', ctypes.c_double * ORC_MAX_COEF), ('phase_shift', ctypes.c_double)]


class Desc(ctypes.Structure):
    _fields_ = [('kind', ctypes.c_int32), ('wrapper', ctypes.c_int32), ('use_minavg', ctypes.c_int32),
                ('alternate_phases', ctypes.c_int32),
                ('frame_rate', ctypes.c_double), ('total_lines', ctypes.c_int32),
                ('odd_first', ctypes.c_int32), ('odd_last', ctypes.c_int32),
                ('even_first', ctypes.c_int32), ('even_last', ctypes.c_int32),
                ('width', ctypes.c_int32), ('height', ctypes.c_int32),
                ('total_width_factor', ctypes.c_double),
                ('fsc', ctypes.c_double), ('frame_cycle', ctypes.c_int32), ('pal3d_disable', ctypes.c_int32),
                ('carrier_phase_step', ctypes.c_double),
                ('fsc_dr', ctypes.c_double), ('fsc_db', ctypes.c_double), ('fdev_dr', ctypes.c_double),
                ('fdev_db', ctypes.c_double), ('flimit_min', ctypes.c_double), ('flimit_max', ctypes.c_double),
                ('bell_f0', ctypes.c_double), ('m0', ctypes.c_double), ('bell_kn', ctypes.c_double),
                ('bell_kd', ctypes.c_double), ('fm_fc', ctypes.c_double),
                ('filters', Filter * ORC_MAX_FILTERS)]


def build_library(force=False):
    if force or not os.path.exists(LIB_PATH) or \
            os.path.getmtime(LIB_PATH) < os.path.getmtime(os.path.join(HERE, 'cm_oracle.cpp')):
        subprocess.check_call(['make', '-C', HERE, 'libcm_oracle.so'], stdout=subprocess.DEVNULL)
    return LIB_PATH


_lib = None


def lib():
    global _lib
    if _lib is None:
        build_library()
        L = ctypes.CDLL(LIB_PATH)
        dp = ctypes.POINTER(ctypes.c_double)
        fp = ctypes.POINTER(ctypes.c_float)
        u8 = ctypes.POINTER(ctypes.c_uint8)
        L.orc_create.restype = ctypes.c_void_p
        L.orc_create.argtypes = [ctypes.POINTER(Desc)]
        L.orc_destroy.argtypes = [ctypes.c_void_p]
        L.orc_last_error.restype = ctypes.c_char_p
        L.orc_modulation_delay.argtypes = [ctypes.c_void_p]
        L.orc_demodulation_delay.argtypes = [ctypes.c_void_p]
        L.orc_modulate.argtypes = [ctypes.c_void_p, ctypes.c_int, ctypes.c_int, dp, dp, dp, ctypes.c_int, dp]
        L.orc_demodulate.argtypes = [ctypes.c_void_p, ctypes.c_int, ctypes.c_int, dp, ctypes.c_int, dp, dp, dp]
        L.orc_modulate_components.argtypes = [ctypes.c_void_p, ctypes.c_int, ctypes.c_int, dp, dp, dp, ctypes.c_int, dp]
        L.orc_demodulate_components.argtypes = [ctypes.c_void_p, ctypes.c_int, ctypes.c_int, dp, ctypes.c_int,
                                                ctypes.c_int, dp, dp, dp]
        L.orc_modulate_frame.argtypes = [ctypes.c_void_p, ctypes.c_int, dp, dp]
        L.orc_demodulate_frame.argtypes = [ctypes.c_void_p, ctypes.c_int, dp, dp]
        L.orc_image_modulate.argtypes = [ctypes.c_void_p, ctypes.c_int, u8, u8]
        L.orc_image_demodulate.argtypes = [ctypes.c_void_p, ctypes.c_int, u8, u8]
        L.orc_demodulate_frames_f32.argtypes = [ctypes.POINTER(Desc), fp, fp, ctypes.c_int64, ctypes.c_int64,
                                                ctypes.c_int]
        L.orc_modulate_frames_f32.argtypes = [ctypes.POINTER(Desc), fp, fp, ctypes.c_int64, ctypes.c_int64,
                                              ctypes.c_int]
        L.orc_firwin41.argtypes = [dp]
        L.orc_resample_up2.argtypes = [dp, ctypes.c_int, dp]
        L.orc_resample_dn2.argtypes = [dp, ctypes.c_int, dp]
        L.orc_resample_dn2.restype = ctypes.c_int
        L.orc_filter_apply.argtypes = [ctypes.POINTER(Filter), dp, ctypes.c_int, dp]
        L.orc_start_phase.restype = ctypes.c_double
        L.orc_start_phase.argtypes = [ctypes.POINTER(Desc), ctypes.c_int, ctypes.c_int]
        L.orc_analog_line.argtypes = [ctypes.POINTER(Desc), ctypes.c_int]
        L.orc_is_alternate_line.argtypes = [ctypes.POINTER(Desc), ctypes.c_int, ctypes.c_int]
        _lib = L
    return _lib


def _dp(a):
    return a.ctypes.data_as(ctypes.POINTER(ctypes.c_double))


def make_filter(f):
    out = Filter()
    if f is None:
        out.present = 0
        return out
    b, a = numpy.atleast_1d(f.b), numpy.atleast_1d(f.a)
    out.nb, out.na, out.shift, out.present = len(b), len(a), int(f.shift), 1
    out.b[:len(b)] = list(b)
    out.a[:len(a)] = list(a)
    out.phase_shift = float(f.phase_shift)
    return out


def make_desc(modem):
    """orc_desc_t for a color_modem_amd modem object."""
    stack = modem._stack()
    d = Desc()
    d.kind = KIND[stack['kind']]
    wrapper = stack.get('demod_wrapper') or stack.get('mod_wrapper')
    if stack.get('demod_wrapper') and stack.get('mod_wrapper'):
        raise NotImplementedError('the oracle descriptor carries one wrapper')
    d.wrapper = WRAP[wrapper]
    backend = stack['backend']
    lc = backend.line_config if hasattr(backend, 'line_config') else backend._line_config
    std = lc.line_standard
    d.frame_rate, d.total_lines = std.frame_rate, std.total_lines
    d.odd_first, d.odd_last = std.odd_field_first_active_line, std.odd_field_last_active_line
    d.even_first, d.even_last = std.even_field_first_active_line, std.even_field_last_active_line
    d.width, d.height = lc.size
    d.total_width_factor = std.total_width_factor
    # Every filter is designed HERE with scipy, call for call as the reference designs it (oracle/cm_oracle_design.py) - the product's own
    # design code (color_modem_amd/design.py) does not reach the oracle; from the modem objects come the variant presets and the geometry.
    from oracle import cm_oracle_design as design
    fs = lc.fs
    if stack['kind'] == 'secam':
        m = backend
        v = m._variant
        d.alternate_phases = 1 if m._alternate_phases else 0
        d.fsc_dr, d.fsc_db = 2.0 * v.fsc_dr / fs, 2.0 * v.fsc_db / fs                      # secam.py:156-159
        d.fdev_dr, d.fdev_db = 2.0 * v.fdev_dr / fs, 2.0 * v.fdev_db / fs
        d.flimit_min = 2.0 * (v.bell_f0 + v.flimit_minbell) / fs                            # secam.py:160-162
        d.flimit_max = 2.0 * (v.bell_f0 + v.flimit_maxbell) / fs
        d.bell_f0 = 2.0 * v.bell_f0 / fs
        d.m0, d.bell_kn, d.bell_kd = v.m0, v.bell_kn, v.bell_kd
        fl, d.fm_fc = design.secam_filters(fs, v)
        d.frame_cycle = 1
    else:
        d.fsc = backend.config.fsc
        d.frame_cycle = backend.frame_cycle
        comb = stack.get('comb')
        pre, extract2x, remove2x, demod_lp, d.carrier_phase_step = design.qam_filters(fs, backend.config)
        pald = design.pald_filter(backend.config.fsc, d.carrier_phase_step) if stack['kind'] in ('pal_d', 'pal_3d') else None
        notches = [design.notch(backend.config.fsc, fs, f.q) if f is not None else None
                   for f in (stack.get('comb_notch'), stack.get('wrapper_notch'))]
        fl = [pre, extract2x, remove2x, demod_lp, pald] + notches
        from color_modem_amd import comb as comb_module
        if stack['kind'] == 'pal_3d' and comb._avg is comb_module.minavg:
            d.use_minavg |= 1
        if stack['kind'] == 'pal_3d':
            d.pal3d_disable = (0 if comb._use_sin else 1) | (0 if comb._use_cos else 2)
        if stack.get('wrapper_avg') is comb_module.minavg:
            d.use_minavg |= 2
    for i, f in enumerate(fl):
        d.filters[i] = make_filter(f)
    return d


def _custom_wrapper_avg(modem):
    """the avg= callable of a SimpleCombModem / Simple3DCombModem when it is neither comb.avg nor comb.minavg, else None"""
    from color_modem_amd import comb as comb_module
    stack = modem._stack() if hasattr(modem, '_stack') else {}
    fn = stack.get('wrapper_avg') if stack.get('demod_wrapper') else None
    return fn if (fn is not None and fn is not comb_module.avg and fn is not comb_module.minavg) else None


def _custom_pal3d_avg(modem):
    """the avg= callable of a bare Pal3DModem that uses both of its estimates, when it is neither comb.avg nor comb.minavg, else None"""
    from color_modem_amd import comb as comb_module
    stack = modem._stack() if hasattr(modem, '_stack') else {}
    if stack.get('kind') != 'pal_3d' or stack.get('demod_wrapper') or stack.get('mod_wrapper'):
        return None
    c3 = stack['comb']
    fn = c3._avg if (c3._use_sin and c3._use_cos) else None
    return fn if (fn is not None and fn is not comb_module.avg and fn is not comb_module.minavg) else None


class OraclePal3DCallable(object):
    """Pal3DModem(avg=f) with a function of the caller's own (pal.py:144-148, 176-179): the C++ oracle knows comb.avg and comb.minavg only, so
    this is pal.py:180-234 restated in Python around C++ oracle objects.  The two estimates the reference hands to f (pal.py:209-211) are what
    Pal3DModem(use_cos=False) and Pal3DModem(use_sin=False) return for (u, v) - up to the V-switch sign, which the reference applies AFTER f
    (pal.py:219-220) and the single-estimate decoders before they return: it is taken off and put back.  Test infrastructure."""

    def __init__(self, modem):
        import copy
        stack = modem._stack()
        c3 = stack['comb']
        self._avg = _custom_pal3d_avg(modem)
        self._lc = stack['backend'].line_config
        only_sin, only_cos = copy.copy(c3), copy.copy(c3)
        only_sin._use_cos = False
        only_cos._use_sin = False
        for m in (only_sin, only_cos):
            m._avg = None
            m.notch = None
        self._a, self._b = OracleModem(only_sin), OracleModem(only_cos)
        self._backend = OracleModem(stack['backend'])
        self._notch = c3.notch                                                       # comb.py:29-31
        self.modulation_delay, self.demodulation_delay = 0, 1
        self.width, self.height = self._a.width, self._a.height
        self._last_frame = self._last_line = -1
        self._k = -1

    def demodulate_components(self, frame, line, composite, strip_chroma=True):
        composite = numpy.asarray(composite, dtype=numpy.float64)
        self._k = self._k + 1 if (frame == self._last_frame and line == self._last_line + 2) else 0   # pal.py:191
        self._last_frame, self._last_line = frame, line
        ya, ua, va = self._a.demodulate_components(frame, line, composite, False)
        yb, ub, vb = self._b.demodulate_components(frame, line, composite, False)
        if self._k == 0:                                                              # pal.py:191-195: returned before the strip
            return ya, ua, va
        if self._k == 1:                                                              # pal.py:199-201: the first line's decode again
            y, u, v = ya, ua, va
        else:
            s = -1.0 if self._lc.is_alternate_line(frame, line - 2) else 1.0          # pal.py:219-220
            u = numpy.asarray(self._avg(ua, ub), dtype=numpy.float64)                 # pal.py:210
            v = s * numpy.asarray(self._avg(s * va, s * vb), dtype=numpy.float64)     # pal.py:211, 219-220
            y = ya                                                                    # pal.py:223: the previous composite row
        if strip_chroma:                                                              # pal.py:225-228
            y = y - self._backend.modulate_components(frame, line - 2, numpy.zeros(len(composite)), u, v)
            if self._notch is not None:
                y = OracleCallableComb._apply_notch(self, y)
        return y, u, v

    def demodulate(self, frame, line, composite):
        return OracleCallableComb._decode_pal(*self.demodulate_components(frame, line, composite))

    def modulate(self, frame, line, r, g, b):
        return self._backend.modulate(frame, line, r, g, b)

    def modulate_components(self, frame, line, y, u, v):
        return self._backend.modulate_components(frame, line, y, u, v)

    def demodulate_frame(self, frame, composite):
        return OracleCallableComb.demodulate_frame(self, frame, composite)


class OracleCallableComb(object):
    """SimpleCombModem / Simple3DCombModem with an avg= callable of the caller's own: the C++ oracle knows comb.avg and comb.minavg only, so
    this is comb.py:71-127 restated in Python around the C++ oracle objects of the wrapped decoder and of the backend modulator (float64, one
    numpy row per call, the callable applied exactly as comb.py:103-104 applies it).  Test infrastructure, like the rest of oracle/."""

    # decode_components of the two QAM families (pal.py:41-46, ntsc.py:36-41), as the reference writes them
    @staticmethod
    def _decode_pal(y, u, v):
        return y + 1.140250855188141 * v, y - 0.5808092090310976 * v - 0.3939307027516405 * u, y + 2.028397565922921 * u

    @staticmethod
    def _decode_ntsc(y, u, v):
        return (0.9999999999999998 * y + 1.133735501874552 * v + 0.007249535771601484 * u,
                y - 0.5766784873222262 * v - 0.3834753199055935 * u,
                y + 0.001087790524980047 * v + 2.037050709207452 * u)

    def __init__(self, modem):
        stack = modem._stack()
        self._avg = _custom_wrapper_avg(modem)
        self._own_delay = 1 if stack['demod_wrapper'] == 'simple_3d' else 0        # comb.py:74, 126
        inner = stack.get('comb') or stack['backend']
        self._inner = OracleModem(inner)
        self._backend = OracleModem(stack['backend'])
        self._decode = self._decode_pal if stack['kind'] in ('pal_s', 'pal_d', 'pal_3d') else self._decode_ntsc
        self._notch = stack.get('wrapper_notch')                                     # comb.py:86-88
        self.modulation_delay = self._inner.modulation_delay                        # comb.py:75
        self.demodulation_delay = self._inner.demodulation_delay + self._own_delay  # comb.py:76
        self.width, self.height = self._inner.width, self._inner.height
        self._last_frame = self._last_line = -1
        self._last = None

    def _apply_notch(self, y):
        import scipy.signal
        f = self._notch                                                              # utils.py:28-36
        if f.shift == 0:
            return scipy.signal.lfilter(f.b, f.a, y)
        assert f.shift > 0
        return scipy.signal.lfilter(f.b, f.a, numpy.concatenate((y, y[-1] * numpy.ones(f.shift))))[f.shift:]

    def demodulate_components(self, frame, line, composite, strip_chroma=True):
        composite = numpy.asarray(composite, dtype=numpy.float64)
        curr = self._inner.demodulate_components(frame, line, composite, False)      # comb.py:98 / 101
        if frame != self._last_frame or line != self._last_line + 2:                  # comb.py:97
            y, u, v = curr
        else:
            y = self._last[0] if self._own_delay else curr[0]                         # comb.py:102
            u = numpy.asarray(self._avg(self._last[1], curr[1]), dtype=numpy.float64)  # comb.py:103
            v = numpy.asarray(self._avg(self._last[2], curr[2]), dtype=numpy.float64)  # comb.py:104
            if strip_chroma:                                                          # comb.py:105-110
                y = y - self._backend.modulate_components(frame, line - 2 * (self._own_delay - self.modulation_delay),
                                                          numpy.zeros(len(composite)), u, v)
                if self._notch is not None:
                    y = self._apply_notch(y)
        self._last_frame, self._last_line, self._last = frame, line, curr
        return y, u, v

    def demodulate(self, frame, line, composite):
        return self._decode(*self.demodulate_components(frame, line, composite))      # comb.py:121-122

    def modulate(self, frame, line, r, g, b):
        return self._backend.modulate(frame, line, r, g, b)                           # comb.py:93-94

    def modulate_components(self, frame, line, y, u, v):
        return self._backend.modulate_components(frame, line, y, u, v)

    def demodulate_frame(self, frame, composite):
        """image.py:75-83: both fields, the delay calls in front, the bottom rows fed again."""
        comp = numpy.asarray(composite, dtype=numpy.float64)
        height, width = comp.shape
        delay = self.demodulation_delay
        out = numpy.zeros((3, height, width))
        for field in range(2):
            for y in range(field, 2 * delay, 2):
                self.demodulate(frame, y, comp[y])
            for y in range(field, height, 2):
                iy = y + 2 * delay
                while iy >= height:
                    iy -= 2
                out[0, y], out[1, y], out[2, y] = self.demodulate(frame, y + 2 * delay, comp[iy])
        return out


class OracleModem(object):
    """Stateful oracle object with the reference's per-row protocol."""

    def __new__(cls, modem):
        if cls is OracleModem and _custom_wrapper_avg(modem) is not None:
            return OracleCallableComb(modem)
        if cls is OracleModem and _custom_pal3d_avg(modem) is not None:
            return OraclePal3DCallable(modem)
        return object.__new__(cls)

    def __init__(self, modem):
        self.desc = make_desc(modem)
        self._h = lib().orc_create(ctypes.byref(self.desc))
        if not self._h:
            raise RuntimeError(lib().orc_last_error().decode())
        self.width, self.height = self.desc.width, self.desc.height
        self.modulation_delay = lib().orc_modulation_delay(self._h)
        self.demodulation_delay = lib().orc_demodulation_delay(self._h)

    def __del__(self):
        if getattr(self, '_h', None):
            lib().orc_destroy(self._h)
            self._h = None

    def demodulate(self, frame, line, composite):
        x = numpy.ascontiguousarray(composite, dtype=numpy.float64)
        n = len(x)
        r, g, b = numpy.empty(n), numpy.empty(n), numpy.empty(n)
        lib().orc_demodulate(self._h, frame, line, _dp(x), n, _dp(r), _dp(g), _dp(b))
        return r, g, b

    def modulate(self, frame, line, r, g, b):
        r, g, b = [numpy.ascontiguousarray(v, dtype=numpy.float64) for v in (r, g, b)]
        out = numpy.empty(len(r))
        lib().orc_modulate(self._h, frame, line, _dp(r), _dp(g), _dp(b), len(r), _dp(out))
        return out

    def modulate_components(self, frame, line, y, u, v):
        y, u, v = [numpy.ascontiguousarray(c, dtype=numpy.float64) for c in (y, u, v)]
        out = numpy.empty(len(y))
        if lib().orc_modulate_components(self._h, frame, line, _dp(y), _dp(u), _dp(v), len(y), _dp(out)):
            raise AttributeError('this stack has no modulate_components')
        return out

    def demodulate_components(self, frame, line, composite, strip_chroma=True):
        x = numpy.ascontiguousarray(composite, dtype=numpy.float64)
        n = len(x)
        y, u, v = numpy.empty(n), numpy.empty(n), numpy.empty(n)
        if lib().orc_demodulate_components(self._h, frame, line, _dp(x), n, 1 if strip_chroma else 0, _dp(y), _dp(u),
                                           _dp(v)):
            raise AttributeError('this stack has no demodulate_components')
        return y, u, v

    def demodulate_frame(self, frame, composite):
        x = numpy.ascontiguousarray(composite, dtype=numpy.float64)
        assert x.shape == (self.height, self.width)
        out = numpy.empty((3, self.height, self.width))
        lib().orc_demodulate_frame(self._h, frame, _dp(x), _dp(out))
        return out

    def modulate_frame(self, frame, rgb):
        x = numpy.ascontiguousarray(rgb, dtype=numpy.float64)
        assert x.shape == (3, self.height, self.width)
        out = numpy.empty((self.height, self.width))
        lib().orc_modulate_frame(self._h, frame, _dp(x), _dp(out))
        return out

    def image_modulate(self, frame, rgb8):
        x = numpy.ascontiguousarray(rgb8, dtype=numpy.uint8)
        assert x.shape == (self.height, self.width, 3)
        out = numpy.empty((self.height, self.width), dtype=numpy.uint8)
        u8 = ctypes.POINTER(ctypes.c_uint8)
        lib().orc_image_modulate(self._h, frame, x.ctypes.data_as(u8), out.ctypes.data_as(u8))
        return out

    def image_demodulate(self, frame, comp8):
        x = numpy.ascontiguousarray(comp8, dtype=numpy.uint8)
        assert x.shape == (self.height, self.width)
        out = numpy.empty((self.height, self.width, 3), dtype=numpy.uint8)
        u8 = ctypes.POINTER(ctypes.c_uint8)
        lib().orc_image_demodulate(self._h, frame, x.ctypes.data_as(u8), out.ctypes.data_as(u8))
        return out


def demodulate_frames_f32(modem, composite, first_frame=0, n_threads=1):
    if _custom_wrapper_avg(modem) is not None or _custom_pal3d_avg(modem) is not None:
        orc = OracleCallableComb(modem) if _custom_wrapper_avg(modem) is not None else OraclePal3DCallable(modem)
        return numpy.stack([orc.demodulate_frame(first_frame + i, f) for i, f in enumerate(numpy.asarray(composite))]).astype(numpy.float32)
    desc = make_desc(modem)
    x = numpy.ascontiguousarray(composite, dtype=numpy.float32)
    n, h, w = x.shape
    assert (h, w) == (desc.height, desc.width)
    out = numpy.empty((n, 3, h, w), dtype=numpy.float32)
    fp = ctypes.POINTER(ctypes.c_float)
    rc = lib().orc_demodulate_frames_f32(ctypes.byref(desc), x.ctypes.data_as(fp), out.ctypes.data_as(fp), n,
                                         first_frame, n_threads)
    if rc:
        raise RuntimeError(lib().orc_last_error().decode())
    return out


def modulate_frames_f32(modem, rgb, first_frame=0, n_threads=1):
    desc = make_desc(modem)
    x = numpy.ascontiguousarray(rgb, dtype=numpy.float32)
    n, _, h, w = x.shape
    assert (h, w) == (desc.height, desc.width)
    out = numpy.empty((n, h, w), dtype=numpy.float32)
    fp = ctypes.POINTER(ctypes.c_float)
    rc = lib().orc_modulate_frames_f32(ctypes.byref(desc), x.ctypes.data_as(fp), out.ctypes.data_as(fp), n,
                                       first_frame, n_threads)
    if rc:
        raise RuntimeError(lib().orc_last_error().decode())
    return out
