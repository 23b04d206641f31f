# -*- coding: utf-8 -*-
"""ctypes binding of libcolor_modem_hip.so (include/color_modem_hip.h).

The library is the product: if it is missing or cannot be loaded this module raises - there is
no Python or CPU substitute behind the Modem / ImageModem entry points.
"""

import ctypes
import os

from color_modem_amd import plan

HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.environ.get('CM_LIB') or os.path.join(HERE, 'libcolor_modem_hip.so')   # CM_LIB: A/B builds of the same ABI

CM_OK, CM_ERR_INVALID, CM_ERR_UNSUPPORTED, CM_ERR_NO_DEVICE, CM_ERR_LAUNCH = 0, -1, -2, -3, -4

# every symbol include/color_modem_hip.h declares
SYMBOLS = ('cm_last_error', 'cm_abi_version', 'cm_device_count', 'cm_plan_create', 'cm_plan_destroy',
           'cm_demodulate_frames', 'cm_modulate_frames', 'cm_demodulate_frames_u8', 'cm_modulate_frames_u8', 'cm_demodulate_run',
           'cm_modulate_run',
           'cm_plan_describe', 'cm_plan_set_small_batch', 'cm_set_pointer_check',
           'cm_mac_plan_create', 'cm_mac_plan_destroy',
           'cm_mac_modulate_frames', 'cm_mac_demodulate_frames', 'cm_mac_modulate_frames_u8', 'cm_mac_demodulate_frames_u8',
           'cm_mac_modulate_run', 'cm_mac_demodulate_run',
           'cm_am_plan_create', 'cm_am_plan_destroy', 'cm_am_modulate_frames', 'cm_am_demodulate_frames',
           'cm_am_modulate_run', 'cm_am_demodulate_run', 'cm_am_modulate_frames_noise', 'cm_am_modulate_run_noise',
           'cm_am_modulate_frames_u8', 'cm_am_demodulate_frames_u8', 'cm_am_plan_set_small_batch',
           'cm_comb_wrap_demodulate_frames', 'cm_comb_wrap_demodulate_frames_u8', 'cm_comb_wrap_demodulate_run', 'cm_filter_rows_f64',
           'cm_comb_wrap_demodulate_frames_fused', 'cm_comb_wrap_demodulate_frames_fused_u8',
           'cm_comb_wrap_calls_per_frame', 'cm_comb_wrap_components_frames', 'cm_comb_wrap_finish_frames', 'cm_comb_wrap_components_run',
           'cm_comb_wrap_finish_run', 'cm_notch_luma_f32')

_lib = None


class MacFir(ctypes.Structure):
    """cm_mac_fir (include/color_modem_hip.h)"""
    _fields_ = [('up', ctypes.c_int32), ('down', ctypes.c_int32), ('n_taps', ctypes.c_int32), ('reserved', ctypes.c_int32),
                ('taps', ctypes.POINTER(ctypes.c_double))]


class MacDesc(ctypes.Structure):
    """cm_mac_desc (include/color_modem_hip.h)"""
    _fields_ = [('width', ctypes.c_int32), ('height', ctypes.c_int32), ('line_width', ctypes.c_int32),
                ('line_shift', ctypes.c_int32), ('even_first', ctypes.c_int32), ('odd_first', ctypes.c_int32),
                ('averaging', ctypes.c_int32), ('reserved', ctypes.c_int32),
                ('resample_fir', ctypes.c_double * 41), ('decode_matrix', ctypes.c_double * 9),
                ('encode_matrix', ctypes.c_double * 9),
                ('luma_in', MacFir), ('chroma_in', MacFir), ('line_out', MacFir), ('line_in', MacFir)]


class NativeError(RuntimeError):
    pass


def lib():
    global _lib
    if _lib is not None:
        return _lib
    if not os.path.exists(LIB_PATH):
        raise NativeError('%s not found: build it with `python -c "import __graft_entry__ as g; g.build()"` '
                          '(hipcc --offload-arch=gfx950); color_modem_amd has no CPU path' % LIB_PATH)
    # torch first: it ships its own libamdhip64 and the library must bind to that copy - two HIP runtimes in one process
    # do not see each other's devices (cm_plan_create then reports CM_ERR_NO_DEVICE on a machine that has one)
    import torch  # noqa: F401
    L = ctypes.CDLL(LIB_PATH)
    fp = ctypes.POINTER(ctypes.c_float)
    vp = ctypes.c_void_p
    L.cm_last_error.restype = ctypes.c_char_p
    L.cm_abi_version.restype = ctypes.c_int
    L.cm_device_count.restype = ctypes.c_int
    L.cm_plan_create.argtypes = [ctypes.POINTER(plan.PlanDesc), ctypes.POINTER(vp)]
    L.cm_plan_destroy.argtypes = [vp]
    L.cm_plan_destroy.restype = None
    L.cm_demodulate_frames.argtypes = [vp, vp, vp, ctypes.c_int64, ctypes.c_int64, vp]
    L.cm_modulate_frames.argtypes = [vp, vp, vp, ctypes.c_int64, ctypes.c_int64, vp]
    L.cm_demodulate_frames_u8.argtypes = [vp, vp, vp, ctypes.c_int64, ctypes.c_int64, vp]
    L.cm_modulate_frames_u8.argtypes = [vp, vp, vp, ctypes.c_int64, ctypes.c_int64, vp]
    L.cm_demodulate_run.argtypes = [vp, vp, vp, ctypes.c_int32, ctypes.c_int32, ctypes.c_int32, ctypes.c_int32, vp]
    L.cm_modulate_run.argtypes = [vp, vp, vp, ctypes.c_int32, ctypes.c_int32, ctypes.c_int32, ctypes.c_int32, vp]
    L.cm_plan_describe.argtypes = [vp, ctypes.c_char_p, ctypes.c_int32]
    L.cm_plan_set_small_batch.argtypes = [vp, ctypes.c_int32]
    L.cm_am_plan_set_small_batch.argtypes = [vp, ctypes.c_int32]
    L.cm_set_pointer_check.argtypes = [ctypes.c_int32]
    L.cm_set_pointer_check.restype = None
    L.cm_mac_plan_create.argtypes = [ctypes.POINTER(MacDesc), ctypes.POINTER(vp)]
    L.cm_mac_plan_destroy.argtypes = [vp]
    L.cm_mac_plan_destroy.restype = None
    md = vp
    L.cm_mac_modulate_frames.argtypes = [md, vp, vp, ctypes.c_int64, ctypes.c_int64, vp]
    L.cm_mac_demodulate_frames.argtypes = [md, vp, vp, ctypes.c_int64, ctypes.c_int64, vp]
    L.cm_mac_modulate_frames_u8.argtypes = [md, vp, vp, ctypes.c_int64, ctypes.c_int64, vp]
    L.cm_mac_demodulate_frames_u8.argtypes = [md, vp, vp, ctypes.c_int64, ctypes.c_int64, vp]
    L.cm_mac_modulate_run.argtypes = [md, vp, vp, ctypes.c_int32, ctypes.c_int32, ctypes.c_int32, ctypes.c_int32, vp]
    L.cm_mac_demodulate_run.argtypes = [md, vp, vp, ctypes.c_int32, ctypes.c_int32, ctypes.c_int32, ctypes.c_int32, vp]
    from color_modem_amd import plan_am
    L.cm_am_plan_create.argtypes = [ctypes.POINTER(plan_am.AmDesc), ctypes.POINTER(vp)]
    L.cm_am_plan_destroy.argtypes = [vp]
    L.cm_am_plan_destroy.restype = None
    L.cm_am_modulate_frames.argtypes = [vp, vp, vp, ctypes.c_int64, ctypes.c_int64, vp]
    L.cm_am_demodulate_frames.argtypes = [vp, vp, vp, ctypes.c_int64, ctypes.c_int64, vp]
    L.cm_am_modulate_frames_u8.argtypes = [vp, vp, vp, ctypes.c_int64, ctypes.c_int64, vp]
    L.cm_am_demodulate_frames_u8.argtypes = [vp, vp, vp, ctypes.c_int64, ctypes.c_int64, vp]
    L.cm_am_modulate_run.argtypes = [vp, vp, vp, ctypes.c_int32, ctypes.c_int32, ctypes.c_int32, ctypes.c_int32, vp]
    L.cm_am_demodulate_run.argtypes = [vp, vp, vp, ctypes.c_int32, ctypes.c_int32, ctypes.c_int32, ctypes.c_int32, vp]
    i32 = ctypes.c_int32
    L.cm_am_modulate_frames_noise.argtypes = [vp, vp, vp, vp, ctypes.c_int64, ctypes.c_int64, vp]
    L.cm_am_modulate_run_noise.argtypes = [vp, vp, vp, vp, i32, i32, i32, i32, vp]
    i64 = ctypes.c_int64
    L.cm_comb_wrap_demodulate_frames.argtypes = [vp, vp, vp, vp, vp, vp, i64, i64, vp]
    L.cm_comb_wrap_demodulate_frames_u8.argtypes = [vp, vp, vp, vp, vp, vp, i64, i64, vp]
    L.cm_comb_wrap_demodulate_run.argtypes = [vp, vp, vp, vp, vp, vp, i32, i32, i32, i32, vp]
    L.cm_comb_wrap_demodulate_frames_fused.argtypes = [vp, vp, vp, vp, vp, vp, vp, i64, i64, vp]
    L.cm_comb_wrap_demodulate_frames_fused_u8.argtypes = [vp, vp, vp, vp, vp, vp, vp, i64, i64, vp]
    L.cm_comb_wrap_calls_per_frame.argtypes = [vp, vp]
    L.cm_comb_wrap_components_frames.argtypes = [vp, vp, vp, vp, vp, vp, i64, i64, vp]
    L.cm_comb_wrap_finish_frames.argtypes = [vp, vp, vp, vp, vp, vp, i64, i64, vp]
    L.cm_comb_wrap_components_run.argtypes = [vp, vp, vp, vp, vp, vp, i32, i32, i32, i32, vp]
    L.cm_comb_wrap_finish_run.argtypes = [vp, vp, vp, vp, vp, vp, i32, i32, i32, i32, vp]
    dp = ctypes.POINTER(ctypes.c_double)
    L.cm_filter_rows_f64.argtypes = [dp, i32, dp, i32, i32, vp, vp, i64, i32, vp]
    L.cm_notch_luma_f32.argtypes = [dp, i32, dp, i32, i32, vp, vp, i64, i64, i32, i32, dp, vp]
    if L.cm_abi_version() != plan.CM_ABI_VERSION:
        raise NativeError('libcolor_modem_hip.so ABI %d, Python side expects %d - rebuild the library'
                          % (L.cm_abi_version(), plan.CM_ABI_VERSION))
    _lib = L
    return L


def check(rc):
    if rc == CM_OK:
        return
    msg = lib().cm_last_error().decode('utf-8', 'replace')
    if rc == CM_ERR_INVALID:
        raise ValueError(msg)
    if rc == CM_ERR_UNSUPPORTED:
        raise NotImplementedError(msg)
    raise NativeError('libcolor_modem_hip: %s (code %d)' % (msg, rc))
