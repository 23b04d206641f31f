# -*- coding: utf-8 -*-
"""The C-ABI shared library: builds, loads, exports every symbol of include/color_modem_hip.h and
refuses to compute without a GPU (there is no CPU path behind it)."""
import ctypes
import os
import re

import pytest

import stacks
from color_modem_amd import _native, plan

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_library_builds_and_loads():
    import __graft_entry__
    __graft_entry__.build()
    L = _native.lib()
    assert L.cm_abi_version() == plan.CM_ABI_VERSION


def test_exports_every_declared_symbol():
    header = open(os.path.join(ROOT, 'include', 'color_modem_hip.h')).read()
    declared = set(re.findall(r'\b(cm_[a-z0-9_]+)\s*\(', header))
    declared -= {'cm_status', 'cm_pipeline'}
    assert declared == set(_native.SYMBOLS), declared ^ set(_native.SYMBOLS)
    L = _native.lib()
    for name in declared:
        assert hasattr(L, name), name


def test_struct_layout_matches_header():
    """sizeof checks: the ctypes mirror of cm_plan_desc must match what the C side compiles."""
    import subprocess
    import tempfile
    src = ('#include <stdio.h>\n#include "color_modem_hip.h"\nint main(void){printf("%zu %zu %zu %zu %zu", sizeof(cm_plan_desc), '
           'sizeof(cm_iir_desc), sizeof(cm_lane_table), sizeof(cm_am_desc), sizeof(cm_mac_desc));'
           'printf(" %zu", sizeof(cm_comb_wrap_desc));return 0;}\n')
    with tempfile.TemporaryDirectory() as td:
        c = os.path.join(td, 't.c')
        open(c, 'w').write(src)
        exe = os.path.join(td, 't')
        subprocess.check_call(['gcc', '-I', os.path.join(ROOT, 'include'), c, '-o', exe])
        sizes = [int(v) for v in subprocess.check_output([exe]).split()]
    from color_modem_amd import plan_am, wrapped
    assert sizes == [ctypes.sizeof(plan.PlanDesc), ctypes.sizeof(plan.IirDesc), ctypes.sizeof(plan.LaneTable),
                     ctypes.sizeof(plan_am.AmDesc), ctypes.sizeof(_native.MacDesc), ctypes.sizeof(wrapped.CombWrapDesc)]


def test_no_gpu_means_loud_failure():
    import torch
    if torch.cuda.is_available():
        pytest.skip('a GPU is present')
    L = _native.lib()
    assert L.cm_device_count() == 0
    bp = plan.build_plan(stacks.make('pal_d', (720, 576)))
    handle = ctypes.c_void_p()
    rc = L.cm_plan_create(ctypes.byref(bp.desc), ctypes.byref(handle))
    assert rc == _native.CM_ERR_NO_DEVICE
    assert b'no HIP device' in L.cm_last_error()
    from color_modem_amd import image
    with pytest.raises(_native.NativeError):
        image.ImageModem(stacks.make('pal_d', (720, 576))).demodulate_frames(__import__('numpy').zeros((1, 576, 720), 'f4'))


def test_descriptor_validation_messages():
    L = _native.lib()
    bp = plan.build_plan(stacks.make('pal_d', (720, 576)))
    handle = ctypes.c_void_p()
    bp.desc.abi_version = 99
    assert L.cm_plan_create(ctypes.byref(bp.desc), ctypes.byref(handle)) == _native.CM_ERR_INVALID
    bp.desc.abi_version = plan.CM_ABI_VERSION
    bp.desc.width = 3
    assert L.cm_plan_create(ctypes.byref(bp.desc), ctypes.byref(handle)) == _native.CM_ERR_UNSUPPORTED
    assert b'at least 4' in L.cm_last_error()
