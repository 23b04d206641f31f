# -*- coding: utf-8 -*-
"""Host sanitizer run of the PRODUCT's plan construction (GPU sanitizers are not available on the pool).

csrc/cm_api.hip compiles under -DCM_HOST_DRY_RUN -fsanitize=address,undefined (host code; no kernel instance is referenced) into tests/sim/libcolor_modem_host_asan.so: the
plan constructors (descriptor validation, kernel-instance selection, every table builder and coefficient conversion of cm_plan.h /
cm_am_plan.h, the scan kernels' constants, the MAC tap tables) then run against host memory on a box without a GPU.  A child process
(tests/host_sanitize_child.py) preloads the ASan runtime, points the Python layer at that library and builds the plans of every stack x variant x width of
tools/support_matrix.py and tests/fuzz_parity.py - the wrapped, fused, two-level, notched, callable and nested engines with all the
plans they hold - describes and destroys them, then feeds malformed descriptors; it must exit cleanly with no sanitizer report."""
import glob
import os
import subprocess
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
LIB = os.path.join(ROOT, 'tests', 'sim', 'libcolor_modem_host_asan.so')
SRC = os.path.join(ROOT, 'color_modem_amd', 'csrc')
HIPCC = os.environ.get('HIPCC', '/opt/rocm/bin/hipcc')

def _asan_runtime():
    hits = sorted(glob.glob('/opt/rocm/lib/llvm/lib/clang/*/lib/linux/libclang_rt.asan-x86_64.so'))
    return hits[-1] if hits else None


def _build():
    sources = glob.glob(os.path.join(SRC, '*')) + [os.path.join(ROOT, 'include', 'color_modem_hip.h')]
    if os.path.exists(LIB) and all(os.path.getmtime(s) <= os.path.getmtime(LIB) for s in sources):
        return
    subprocess.check_call([HIPCC, '--offload-arch=gfx950', '-O1', '-g', '-std=c++17', '-fPIC', '-shared', '-w',
                           '-fsanitize=address,undefined', '-fno-sanitize=vptr,function', '-fno-omit-frame-pointer', '-shared-libasan',
                           '-DCM_HOST_DRY_RUN', '-o', LIB, os.path.join(SRC, 'cm_api.hip')])


def test_plan_construction_under_asan_ubsan():
    asan = _asan_runtime()
    if asan is None or not os.path.exists(HIPCC):
        pytest.skip('no hipcc / clang ASan runtime in this image')
    _build()
    env = dict(os.environ, LD_PRELOAD=asan, ASAN_OPTIONS='detect_leaks=0:abort_on_error=0:exitcode=97:verify_asan_link_order=0',
               UBSAN_OPTIONS='halt_on_error=1:print_stacktrace=1:exitcode=98')
    proc = subprocess.run([sys.executable, os.path.join(ROOT, 'tests', 'host_sanitize_child.py'), LIB], env=env, stdout=subprocess.PIPE,
                          stderr=subprocess.STDOUT, universal_newlines=True, timeout=900)
    assert proc.returncode == 0, proc.stdout[-6000:]
    assert 'sanitized host run complete' in proc.stdout
    assert 'ERROR: AddressSanitizer' not in proc.stdout and 'runtime error:' not in proc.stdout, proc.stdout[-6000:]
