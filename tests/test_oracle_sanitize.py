# -*- coding: utf-8 -*-
"""Host sanitizer run of the CPU oracle (ASan / UBSan on the CPU build; GPU sanitizers are not
available on the pool).  oracle/Makefile builds libcm_oracle_asan.so (-fsanitize=address,undefined); a child process
preloads libasan, drives every entry-point family of the oracle on small inputs - encoders, decoders with one and two
lines of history, SECAM, the threaded batch calls, odd heights, the bottom-edge re-feed - and must exit cleanly with no
sanitizer report."""
import os
import subprocess
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
ORACLE = os.path.join(ROOT, 'oracle')

CHILD = r'''
import ctypes, os, sys
sys.path.insert(0, %(root)r); sys.path.insert(0, os.path.join(%(root)r, 'tests'))
import numpy
from oracle import cm_oracle
cm_oracle.LIB_PATH = %(lib)r
cm_oracle.build_library = lambda force=False: cm_oracle.LIB_PATH
import stacks
from color_modem_amd import testing
for stack, size in (('pal_d', (720, 7)), ('pal_3d', (704, 8)), ('ntsc_comb_3d', (720, 5)), ('ntsc', (640, 4)),
                    ('secam', (720, 6)), ('secam_avg', (720, 5)), ('pal_avg', (720, 3)), ('pal_3d_minavg', (720, 6)),
                    ('pal_d_notch', (720, 4))):
    modem = stacks.make(stack, size)
    rgb = testing.synthetic_rgb(3, size[1], size[0], seed=5)
    comp = cm_oracle.modulate_frames_f32(modem, rgb, first_frame=2, n_threads=3)
    back = cm_oracle.demodulate_frames_f32(modem, comp, first_frame=2, n_threads=2)
    assert numpy.all(numpy.isfinite(comp)) and numpy.all(numpy.isfinite(back)), stack
    orc = cm_oracle.OracleModem(modem)
    for f, y in ((0, 0), (0, 2), (0, 4), (1, 1), (1, 3), (1, 9)):
        orc.demodulate(f, y, comp[0, y %% size[1]].astype(numpy.float64))
        orc.modulate(f, y, *[rgb[0, c, y %% size[1]].astype(numpy.float64) for c in range(3)])
print('sanitized run complete')
'''


def _libasan():
    try:
        path = subprocess.check_output(['gcc', '-print-file-name=libasan.so'], universal_newlines=True).strip()
    except (OSError, subprocess.CalledProcessError):
        return None
    return path if os.path.isabs(path) and os.path.exists(path) else None


def test_oracle_under_asan_ubsan():
    asan = _libasan()
    if asan is None:
        pytest.skip('no libasan for this gcc')
    subprocess.check_call(['make', '-s', '-C', ORACLE, 'libcm_oracle_asan.so'])
    lib = os.path.join(ORACLE, 'libcm_oracle_asan.so')
    env = dict(os.environ, LD_PRELOAD=asan, ASAN_OPTIONS='detect_leaks=0:abort_on_error=0:exitcode=97',
               UBSAN_OPTIONS='halt_on_error=1:print_stacktrace=1:exitcode=98')
    proc = subprocess.run([sys.executable, '-c', CHILD % {'root': ROOT, 'lib': lib}], env=env, stdout=subprocess.PIPE,
                          stderr=subprocess.STDOUT, universal_newlines=True, timeout=900)
    assert proc.returncode == 0, proc.stdout[-4000:]
    assert 'sanitized run complete' in proc.stdout
    assert 'ERROR: AddressSanitizer' not in proc.stdout and 'runtime error:' not in proc.stdout, proc.stdout[-4000:]
