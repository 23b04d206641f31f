# -*- coding: utf-8 -*-
"""The streaming stages of color_modem_amd/csrc/cm_stages.h, compiled for the host (tests/sim), against
the oracle: float64 proves the stream schedule / edge handling / coefficient tables, float32 bounds the
rounding error the device kernels can have.  No GPU needed."""
import ctypes
import os

import numpy
import pytest

import stacks
from color_modem_amd import plan, testing

SIM = os.path.join(os.path.dirname(os.path.abspath(__file__)), 'sim', 'libcm_sim.so')


@pytest.fixture(scope='module')
def sim():
    import __graft_entry__
    __graft_entry__.build()
    L = ctypes.CDLL(SIM)
    dp = ctypes.POINTER(ctypes.c_double)
    for fn in (L.cm_sim_demodulate_run_f64, L.cm_sim_demodulate_run_f32):
        fn.argtypes = [ctypes.POINTER(plan.PlanDesc), dp, dp] + [ctypes.c_int] * 5
    L.cm_sim_last_error.restype = ctypes.c_char_p
    return L


def run(L, bp, comp, frame, first_line, k0, f32, mid=1):
    n, w = comp.shape
    comp = numpy.ascontiguousarray(comp, dtype=numpy.float64)
    out = numpy.zeros((n, 3, w))
    dp = ctypes.POINTER(ctypes.c_double)
    fn = L.cm_sim_demodulate_run_f32 if f32 else L.cm_sim_demodulate_run_f64
    rc = fn(ctypes.byref(bp.desc), comp.ctypes.data_as(dp), out.ctypes.data_as(dp), n, frame, first_line, k0, mid)
    assert rc == 0, L.cm_sim_last_error()
    return out


@pytest.mark.parametrize('stack,size', [('pal_d', (720, 576)), ('pal_s', (720, 576)), ('pal_3d', (720, 576)),
                                        ('ntsc', (720, 480)), ('ntsc_comb', (720, 480)),
                                        ('ntsc_comb_simple', (720, 480)), ('ntsc_comb_3d', (720, 480)),
                                        ('pal_d', (704, 8)), ('ntsc_a', (720, 480)), ('ntsc_comb_3d_a', (720, 480))])
@pytest.mark.parametrize('frame,first_line', [(0, 0), (1, 1), (3, 2)])
def test_streaming_matches_oracle(sim, stack, size, frame, first_line):
    from oracle import cm_oracle
    modem = stacks.make(stack, size, explicit=size[1] < 400)
    bp = plan.build_plan(modem)
    n = 5
    comp = testing.synthetic_composite(1, n, size[0], seed=17 + frame)[0]
    orc = cm_oracle.OracleModem(modem)
    ref = numpy.stack([numpy.stack(orc.demodulate(frame, first_line + 2 * i, comp[i].astype(numpy.float64)))
                       for i in range(n)])
    o64 = run(sim, bp, comp, frame, first_line, 0, f32=False)
    o64_edge = run(sim, bp, comp, frame, first_line, 0, f32=False, mid=0)
    o32 = run(sim, bp, comp, frame, first_line, 0, f32=True)
    assert numpy.array_equal(o64, o64_edge), 'edge-free body differs from the guarded body'
    for i in range(n):
        assert stacks.rel_err(o64[i], ref[i]) < 1e-11, (stack, i)
        assert stacks.rel_err(o32[i], ref[i]) < 2e-6, (stack, i)


@pytest.mark.parametrize('stack,size', [('pal_d', (768, 576)), ('pal_d', (640, 576)), ('pal_3d', (1920, 576)), ('pal_s', (1024, 576)),
                                        ('ntsc', (768, 480)), ('ntsc_comb', (704, 480)), ('ntsc_comb_3d', (1440, 480))])
def test_run_time_shape_matches_oracle(sim, stack, size):
    """Other sampling rates: cascades padded with identity sections, shift parities and pre-correction shift read at run time
    (cm_stages.h: Sys<..., RT>); the PAL-D front end with odd shifts is exercised here."""
    from oracle import cm_oracle
    modem = stacks.make(stack, size, explicit=False)
    bp = plan.build_plan(modem)
    n, frame, first_line = 4, 1, 1
    comp = testing.synthetic_composite(1, n, size[0], seed=23)[0]
    orc = cm_oracle.OracleModem(modem)
    ref = numpy.stack([numpy.stack(orc.demodulate(frame, first_line + 2 * i, comp[i].astype(numpy.float64)))
                       for i in range(n)])
    o64 = run(sim, bp, comp, frame, first_line, 0, f32=False)
    o32 = run(sim, bp, comp, frame, first_line, 0, f32=True)
    for i in range(n):
        assert stacks.rel_err(o64[i], ref[i]) < 1e-11, (stack, i)
        assert stacks.rel_err(o32[i], ref[i]) < 3e-6, (stack, i)


@pytest.fixture(scope='module')
def sim_secam(sim):
    dp = ctypes.POINTER(ctypes.c_double)
    for fn in (sim.cm_sim_secam_demodulate_run_f64, sim.cm_sim_secam_demodulate_run_f32,
               sim.cm_sim_secam_modulate_run_f64, sim.cm_sim_secam_modulate_run_f32):
        fn.argtypes = [ctypes.POINTER(plan.PlanDesc), dp, dp] + [ctypes.c_int] * 4
    return sim


@pytest.mark.parametrize('frame,first_line', [(0, 0), (1, 1), (5, 4)])
def test_secam_streaming_matches_oracle(sim_secam, frame, first_line):
    """SECAM encoder and decoder stages (float64: schedule; float32 (+ float64 phase path): device rounding)."""
    from oracle import cm_oracle
    dp = ctypes.POINTER(ctypes.c_double)
    n = 4
    for stack in ('secam', 'secam_avg'):
        modem = stacks.make(stack, (720, 576), explicit=False)
        bp = plan.build_plan(modem)
        rgb = testing.synthetic_rgb(1, n, 720, seed=5 + frame)[0].astype(numpy.float64)
        orc = cm_oracle.OracleModem(modem)
        ref = numpy.stack([orc.modulate(frame, first_line + 2 * i, rgb[0, i], rgb[1, i], rgb[2, i]) for i in range(n)])
        rows = numpy.ascontiguousarray(rgb.transpose(1, 0, 2))
        for fn, tol in ((sim_secam.cm_sim_secam_modulate_run_f64, 1e-11), (sim_secam.cm_sim_secam_modulate_run_f32, 2e-6)):
            out = numpy.zeros((n, 720))
            assert fn(ctypes.byref(bp.desc), rows.ctypes.data_as(dp), out.ctypes.data_as(dp), n, frame, first_line, 0) == 0
            for i in range(n):
                assert stacks.rel_err(out[i], ref[i]) < tol, (stack, i)
    modem = stacks.make('secam', (720, 576), explicit=False)
    bp = plan.build_plan(modem)
    comp = numpy.ascontiguousarray(ref.astype(numpy.float32).astype(numpy.float64))  # a valid SECAM signal (averaged variant)
    orc = cm_oracle.OracleModem(modem)
    want = numpy.stack([numpy.stack(orc.demodulate(frame, first_line + 2 * i, comp[i])) for i in range(n)])
    for fn, tol in ((sim_secam.cm_sim_secam_demodulate_run_f64, 1e-11), (sim_secam.cm_sim_secam_demodulate_run_f32, 8e-6)):
        out = numpy.zeros((n, 3, 720))
        assert fn(ctypes.byref(bp.desc), comp.ctypes.data_as(dp), out.ctypes.data_as(dp), n, frame, first_line, 0) == 0
        for i in range(n):
            assert stacks.rel_err(out[i], want[i]) < tol, i


def test_secam_margin_frame_row_ends_in_float64(sim_secam):
    """tests/golden/secam_iii_640_margin.npz (SECAM III, 640x76): all-float32 stages sat at 1.04e-5 in one start-of-row sample
    (column 11).  With the band-pass + bell of the guarded bodies in float64 (cm_stages.h: SecamBp64) the float32 stage code
    stays below 1e-6 on the whole frame - the bound the GPU test of the same frame holds the kernels to is 2e-6."""
    import os
    from color_modem_amd import line
    from color_modem_amd.color import secam
    dp = ctypes.POINTER(ctypes.c_double)
    g = numpy.load(os.path.join(stacks.GOLDEN, 'secam_iii_640_margin.npz'))
    size = [int(x) for x in g['size']]
    lc = line.LineConfig((size[0], size[1]), line.LineStandard.detect(size[2]))
    modem = secam.SecamModem(lc, getattr(secam.SecamVariant, str(g['vname'])))
    bp = plan.build_plan(modem)
    comp, first = g['comp'][0].astype(numpy.float64), int(g['first'])
    for parity in (0, 1):
        rows = numpy.ascontiguousarray(comp[parity::2])
        n, w = rows.shape
        o64, o32 = numpy.zeros((n, 3, w)), numpy.zeros((n, 3, w))
        assert sim_secam.cm_sim_secam_demodulate_run_f64(ctypes.byref(bp.desc), rows.ctypes.data_as(dp), o64.ctypes.data_as(dp), n, first, parity, 0) == 0
        assert sim_secam.cm_sim_secam_demodulate_run_f32(ctypes.byref(bp.desc), rows.ctypes.data_as(dp), o32.ctypes.data_as(dp), n, first, parity, 0) == 0
        assert numpy.abs(o32 - o64).max() / numpy.abs(o64).max() < 1e-6
