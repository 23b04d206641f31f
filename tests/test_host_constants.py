# -*- coding: utf-8 -*-
"""Host-side logic (no GPU): the scipy-based design code and the line/carrier bookkeeping of
color_modem_amd against the constants the reference derives (tests/golden/plans.json)."""
import numpy
import pytest

import stacks
from color_modem_amd import line, plan
from color_modem_amd.color import ntsc, pal, secam

PLANS = stacks.plans()


def check_filter(f, ref, name):
    assert f.shift == ref['shift'], name
    numpy.testing.assert_allclose(f.b, [float(v) for v in ref['b']], rtol=0, atol=1e-14, err_msg=name)
    numpy.testing.assert_allclose(f.a, [float(v) for v in ref['a']], rtol=0, atol=1e-14, err_msg=name)
    assert abs(f.phase_shift - float(ref['phase_shift'])) < 1e-12, name
    # the cascade form handed to the device is the same transfer function
    b, a = numpy.array([1.0]), numpy.array([1.0])
    for row in f.sos():
        b, a = numpy.convolve(b, row[:3]), numpy.convolve(a, row[3:])
    n = max(len(f.b), len(f.a))
    numpy.testing.assert_allclose(numpy.trim_zeros(b, 'b'), f.b, rtol=0, atol=1e-12, err_msg=name + ' sos')
    numpy.testing.assert_allclose(numpy.trim_zeros(a, 'b'), f.a, rtol=0, atol=1e-12, err_msg=name + ' sos')
    assert n <= 9


def check_qam(backend, ref, height):
    assert repr(float(backend.line_config.fs)) == ref['fs']
    assert abs(backend.qam.carrier_phase_step - float(ref['carrier_phase_step'])) < 1e-15
    assert abs(backend.line_shift - float(ref['line_shift'])) < 1e-13
    assert abs(backend.frame_shift - float(ref['frame_shift'])) < 1e-13
    assert backend.frame_cycle == ref['frame_cycle']
    for key, attr in (('precorrect', '_chroma_precorrect_lowpass'), ('extract2x', '_extract_chroma2x'),
                      ('remove2x', '_remove_chroma2x'), ('demod_lp', '_demod_lowpass')):
        check_filter(getattr(backend.qam, attr), ref[key], key)
    for f, row in enumerate(ref['start_phase']):
        for y, v in enumerate(row):
            assert abs(backend.start_phase(f, y) - float(v)) < 1e-12, (f, y)
            assert backend.line_config.is_alternate_line(f, y) == ref['alt'][f][y]
    assert [backend.line_config.analog_line(y) for y in range(len(ref['analog_line']))] == ref['analog_line']


def test_pal_plan_constants():
    m = pal.PalDModem(line.LineConfig((720, 576)))
    ref = PLANS['pal_720x576']
    check_qam(m.backend, ref, 576)
    check_filter(m._filter, ref['pald_lp'], 'pald_lp')
    assert abs(m._sin_factor - float(ref['sin_factor'])) < 1e-15
    assert abs(m._cos_factor - float(ref['cos_factor'])) < 1e-15


def test_ntsc_plan_constants():
    """NTSC-M needs the pre-validation iirdesign behaviour; goldens were made with the same shim."""
    m = ntsc.NtscCombModem(line.LineConfig((720, 480)))
    ref = PLANS['ntsc_720x480']
    check_qam(m.backend, ref, 480)
    assert abs(m._factor - float(ref['comb_factor'])) < 1e-15


def test_secam_plan_constants():
    m = secam.SecamModem(line.LineConfig((720, 576)))
    ref = PLANS['secam_720x576']
    for key, attr in (('fsc_dr', '_fsc_dr'), ('fsc_db', '_fsc_db'), ('fdev_dr', '_fdev_dr'), ('fdev_db', '_fdev_db'),
                      ('flimit_min', '_flimit_min'), ('flimit_max', '_flimit_max'), ('bell_f0', '_bell_f0')):
        assert abs(getattr(m, attr) - float(ref[key])) < 1e-15, key
    for key, f in (('precorrect_lp', m._chroma_precorrect_lowpass), ('lf_precorrect', m._chroma_precorrect),
                   ('lf_reverse', m._reverse_chroma_precorrect), ('bell', m._chroma_demod_bell),
                   ('chroma_bp', m._chroma_demod_chroma_filter), ('luma_bs', m._chroma_demod_luma_filter),
                   ('fm_lp', m._chroma_demod._lowpass)):
        check_filter(f, ref[key], key)
    assert abs(m._chroma_demod._fc - float(ref['fm_fc'])) < 1e-15
    for f, row in enumerate(ref['start_phase_inverted']):
        for y, v in enumerate(row):
            assert m._start_phase_inverted(f, y) == v


def test_resample_fir_matches_scipy_and_oracle():
    import ctypes
    from oracle import cm_oracle
    h_ref = numpy.array([float(v) for v in PLANS['resample_fir']])
    numpy.testing.assert_allclose(plan.resample_fir(), h_ref, rtol=0, atol=2e-16)  # color_modem_amd/design.py's own Kaiser / I0 series vs the reference's scipy: 1 ulp of the centre tap
    h = numpy.empty(41)
    cm_oracle.lib().orc_firwin41(h.ctypes.data_as(ctypes.POINTER(ctypes.c_double)))
    numpy.testing.assert_allclose(h, h_ref, rtol=0, atol=1e-15)  # own Kaiser/I0 evaluation vs scipy: 1-2 ulp


def test_oracle_primitives_match_scipy():
    import ctypes
    import scipy.signal
    from oracle import cm_oracle
    dp = ctypes.POINTER(ctypes.c_double)
    rng = numpy.random.default_rng(5)
    for n in (1, 2, 7, 40, 720, 737):
        x = rng.standard_normal(n)
        up = numpy.empty(2 * n)
        cm_oracle.lib().orc_resample_up2(x.ctypes.data_as(dp), n, up.ctypes.data_as(dp))
        numpy.testing.assert_allclose(up, scipy.signal.resample_poly(x, 2, 1), rtol=0, atol=1e-14)
        dn = numpy.empty((n + 1) // 2)
        k = cm_oracle.lib().orc_resample_dn2(x.ctypes.data_as(dp), n, dn.ctypes.data_as(dp))
        assert k == len(dn)
        numpy.testing.assert_allclose(dn, scipy.signal.resample_poly(x, 1, 2), rtol=0, atol=1e-14)


def test_line_standard_detect_and_errors():
    assert line.LineStandard.detect(480) is line.LineStandard.NTSC_525
    assert line.LineStandard.detect(576) is line.LineStandard.GERBER_625
    assert line.LineStandard.detect(376) is line.LineStandard.BAIRD_405
    assert line.LineStandard.detect(405) is line.LineStandard.NTSC_525
    assert line.LineStandard.detect(738) is line.LineStandard.FRENCH_819
    assert line.LineStandard.detect(760) is line.LineStandard.BELGIAN_819
    with pytest.raises(IndexError):
        line.LineStandard.detect(2000)
    lc = line.LineConfig((720, 576))
    assert lc.fs == 13.5e6
    assert [lc.analog_line(y) for y in range(4)] == [23, 336, 24, 337]


def test_lane_tables_shapes_and_regimes():
    for name, size in (('pal_d', (720, 576)), ('pal_3d', (720, 576)), ('ntsc_comb_3d', (720, 480)), ('pal_s', (720, 16))):
        bp = plan.build_plan(stacks.make(name, size))
        d = bp.desc
        assert d.demod_main.n_lines >= size[1] + 2 * d.demodulation_delay
        assert d.demod_main.frame_cycle in (2, 4)
        tab = numpy.ctypeslib.as_array(d.demod_main.table, shape=(d.demod_main.frame_cycle, 3, d.demod_main.n_lines,
                                                                   plan.CM_LANE_DOUBLES))
        assert numpy.all(numpy.isfinite(tab))
        # detector phase is a unit vector wherever it is defined
        r = numpy.hypot(tab[..., 0], tab[..., 1])
        assert numpy.all((numpy.abs(r - 1.0) < 1e-12) | (r == 0.0))
    assert plan.build_plan(stacks.make('pal_d', (720, 576))).desc.first_is_plain == 1
    assert plan.build_plan(stacks.make('pal_s', (720, 576))).desc.main_luma_bandstop == 1


def test_unsupported_options_fail_loudly():
    from color_modem_amd import comb
    lc = line.LineConfig((720, 576))
    with pytest.raises(NotImplementedError):       # Q = 1: the notch's group delay at DC rounds to 1 sample
        plan.build_plan(pal.PalDModem(lc, notch=1.0))
    with pytest.raises(NotImplementedError):       # arbitrary averaging callables cannot be compiled into the plan
        plan.build_plan(comb.SimpleCombModem(ntsc.NtscCombModem(line.LineConfig((720, 480))), avg=max))
    with pytest.raises(NotImplementedError):
        plan.build_plan(pal.Pal3DModem(lc, avg=lambda a, b: a))
    with pytest.raises(NotImplementedError):
        plan.build_plan(comb.SimpleCombModem(pal.PalDModem(lc)))
    # built options
    assert plan.build_plan(pal.PalDModem(lc, notch=2.0)).desc.notch.n_sections == 1
    d = plan.build_plan(comb.Simple3DCombModem(ntsc.NtscCombModem(line.LineConfig((720, 480))), avg=comb.minavg)).desc
    assert d.chroma_average == plan.CM_AVG_MIN


def _turned(entry, c, s, pairs):
    out = entry.copy()
    for ia, ib in pairs:      # (a, b) -> (a c - b s, a s + b c)
        a, b = entry[ia], entry[ib]
        out[ia], out[ib] = a * c - b * s, a * s + b * c
    return out


@pytest.mark.parametrize('stack,size', [('pal_d', (720, 576)), ('pal_3d', (720, 576)), ('ntsc_comb_3d', (720, 480)),
                                        ('pal_s', (720, 576)), ('pal_3d_minavg', (720, 576))])
def test_frame_rotation_reproduces_per_frame_tables(stack, size, monkeypatch):
    """Long sub-carrier cycles: table(F) == table(F % 2) with every phase advanced by frame_rotation[F] - checked
    on the short-cycle systems, where both layouts can be built (cm_plan_desc::frame_rotation)."""
    exact = plan.QamTables(stacks.make(stack, size)._stack())
    monkeypatch.setattr(plan, 'MAX_TABLE_CYCLE', 1)
    turning = plan.QamTables(stacks.make(stack, size)._stack())
    assert turning.rotating and turning.table_frames == 2 and not exact.rotating
    assert turning.rotation_cycle == exact.cycle or exact.cycle == 1
    rot = turning.frame_rotation()
    # demodulator: [0],[1] = sin, cos(theta): the pair (cos, sin) turns; likewise [2],[3]; coefficient pairs [4..15]
    demod_pairs = [(1, 0), (3, 2)] + [(4 + 2 * j, 5 + 2 * j) for j in range(6)] + [(20 + 2 * j, 21 + 2 * j) for j in range(6)]
    want, bits_w = exact.demod_main_table()
    have, bits_h = turning.demod_main_table()
    assert bits_w == bits_h and have.shape[0] == 2
    for F in range(exact.cycle):
        c, s = rot[F % turning.rotation_cycle]
        for k in range(3):
            for ln in range(0, want.shape[2], 7):
                got = _turned(have[F % 2, k, ln], c, s, demod_pairs)
                numpy.testing.assert_allclose(got, want[F, k, ln], rtol=0, atol=1e-12, err_msg=str((F, k, ln)))
    if exact.first_is_plain:
        want, have = exact.demod_first_table(), turning.demod_first_table()
        for F in range(exact.cycle):
            c, s = rot[F]
            for ln in range(0, want.shape[2], 5):
                got = _turned(have[F % 2, 0, ln], c, s, [(1, 0), (4, 5), (10, 11)])
                numpy.testing.assert_allclose(got, want[F, 0, ln], rtol=0, atol=1e-12)
    want, have = exact.mod_table(), turning.mod_table()
    for F in range(exact.cycle):
        c, s = rot[F]
        for ln in range(0, want.shape[2], 5):
            got = _turned(have[F % 2, 1, ln], c, s, [(1, 0)])
            numpy.testing.assert_allclose(got, want[F, 1, ln], rtol=0, atol=1e-12)


def test_long_cycle_plans_build():
    """4.43 MHz colour on 525 lines repeats every 4800 frames (utils.py:78-80); PAL-M on 625 lines every 286."""
    for modem, cyc in ((ntsc.NtscCombModem(line.LineConfig((720, 480)), ntsc.NtscVariant.NTSC443), 4800),
                       (pal.PalDModem(line.LineConfig((720, 480))), 4800),
                       (pal.PalSModem(line.LineConfig((720, 576)), pal.PalVariant.PAL_M), 286)):
        bp = plan.build_plan(modem)
        assert bp.desc.frame_rotation_cycle == cyc
        assert bp.desc.demod_main.frame_cycle == 2 and bp.desc.mod_main.frame_cycle == 2


def test_generated_wide_shapes_header_is_current():
    """csrc/cm_shapes_wide.h is what tools/gen_wide_shapes.py writes from the package's own filter designs (round 6): a design change
    that moves a section count or a shift parity at one of its widths must regenerate the header, or those widths fall back to the
    run-time shape silently."""
    import importlib.util
    import os
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    spec = importlib.util.spec_from_file_location('gen_wide_shapes', os.path.join(root, 'tools', 'gen_wide_shapes.py'))
    gen = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(gen)
    with open(gen.HEADER) as fh:
        assert fh.read() == gen.render(), 'run python tools/gen_wide_shapes.py'


def test_recorded_hbm_traffic_belongs_to_these_sources():
    """profiles/traffic.json (the PMC traffic bench.py reports as roofline.traffic) names the sources it was measured on; bench.py drops the
    figure for any other tree.  A source change without a new tools/profile_bench.sh run would therefore ship a bench line without traffic:
    this test says so here, before the GPU box does."""
    import json
    import os
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    sys.path.insert(0, root)
    import bench
    with open(os.path.join(root, 'profiles', 'traffic.json')) as fh:
        tj = json.load(fh)
    assert tj['src_sha16'] == bench.sources_sha16(), 'csrc/ or include/ changed since the traffic was measured: run tools/profile_bench.sh and copy traffic.json'
    assert tj['frames'] == 1000 and tj['hbm_bytes_per_launch'] >= tj['algorithmic_bytes_per_launch']
