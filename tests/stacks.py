# -*- coding: utf-8 -*-
"""Modem stacks named as in tests/golden/make_golden.py, built from color_modem_amd classes."""
import json
import os

import numpy

from color_modem_amd import comb, line
from color_modem_amd.color import ntsc, pal, secam

GOLDEN = os.path.join(os.path.dirname(os.path.abspath(__file__)), 'golden')

STACKS = {
    'pal_s': lambda lc: pal.PalSModem(lc),
    'pal_d': lambda lc: pal.PalDModem(lc),
    'pal_3d': lambda lc: pal.Pal3DModem(lc),
    'ntsc': lambda lc: ntsc.NtscModem(lc),
    'ntsc_comb': lambda lc: ntsc.NtscCombModem(lc),
    'ntsc_comb_simple': lambda lc: comb.SimpleCombModem(ntsc.NtscCombModem(lc)),
    'ntsc_comb_3d': lambda lc: comb.Simple3DCombModem(ntsc.NtscCombModem(lc)),
    'pal_avg': lambda lc: comb.ColorAveragingModem(pal.PalSModem(lc)),
    'ntsc_avg': lambda lc: comb.ColorAveragingModem(ntsc.NtscModem(lc)),
    'secam': lambda lc: secam.SecamModem(lc),
    'secam_avg': lambda lc: comb.ColorAveragingModem(secam.SecamModem(lc)),
}
# options and variants (tests/golden/make_golden.py: option_cases)
STACKS.update({
    'pal_d_notch': lambda lc: pal.PalDModem(lc, notch=5.0),
    'pal_3d_notch': lambda lc: pal.Pal3DModem(lc, notch=3.0),
    'ntsc_comb_3d_notch': lambda lc: comb.Simple3DCombModem(ntsc.NtscCombModem(lc, notch=2.5), notch=8.0),
    'pal_3d_minavg': lambda lc: pal.Pal3DModem(lc, avg=comb.minavg),
    'pal_3d_sin': lambda lc: pal.Pal3DModem(lc, use_cos=False),
    'pal_3d_cos': lambda lc: pal.Pal3DModem(lc, use_sin=False, notch=4.0),
    'ntsc_simple_minavg': lambda lc: comb.SimpleCombModem(ntsc.NtscModem(lc), avg=comb.minavg),
    'ntsc_comb_3d_minavg': lambda lc: comb.Simple3DCombModem(ntsc.NtscCombModem(lc), avg=comb.minavg),
    'secam_i': lambda lc: secam.SecamModem(lc, secam.SecamVariant.SECAM_I),
    'secam_ii': lambda lc: secam.SecamModem(lc, secam.SecamVariant.SECAM_II),
    'pal_d_palm': lambda lc: pal.PalDModem(lc, pal.PalVariant.PAL_M),
    'pal_s_palm': lambda lc: pal.PalSModem(lc, pal.PalVariant.PAL_M),
    'ntsc_comb_443': lambda lc: ntsc.NtscCombModem(lc, ntsc.NtscVariant.NTSC443),
    'ntsc_443': lambda lc: ntsc.NtscModem(lc, ntsc.NtscVariant.NTSC443),
    'ntsc_a': lambda lc: ntsc.NtscModem(lc, ntsc.NtscVariant.NTSC_A),
    'ntsc_comb_a': lambda lc: ntsc.NtscCombModem(lc, ntsc.NtscVariant.NTSC_A),
    'ntsc_comb_3d_a': lambda lc: comb.Simple3DCombModem(ntsc.NtscCombModem(lc, ntsc.NtscVariant.NTSC_A)),
    'pal_d_60': lambda lc: pal.PalDModem(lc),
    'pal_s_60': lambda lc: pal.PalSModem(lc),
})
# variant families pinned by tests/golden/make_golden.py: variant_cases
STACKS.update({
    'pal_d_paln': lambda lc: pal.PalDModem(lc, pal.PalVariant.PAL_N),
    'pal_s_paln': lambda lc: pal.PalSModem(lc, pal.PalVariant.PAL_N),
    'ntsc_comb_n': lambda lc: ntsc.NtscCombModem(lc, ntsc.NtscVariant.NTSC_N),
    'ntsc_n': lambda lc: ntsc.NtscModem(lc, ntsc.NtscVariant.NTSC_N),
    'ntsc_comb_3d_361': lambda lc: comb.Simple3DCombModem(ntsc.NtscCombModem(lc, ntsc.NtscVariant.NTSC361)),
    'ntsc_361': lambda lc: ntsc.NtscModem(lc, ntsc.NtscVariant.NTSC361),
    'ntsc_comb_i': lambda lc: ntsc.NtscCombModem(lc, ntsc.NtscVariant.NTSC_I),
    'ntsc_i': lambda lc: ntsc.NtscModem(lc, ntsc.NtscVariant.NTSC_I),
    'secam_iii': lambda lc: secam.SecamModem(lc, secam.SecamVariant.SECAM_III),
    'secam_m': lambda lc: secam.SecamModem(lc, secam.SecamVariant.SECAM_M),
    'secam_n': lambda lc: secam.SecamModem(lc, secam.SecamVariant.SECAM_N),
    'secam_a': lambda lc: secam.SecamModem(lc, secam.SecamVariant.SECAM_A),
})
# comb wrappers around the PAL delay-line decoders (a composition of kernels: color_modem_amd/wrapped.py)
STACKS.update({
    'simple3d_pald': lambda lc: comb.Simple3DCombModem(pal.PalDModem(lc)),
    'simple_pald': lambda lc: comb.SimpleCombModem(pal.PalDModem(lc)),
    'simple3d_pal3d': lambda lc: comb.Simple3DCombModem(pal.Pal3DModem(lc)),
    'simple3d_pald_minavg': lambda lc: comb.Simple3DCombModem(pal.PalDModem(lc), avg=comb.minavg),
    'simple3d_pald_notch': lambda lc: comb.Simple3DCombModem(pal.PalDModem(lc), notch=6.0),
    'simple_pal3d_notch': lambda lc: comb.SimpleCombModem(pal.Pal3DModem(lc), notch=3.0, avg=comb.minavg),
    # (no reference vectors of their own: the oracle, pinned on the stacks above, is the check)
    'simple3d_pal3d_minavg2': lambda lc: comb.Simple3DCombModem(pal.Pal3DModem(lc, avg=comb.minavg), avg=comb.minavg, notch=5.0),
    'simple_pal3d_sin': lambda lc: comb.SimpleCombModem(pal.Pal3DModem(lc, use_cos=False)),
})


# avg= callables of the caller's own (comb.py:72, 81-84) - the same two functions in tests/stacks.py and tests/golden/make_golden.py
def weighted_avg(last, curr):
    return 0.25 * last + 0.75 * curr


def damped_avg(last, curr):      # non-linear and continuous (a discontinuous pick turns float32 rounding of its inputs into a different branch)
    return 0.5 * (last + curr) / (1.0 + 4.0 * abs(last - curr))


def numpy_damped_avg(last, curr):     # written against numpy ufuncs, like the reference's own minavg (comb.py:13-15): cannot take device tensors
    return 0.5 * (last + curr) * numpy.exp(-2.0 * numpy.abs(last - curr))


# notch= values whose FilterFunction shift is not 0 (comb.py:18-20 over utils.py:9-26): +1 at q = 1.0, +7 at q = 0.7 (PAL at 13.5 MHz; the values with a negative shift are unstable filters: the reference's own output overflows)
STACKS.update({
    'pal_d_notchq1': lambda lc: pal.PalDModem(lc, notch=1.0),
    'pal_3d_notchq07': lambda lc: pal.Pal3DModem(lc, notch=0.7),
    'ntsc_comb_3d_notchq1': lambda lc: comb.Simple3DCombModem(ntsc.NtscCombModem(lc), notch=1.0),
    'simple_pald_notchq1': lambda lc: comb.SimpleCombModem(pal.PalDModem(lc), notch=1.0),
})
STACKS.update({
    'simple3d_pald_favg': lambda lc: comb.Simple3DCombModem(pal.PalDModem(lc), avg=weighted_avg),
    'simple_pal3d_favg': lambda lc: comb.SimpleCombModem(pal.Pal3DModem(lc), avg=damped_avg, notch=4.0),
    'simple_ntsc_favg': lambda lc: comb.SimpleCombModem(ntsc.NtscModem(lc), avg=damped_avg),
    'simple3d_ntsccomb_favg': lambda lc: comb.Simple3DCombModem(ntsc.NtscCombModem(lc), avg=weighted_avg),
    # Pal3DModem's own average of its two estimates as a callable (pal.py:144-148, 209-211)
    'pal_3d_favg': lambda lc: pal.Pal3DModem(lc, avg=damped_avg, notch=4.0),
    'pal_3d_wavg': lambda lc: pal.Pal3DModem(lc, avg=weighted_avg),
})
STANDARD = {'pal': 'GERBER_625', 'ntsc': 'NTSC_525', 'secam': 'GERBER_625', 'simple3d': 'GERBER_625', 'simple': 'GERBER_625'}
STANDARD_OF = {'pal_d_palm': 'NTSC_525', 'pal_s_palm': 'NTSC_525', 'pal_d_60': 'NTSC_525', 'pal_s_60': 'NTSC_525',
               'ntsc_comb_n': 'GERBER_625', 'ntsc_n': 'GERBER_625', 'ntsc_comb_i': 'GERBER_625', 'ntsc_i': 'GERBER_625',
               'secam_m': 'NTSC_525', 'secam_a': 'BAIRD_405', 'simple_ntsc_favg': 'NTSC_525', 'simple3d_ntsccomb_favg': 'NTSC_525'}


def line_config(stack, size, explicit=True):
    if not explicit:
        return line.LineConfig(tuple(int(v) for v in size))
    std = getattr(line.LineStandard, STANDARD_OF.get(stack, STANDARD[stack.split('_')[0]]))
    return line.LineConfig(tuple(int(v) for v in size), std)


def make(stack, size, explicit=True):
    return STACKS[stack](line_config(stack, size, explicit))


def load(name):
    return numpy.load(os.path.join(GOLDEN, name + '.npz'))


def plans():
    with open(os.path.join(GOLDEN, 'plans.json')) as fh:
        return json.load(fh)


def rel_err(out, ref):
    """max |out - ref| / max |ref| (the tolerance convention of SURVEY.md Appendix C)."""
    out = numpy.asarray(out, dtype=numpy.float64)
    ref = numpy.asarray(ref, dtype=numpy.float64)
    return float(numpy.max(numpy.abs(out - ref)) / numpy.max(numpy.abs(ref)))


def allclose_violations(out, ref, rtol=1e-5, atol=1e-6):
    """Number of samples outside numpy.allclose(out, ref, rtol, atol) - the element-wise criterion SURVEY.md Appendix C
    asks to be reported beside rel_err (pure relative error is ill-defined at zero crossings, hence atol)."""
    out = numpy.asarray(out, dtype=numpy.float64)
    ref = numpy.asarray(ref, dtype=numpy.float64)
    return int(numpy.count_nonzero(numpy.abs(out - ref) > atol + rtol * numpy.abs(ref)))


def parity_report(out, ref):
    """Per plane of [..., 3, H, W] results: (rel_err, allclose violations, samples)."""
    out = numpy.asarray(out)
    ref = numpy.asarray(ref)
    rows = []
    for p in range(out.shape[-3]):
        o, r = out[..., p, :, :], ref[..., p, :, :]
        rows.append((rel_err(o, r), allclose_violations(o, r), int(o.size)))
    return rows


# ---- the kernel family a golden test runs on ('rows': the streaming kernels on whole rows - what bench.py times; 'scan': the row-parallel kernels) ----
MODES = ['rows', 'scan']


def pinned(target, mode):
    """target.set_small_batch(mode) (an engine or a modem); a plan whose shape no scan kernel serves skips 'scan'."""
    import pytest
    try:
        target.set_small_batch(mode)
    except NotImplementedError as e:
        pytest.skip('%s: %s' % (mode, str(e)[:80]))
    return target


def skip_unserved(mode, fn):
    """fn(), skipping the test where a 'scan' pin meets a direction / entry point the scan kernels do not serve"""
    import pytest
    try:
        return fn()
    except NotImplementedError as e:
        if mode == 'scan' and 'scan' in str(e):
            pytest.skip('scan: %s' % str(e)[:80])
        raise


# ---- nested stacks (round 6; tests/golden/make_golden_nested.py names): they run level by level (color_modem_amd/generic.py) ------
def _nested():
    from color_modem_amd.color import niir
    return {
        'simple_avg_pals': ('GERBER_625', lambda lc: comb.SimpleCombModem(comb.ColorAveragingModem(pal.PalSModem(lc)))),
        'simple3d_avg_pald_minavg': ('GERBER_625', lambda lc: comb.Simple3DCombModem(comb.ColorAveragingModem(pal.PalDModem(lc)), avg=comb.minavg)),
        'simple_simple_ntsc': ('NTSC_525', lambda lc: comb.SimpleCombModem(comb.SimpleCombModem(ntsc.NtscModem(lc)))),
        'simple3d_simple_ntsccomb': ('NTSC_525', lambda lc: comb.Simple3DCombModem(comb.SimpleCombModem(ntsc.NtscCombModem(lc)), avg=comb.minavg)),
        'simple3d_pal3d_favg': ('GERBER_625', lambda lc: comb.Simple3DCombModem(pal.Pal3DModem(lc, avg=damped_avg), notch=4.0)),
        'avg_pal3d_favg': ('GERBER_625', lambda lc: comb.ColorAveragingModem(pal.Pal3DModem(lc, avg=weighted_avg))),
        'simple_niir_hue': ('GERBER_625', lambda lc: comb.SimpleCombModem(niir.HueCorrectingNiirModem(lc))),
        'simple3d_niir': ('GERBER_625', lambda lc: comb.Simple3DCombModem(niir.NiirModem(lc), avg=weighted_avg)),
        'avg_avg_secam': ('GERBER_625', lambda lc: comb.ColorAveragingModem(comb.ColorAveragingModem(secam.SecamModem(lc)))),
        'avg_niir': ('GERBER_625', lambda lc: comb.ColorAveragingModem(niir.NiirModem(lc))),
        'avg_avg_pals': ('GERBER_625', lambda lc: comb.ColorAveragingModem(comb.ColorAveragingModem(pal.PalSModem(lc)))),
    }


NESTED = _nested()


def make_nested(name, size):
    std, factory = NESTED[name]
    return factory(line.LineConfig(tuple(int(v) for v in size), getattr(line.LineStandard, std)))
