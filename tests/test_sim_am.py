# -*- coding: utf-8 -*-
"""The streaming stages of csrc/cm_am_stages.h (Proto-SECAM, NIIR) compiled for the host (tests/sim/cm_sim_am.cpp) against
the numpy oracle: float64 checks the schedule - resampler phases, FilterFunction shifts at the 3x rate, row edges - to
1e-11, float32 predicts the rounding error of the device kernels.  CPU only."""
import ctypes
import os
import subprocess

import numpy
import pytest

import am_stacks
from color_modem_amd import comb, line, plan_am, testing
from color_modem_amd.color import niir, protosecam
from oracle import cm_oracle_am as oa

HERE = os.path.dirname(os.path.abspath(__file__))
SIM = os.path.join(HERE, 'sim')


def _lib():
    so, src = os.path.join(SIM, 'libcm_sim_am.so'), os.path.join(SIM, 'cm_sim_am.cpp')
    deps = [src] + [os.path.join(HERE, '..', 'color_modem_amd', 'csrc', f) for f in ('cm_am_stages.h', 'cm_am_plan.h', 'cm_stages.h', 'cm_plan.h')]
    if not os.path.exists(so) or any(os.path.getmtime(p) > os.path.getmtime(so) for p in deps):
        subprocess.check_call(['g++', '-O2', '-std=c++17', '-fPIC', '-shared', '-w', '-o', so, src])
    L = ctypes.CDLL(so)
    dp = ctypes.POINTER(ctypes.c_double)
    for fn in (L.am_sim_demod_run, L.am_sim_mod_run):
        fn.argtypes = [ctypes.POINTER(plan_am.AmDesc), ctypes.c_int, dp, dp, ctypes.c_int, ctypes.c_longlong, ctypes.c_int, ctypes.c_int]
    L.am_sim_last_error.restype = ctypes.c_char_p
    return L


def _run(fn, desc, use_float, inp, out_shape, frame, first_line, k0):
    inp = numpy.ascontiguousarray(inp, dtype=numpy.float64)
    out = numpy.zeros(out_shape)
    dp = ctypes.POINTER(ctypes.c_double)
    rc = fn(ctypes.byref(desc), use_float, inp.ctypes.data_as(dp), out.ctypes.data_as(dp), inp.shape[0], frame, first_line, k0)
    assert rc == 0, _lib().am_sim_last_error()
    return out


CASES = [('proto', (720, 10), 'FRENCH_819'), ('proto_nofilter', (720, 6), 'BELGIAN_819'), ('proto', (1024, 6), 'GERBER_625'),
         ('proto', (480, 6), 'FRENCH_819'), ('proto_avg', (720, 8), 'FRENCH_819')]


@pytest.mark.parametrize('stack,size,std', CASES)
@pytest.mark.parametrize('use_float', [0, 1])
def test_proto_runs_against_oracle(stack, size, std, use_float):
    L = _lib()
    lc = line.LineConfig(size, getattr(line.LineStandard, std))
    modem = am_stacks.STACKS[stack](lc)
    inner = modem.backend if stack == 'proto_avg' else modem
    desc = plan_am.build_am_desc(modem)
    W, H = size
    n = H // 2
    rgb = testing.synthetic_rgb(1, H, W, seed=17)[0].astype(numpy.float64)
    tol = 3e-6 if use_float else 1e-11
    for frame, field in ((3, 0), (4, 1)):
        lines = list(range(field, H, 2))
        # ---- encoder: the oracle object row by row (with the wrapper of comb.py:141-152 when averaging)
        orc = oa.make(inner)
        rows = numpy.stack([rgb[:, y] for y in lines])                      # [n][3][W]
        want = []
        state = {'y': None}
        for i, y in enumerate(lines):
            r, g, b = rows[i]
            if stack == 'proto_avg':
                yy, u, v = inner.encode_components(r, g, b)
                if i == 0:
                    state = {'y': yy, 'u': u, 'v': v}
                py, pu, pv = state['y'], state['u'], state['v']
                state = {'y': yy, 'u': u, 'v': v}
                want.append(orc.modulate_components(frame, y - 2, py, 0.5 * (u + pu), 0.5 * (v + pv)))
            else:
                want.append(orc.modulate(frame, y, r, g, b))
        want = numpy.stack(want)
        got = _run(L.am_sim_mod_run, desc, use_float, rows, (len(lines), W), frame, field, 0)
        assert numpy.abs(got - want).max() < tol * max(1.0, numpy.abs(want).max()), (stack, 'mod', frame)
        # ---- decoder on the oracle's composite
        dec = oa.make(inner)
        back = numpy.stack([numpy.stack(dec.demodulate(frame, y, want[i])) for i, y in enumerate(lines)])
        got = _run(L.am_sim_demod_run, desc, use_float, want, (len(lines), 3, W), frame, field, 0)
        assert numpy.abs(got - back).max() < tol * max(1.0, numpy.abs(back).max()), (stack, 'demod', frame)


NIIR_CASES = [('niir', (720, 10), 'GERBER_625'), ('niir_hue', (720, 8), 'GERBER_625'), ('niir', (768, 6), 'NTSC_525'),
              ('niir', (1024, 6), 'GERBER_625'), ('niir', (640, 6), 'GERBER_625')]


@pytest.mark.parametrize('stack,size,std', NIIR_CASES)
@pytest.mark.parametrize('use_float', [0, 1])
def test_niir_runs_against_oracle(stack, size, std, use_float):
    L = _lib()
    lc = line.LineConfig(size, getattr(line.LineStandard, std))
    modem = am_stacks.STACKS[stack](lc)
    desc = plan_am.build_am_desc(modem)
    comp_desc = plan_am.build_am_desc(modem, components=True)
    W, H = size
    rgb = testing.synthetic_rgb(1, H, W, seed=23)[0].astype(numpy.float64)
    tol = 2e-5 if use_float else 1e-10
    for frame, field in ((1, 0), (4798, 1)):
        lines = list(range(field, H, 2))
        rows = numpy.stack([rgb[:, y] for y in lines])
        orc = oa.make(modem)
        delay = 1 if stack == 'niir_hue' else 0
        want = numpy.stack([orc.modulate(frame, y + 2 * delay, *rows[i]) for i, y in enumerate(lines)])
        got = _run(L.am_sim_mod_run, desc, use_float, rows, (len(lines), W), frame, field + 2 * delay, 0)
        assert numpy.abs(got - want).max() < tol * max(1.0, numpy.abs(want).max()), (stack, 'mod', frame)
        dec = oa.make(modem)
        back = numpy.stack([numpy.stack(dec.demodulate(frame, y, want[i])) for i, y in enumerate(lines)])
        got = _run(L.am_sim_demod_run, desc, use_float, want, (len(lines), 3, W), frame, field, 0)
        # the hue is the angle of a decimated product pair: where that pair is small (saturation near zero) single float32
        # samples sit an order of magnitude above the rest - bound the bulk tightly and the isolated samples loosely
        err = numpy.abs(got - back) / max(1.0, numpy.abs(back).max())
        assert numpy.quantile(err, 0.999) < (1e-5 if use_float else 1e-10), (stack, 'demod', frame)
        assert err.max() < (1e-4 if use_float else 1e-10), (stack, 'demod', frame, err.max())
        # the component protocol with the chroma left in the luma
        dec = oa.make(modem)
        back = numpy.stack([numpy.stack(dec.demodulate_components(frame, y, want[i], strip_chroma=False)) for i, y in enumerate(lines)])
        got = _run(L.am_sim_demod_run, comp_desc, use_float | 2, want, (len(lines), 3, W), frame, field, 0)
        err = numpy.abs(got - back) / max(1.0, numpy.abs(back).max())
        assert numpy.quantile(err, 0.999) < (1e-5 if use_float else 1e-10) and err.max() < (1e-4 if use_float else 1e-10), (stack, 'components', frame)


@pytest.mark.parametrize('stack,size,std', NIIR_CASES)
def test_niir_decoder_precision_split(stack, size, std):
    """What the device decoders compute since round 4 (cm_am_stages.h: NiirHue; the simulator's mode 4): the hue path - interpolator,
    band-pass, low-pass, M / S, the hue products and their two decimators - in float64, saturation / re-modulation / niir_finish in float32.
    EVERY sample inside 2e-6 of full scale on pictures where the all-float32 build (mode 1, above) needs its quantile."""
    L = _lib()
    lc = line.LineConfig(size, getattr(line.LineStandard, std))
    modem = am_stacks.STACKS[stack](lc)
    desc = plan_am.build_am_desc(modem)
    comp_desc = plan_am.build_am_desc(modem, components=True)
    W, H = size
    rgb = testing.synthetic_rgb(1, H, W, seed=23)[0].astype(numpy.float64)
    for frame, field in ((1, 0), (4798, 1)):
        lines = list(range(field, H, 2))
        rows = numpy.stack([rgb[:, y] for y in lines])
        orc = oa.make(modem)
        delay = 1 if stack == 'niir_hue' else 0
        comp = numpy.stack([orc.modulate(frame, y + 2 * delay, *rows[i]) for i, y in enumerate(lines)])
        dec = oa.make(modem)
        back = numpy.stack([numpy.stack(dec.demodulate(frame, y, comp[i])) for i, y in enumerate(lines)])
        got = _run(L.am_sim_demod_run, desc, 4, comp, (len(lines), 3, W), frame, field, 0)
        assert (numpy.abs(got - back) / max(1.0, numpy.abs(back).max())).max() < 2e-6, (stack, 'demod', frame)
        dec = oa.make(modem)
        back = numpy.stack([numpy.stack(dec.demodulate_components(frame, y, comp[i], strip_chroma=False)) for i, y in enumerate(lines)])
        got = _run(L.am_sim_demod_run, comp_desc, 4 | 2, comp, (len(lines), 3, W), frame, field, 0)
        assert (numpy.abs(got - back) / max(1.0, numpy.abs(back).max())).max() < 2e-6, (stack, 'components', frame)


@pytest.mark.parametrize('stack', ['niir', 'niir_hue'])
@pytest.mark.parametrize('use_float', [0, 1])
def test_niir_encoder_on_grey_pictures(stack, use_float):
    """Grey and nearly grey pixels (tests/golden/am_mod_niir*_grey.npz, made by the reference): the pedestal's hue is the angle of the
    rounding residues of niir.py:35-36 - the stage code forms them in float64 in the reference's own operation order wherever the
    saturation is small (cm_am_stages.h: niir_chroma_f64), so the float32 build holds 1e-5 there too."""
    L = _lib()
    z = am_stacks.load('am_mod_%s_grey' % stack)
    modem = am_stacks.make(stack, z)
    desc = plan_am.build_am_desc(modem)
    W, H = [int(v) for v in z['size']]
    delay = 1 if stack == 'niir_hue' else 0
    for i, frame in enumerate(int(f) for f in z['frames']):
        rgb = z['inp'][i].astype(numpy.float64)
        for field in (0, 1):
            lines = list(range(field, H, 2))
            # the row schedule of image.py:47-55: `delay` warm-up calls, then rows clamped into the picture
            calls = [y for y in range(field, 2 * delay, 2)] + [y + 2 * delay for y in lines]
            src = [min(y, H - 1 - ((H - 1 - y) % 2)) if y < H else y - 2 * ((y - H) // 2 + 1) for y in calls]
            rows = numpy.stack([rgb[:, y] for y in src])
            got = _run(L.am_sim_mod_run, desc, use_float, rows, (len(calls), W), frame, calls[0], 0)[len(calls) - len(lines):]
            want = z['out'][i][lines]
            assert numpy.abs(got - want).max() < (1e-5 if use_float else 1e-11), (stack, frame, field, numpy.abs(got - want).max())
