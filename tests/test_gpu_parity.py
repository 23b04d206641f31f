# -*- coding: utf-8 -*-
"""GPU parity: libcolor_modem_hip.so (through the Python API) against the reference-generated
goldens and against the CPU oracle on seeded inputs.  Tolerance: max|out - ref| <= 1e-5 * max|ref|
per frame (BASELINE.json north_star; convention of SURVEY.md Appendix C)."""
import glob
import os

import numpy
import pytest

import stacks
from color_modem_amd import image, testing

pytestmark = pytest.mark.gpu
TOL = 1e-5

DEMOD_FRAMES = sorted(os.path.basename(p)[:-4] for p in glob.glob(os.path.join(stacks.GOLDEN, 'frames_demod_*.npz')))
DEMOD_ROWS = sorted(os.path.basename(p)[:-4] for p in glob.glob(os.path.join(stacks.GOLDEN, 'rows_demod_*.npz')))


def stack_of(name, prefix):
    import re
    return re.sub(r'_w\d+$', '', name[len(prefix):].split('_noise_')[0])     # ..._w768: the same stack at another image width


@pytest.mark.parametrize('name', DEMOD_FRAMES)
def test_frames_demod_golden(name):
    g = stacks.load(name)
    if int(g['size'][0]) % 4:
        pytest.skip('width not a multiple of 4')
    modem = stacks.make(stack_of(name, 'frames_demod_'), g['size'])
    im = image.ImageModem(modem)
    for i, f in enumerate(g['frames']):
        out = im.demodulate_frames(g['inp'][i:i + 1], first_frame=int(f))[0]
        assert out.dtype == numpy.float32
        assert stacks.rel_err(out, g['out'][i]) < TOL, (name, int(f))
    # the same frames as one batch when they are consecutive
    frames = [int(f) for f in g['frames']]
    if frames == list(range(frames[0], frames[0] + len(frames))):
        out = im.demodulate_frames(g['inp'], first_frame=frames[0])
        for i in range(len(frames)):
            assert stacks.rel_err(out[i], g['out'][i]) < TOL


@pytest.mark.parametrize('name', DEMOD_ROWS)
def test_rows_demod_golden(name):
    """The stateful per-row protocol (Modem.demodulate) at full-height line numbers."""
    g = stacks.load(name)
    modem = stacks.make(stack_of(name, 'rows_demod_'), g['size'], explicit=False)
    for i, (f, y) in enumerate(g['seq']):
        out = numpy.stack(modem.demodulate(int(f), int(y), g['inp'][i]))
        assert stacks.rel_err(out, g['out'][i]) < TOL, (name, int(f), int(y))


@pytest.mark.parametrize('stack,size,n_frames,first', [
    ('pal_d', (720, 576), 3, 2), ('pal_d', (720, 575), 2, 5), ('pal_d', (704, 6), 5, 0), ('pal_d', (716, 10), 2, 1), ('pal_d', (688, 5), 3, 2), ('pal_s', (720, 32), 2, 1),
    ('pal_3d', (720, 64), 3, 3), ('ntsc', (720, 480), 1, 1), ('ntsc_comb', (720, 33), 3, 0),
    ('ntsc_comb_simple', (720, 24), 2, 1), ('ntsc_comb_3d', (720, 480), 2, 1),
])
def test_frames_demod_vs_oracle(stack, size, n_frames, first):
    from oracle import cm_oracle
    modem = stacks.make(stack, size)
    comp = testing.synthetic_composite(n_frames, size[1], size[0], seed=900 + size[1])
    got = image.ImageModem(modem).demodulate_frames(comp, first_frame=first)
    want = cm_oracle.demodulate_frames_f32(modem, comp, first_frame=first, n_threads=8)
    for i in range(n_frames):
        assert stacks.rel_err(got[i], want[i]) < TOL, (stack, i)


# ---- modulators ---------------------------------------------------------------------------------------
MOD_FRAMES = sorted(os.path.basename(p)[:-4] for p in glob.glob(os.path.join(stacks.GOLDEN, 'frames_mod_*.npz')))


@pytest.mark.parametrize('name', MOD_FRAMES)
def test_frames_mod_golden(name):
    g = stacks.load(name)
    modem = stacks.make(stack_of(name, 'frames_mod_'), g['size'])
    im = image.ImageModem(modem)
    for i, f in enumerate(g['frames']):
        out = im.modulate_frames(g['inp'][i:i + 1], first_frame=int(f))[0]
        assert stacks.rel_err(out, g['out'][i]) < TOL, (name, int(f))


@pytest.mark.parametrize('stack,size,n_frames,first', [
    ('pal_s', (720, 576), 2, 1), ('ntsc', (720, 480), 2, 0), ('pal_avg', (720, 31), 3, 2), ('ntsc_avg', (704, 16), 2, 1),
    ('secam', (720, 576), 2, 5), ('secam_avg', (720, 33), 7, 0),
])
def test_frames_mod_vs_oracle(stack, size, n_frames, first):
    from oracle import cm_oracle
    modem = stacks.make(stack, size)
    rgb = testing.synthetic_rgb(n_frames, size[1], size[0], seed=40 + size[1])
    got = image.ImageModem(modem).modulate_frames(rgb, first_frame=first)
    want = cm_oracle.modulate_frames_f32(modem, rgb, first_frame=first, n_threads=8)
    for i in range(n_frames):
        assert stacks.rel_err(got[i], want[i]) < TOL, (stack, i)


def test_rows_mod_protocol():
    """Modem.modulate row by row, including the reset when a run is broken, against the oracle object."""
    from oracle import cm_oracle
    for stack in ('pal_s', 'pal_avg'):
        modem = stacks.make(stack, (720, 576), explicit=False)
        orc = cm_oracle.OracleModem(modem)
        rgb = testing.synthetic_rgb(1, 8, 720, seed=61)[0]
        seq = [(2, 0), (2, 2), (2, 4), (2, 9), (2, 11), (3, 13), (3, 15), (3, 17)]
        for i, (f, y) in enumerate(seq):
            got = modem.modulate(f, y, rgb[0, i], rgb[1, i], rgb[2, i])
            want = orc.modulate(f, y, rgb[0, i], rgb[1, i], rgb[2, i])
            assert stacks.rel_err(got, want) < TOL, (stack, f, y)


# ---- SECAM decoder on valid signals (the FM discriminator is ill-conditioned on noise) -----------------
@pytest.mark.parametrize('size,n_frames,first', [((720, 576), 2, 3), ((720, 17), 4, 0), ((704, 8), 2, 1)])
def test_secam_demod_vs_oracle(size, n_frames, first):
    from oracle import cm_oracle
    modem = stacks.make('secam', size)
    rgb = testing.synthetic_rgb(n_frames, size[1], size[0], seed=70 + size[1])
    comp = cm_oracle.modulate_frames_f32(modem, rgb, first_frame=first, n_threads=8)
    got = image.ImageModem(modem).demodulate_frames(comp, first_frame=first)
    want = cm_oracle.demodulate_frames_f32(modem, comp, first_frame=first, n_threads=8)
    for i in range(n_frames):
        assert stacks.rel_err(got[i], want[i]) < TOL, i


def test_secam_round_trip_on_device():
    """BASELINE.json configs[3]: encode + decode on the GPU; the round trip must equal the oracle's round trip."""
    from oracle import cm_oracle
    modem = stacks.make('secam', (720, 576))
    im = image.ImageModem(modem)
    rgb = testing.synthetic_rgb(2, 576, 720, seed=99)
    comp = im.modulate_frames(rgb, first_frame=4)
    back = im.demodulate_frames(comp, first_frame=4)
    comp_ref = cm_oracle.modulate_frames_f32(modem, rgb, first_frame=4, n_threads=8)
    back_ref = cm_oracle.demodulate_frames_f32(modem, comp_ref, first_frame=4, n_threads=8)
    assert stacks.rel_err(comp, comp_ref) < TOL
    # each leg against the oracle on the SAME input: strict
    assert stacks.rel_err(im.demodulate_frames(comp_ref, first_frame=4), back_ref) < TOL
    back_of_gpu_comp = cm_oracle.demodulate_frames_f32(modem, comp, first_frame=4, n_threads=8)
    assert stacks.rel_err(back, back_of_gpu_comp) < TOL
    # end to end: the FM decoder amplifies input differences (frequency / fdev), the oracle's as much as the kernel's - the
    # round trips may differ by what the ORACLE decoder makes of the encoder's float32 rounding, plus the tolerance
    assert stacks.rel_err(back, back_ref) <= stacks.rel_err(back_of_gpu_comp, back_ref) + TOL


# ---- the BASELINE.json frame sizes under both criteria -------------------------------------------
@pytest.mark.parametrize('stack,enc,size,first', [
    ('pal_d', 'pal_s', (720, 576), 1), ('ntsc_comb_3d', 'ntsc', (720, 480), 0), ('secam', 'secam', (720, 576), 2),
    ('pal_3d', 'pal_s', (720, 576), 3), ('ntsc', 'ntsc', (720, 480), 1),
])
def test_baseline_sizes_allclose(stack, enc, size, first):
    """max |out - ref| / max |ref| <= 1e-5 per plane AND numpy.allclose(rtol=1e-5, atol=1e-6) sample by sample, on a valid
    colour signal at the benchmark frame sizes; the per-plane figures go to gpurun_out/parity_report.txt."""
    from oracle import cm_oracle
    modem = stacks.make(stack, size)
    rgb = testing.synthetic_rgb(2, size[1], size[0], seed=31 + size[1])
    comp = cm_oracle.modulate_frames_f32(stacks.make(enc, size), rgb, first_frame=first, n_threads=8)
    got = image.ImageModem(modem).demodulate_frames(comp, first_frame=first)
    want = cm_oracle.demodulate_frames_f32(modem, comp, first_frame=first, n_threads=8)
    report = stacks.parity_report(got, want)
    try:
        os.makedirs(os.path.join(os.path.dirname(stacks.GOLDEN), '..', 'gpurun_out'), exist_ok=True)
        with open(os.path.join(os.path.dirname(stacks.GOLDEN), '..', 'gpurun_out', 'parity_report.txt'), 'a') as fh:
            for plane, (err, bad, n) in zip('RGB', report):
                fh.write('%-14s %dx%d plane %s: rel_err %.3e, allclose(rtol 1e-5, atol 1e-6) violations %d of %d\n'
                         % (stack, size[0], size[1], plane, err, bad, n))
    except OSError:
        pass
    for err, bad, n in report:
        assert err < TOL, (stack, report)
        assert bad == 0, (stack, report)


# ---- the per-row protocol on device-resident history (engine.RowSession) ---------------------------------------------------
def test_row_protocol_long_run_and_lines_beyond_the_image():
    """Modem.demodulate row by row: a run longer than the session's history buffer (compaction), a switch between
    demodulate() and demodulate_components() in mid-run (two plans, one run state), and line numbers far beyond the
    image height (the reference takes any: line.py:57-65; the plan's per-line tables are grown on demand)."""
    from oracle import cm_oracle
    for stack in ('pal_d', 'ntsc_comb_3d', 'secam'):
        size = (720, 8)
        modem = stacks.make(stack, size)
        orc = cm_oracle.OracleModem(modem)
        comp = testing.synthetic_composite(1, 200, 720, seed=5)[0]
        if stack == 'secam':   # a valid signal for the FM discriminator
            rgb = testing.synthetic_rgb(1, 200, 720, seed=6)[0].astype(numpy.float64)
            enc = cm_oracle.OracleModem(modem)
            comp = numpy.stack([enc.modulate(3, 2 * i, rgb[0, i], rgb[1, i], rgb[2, i]) for i in range(200)]).astype(numpy.float32)
        for i in range(150):                     # lines 0, 2, ..., 298 of frame 3: one run, far beyond 8 rows
            line = 2 * i
            want = orc.demodulate(3, line, comp[i].astype(numpy.float64))
            if stack != 'secam' and i in (70, 71):
                y, u, v = modem.demodulate_components(3, line, comp[i])
                got = modem.decode_components(y, u, v)
            else:
                got = modem.demodulate(3, line, comp[i])
            assert stacks.rel_err(numpy.stack(got), numpy.stack(want)) < TOL, (stack, line)


def test_out_argument_is_validated():
    import torch
    modem = stacks.make('pal_d', (720, 8))
    eng = image.ImageModem(modem)._engine()
    comp = torch.zeros((2, 8, 720), dtype=torch.float32, device='cuda')
    good = torch.empty((2, 3, 8, 720), dtype=torch.float32, device='cuda')
    assert eng.demodulate_frames(comp, 0, out=good) is good
    for bad in (torch.empty((2, 3, 8, 719), dtype=torch.float32, device='cuda'),
                torch.empty((2, 3, 8, 720), dtype=torch.float64, device='cuda'),
                torch.empty((2, 3, 8, 720), dtype=torch.float32),
                torch.empty((2, 3, 8, 1440), dtype=torch.float32, device='cuda')[..., ::2]):
        with pytest.raises(ValueError):
            eng.demodulate_frames(comp, 0, out=bad)


def test_plan_refuses_host_pointers():
    """The C ABI takes device pointers; a host buffer must come back as CM_ERR_INVALID, not as a GPU fault."""
    import ctypes
    from color_modem_amd import _native
    modem = stacks.make('pal_d', (720, 8))
    eng = image.ImageModem(modem)._engine()
    host_in = numpy.zeros((1, 8, 720), dtype=numpy.float32)
    host_out = numpy.zeros((1, 3, 8, 720), dtype=numpy.float32)
    rc = _native.lib().cm_demodulate_frames(eng._plan, host_in.ctypes.data, host_out.ctypes.data, 1, 0, None)
    assert rc == _native.CM_ERR_INVALID
    assert b'device' in _native.lib().cm_last_error()


# ---- comb wrappers around the PAL delay-line decoders (color_modem_amd/wrapped.py) -----------------------------------------
@pytest.mark.parametrize('stack,size,first', [('simple3d_pald', (720, 40), 1), ('simple_pald', (720, 21), 2), ('simple3d_pal3d', (720, 24), 3),
                                              ('simple3d_pald_minavg', (704, 12), 0)])
def test_wrapped_pal_comb_vs_oracle(stack, size, first):
    from oracle import cm_oracle
    modem = stacks.make(stack, size)
    rgb = testing.synthetic_rgb(2, size[1], size[0], seed=12 + size[1])
    comp = cm_oracle.modulate_frames_f32(stacks.make('pal_s', size), rgb, first_frame=first, n_threads=4)
    got = image.ImageModem(modem).demodulate_frames(comp, first_frame=first)
    want = cm_oracle.demodulate_frames_f32(modem, comp, first_frame=first, n_threads=4)
    for i in range(2):
        assert stacks.rel_err(got[i], want[i]) < TOL, (stack, i)
    # the per-row protocol on the same stack, with a break in the run
    orc = cm_oracle.OracleModem(modem)
    for f, y in ((first, 0), (first, 2), (first, 4), (first, 6), (first, 11), (first, 13), (first + 1, 15)):
        row = comp[0, y % size[1]]
        got_row = numpy.stack(modem.demodulate(f, y, row))
        want_row = numpy.stack(orc.demodulate(f, y, row.astype(numpy.float64)))
        assert stacks.rel_err(got_row, want_row) < TOL, (stack, f, y)


# ---- the time-blocked decoder with the half-band FIRs on the matrix pipe (csrc/cm_blk_kernels.h, opt-in: CM_BLK=1) ----------
@pytest.mark.gpu
def test_secam_float32_margin_case_and_the_float64_switch():
    """A composite frame the round-2 fuzz campaign found (seed 301: SECAM III, 640x76, oracle-encoded): with every stage in
    float32 the decoder sat at 1.04e-5 of full scale in ONE start-of-row colour-difference sample (the band-pass starts from
    zero state on a luma step: large states, small sub-carrier).  The float32 kernels now run the band-pass + bell of their
    guarded bodies - the row ends - in float64 (cm_stages.h: SecamBp64): the default path must hold
    1e-5 with margin; `modem.float64_front_end = True` (cm_secam_desc.present | CM_SECAM_FLOAT64) selects the float64 front end."""
    from oracle import cm_oracle
    from color_modem_amd import line
    from color_modem_amd.color import secam
    g = numpy.load(os.path.join(os.path.dirname(__file__), 'golden', 'secam_iii_640_margin.npz'))
    size = [int(x) for x in g['size']]
    comp, first = g['comp'], int(g['first'])

    def run(f64):
        lc = line.LineConfig((size[0], size[1]), line.LineStandard.detect(size[2]))
        modem = secam.SecamModem(lc, getattr(secam.SecamVariant, str(g['vname'])))
        modem.float64_front_end = f64
        im = image.ImageModem(modem)
        got = im.demodulate_frames(comp, first_frame=first)
        assert ('float64' in im._engine().describe()) == f64
        want = cm_oracle.demodulate_frames_f32(modem, comp, first_frame=first, n_threads=4)
        return numpy.abs(numpy.asarray(got, dtype=numpy.float64) - want) / numpy.abs(want).max()

    err = run(False)
    assert err.max() < 2e-6, err.max()          # (1.04e-5 before; tests/sim predicts 7.9e-7)
    err64 = run(True)
    assert err64.max() < 2e-6, err64.max()


def test_blocked_mfma_decoder_parity(monkeypatch):
    """demod_blk_kernel: split-float16 Toeplitz MFMAs for the five FIR chains, luma source added at the flush.  Same
    goldens, same tolerance as the streaming kernel it is an alternative to."""
    from oracle import cm_oracle
    monkeypatch.setenv('CM_BLK', '1')
    for name in ('frames_demod_pal_d', 'frames_demod_pal_d_noise_720x8', 'frames_demod_pal_d_noise_704x7'):
        g = stacks.load(name)
        im = image.ImageModem(stacks.make('pal_d', g['size']))
        assert 'demod_blk_kernel' in im._engine().describe()
        for i, f in enumerate(g['frames']):
            out = im.demodulate_frames(g['inp'][i:i + 1], first_frame=int(f))[0]
            assert stacks.rel_err(out, g['out'][i]) < TOL, (name, int(f))
    modem = stacks.make('pal_d', (720, 576))
    rgb = testing.synthetic_rgb(2, 576, 720, seed=4)
    comp = cm_oracle.modulate_frames_f32(stacks.make('pal_s', (720, 576)), rgb, first_frame=6, n_threads=8)
    got = image.ImageModem(modem).demodulate_frames(comp, first_frame=6)
    want = cm_oracle.demodulate_frames_f32(modem, comp, first_frame=6, n_threads=8)
    for err, bad, n in stacks.parity_report(got, want):
        assert err < TOL and bad == 0, (err, bad)
