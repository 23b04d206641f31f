# -*- coding: utf-8 -*-
"""Generate the golden vectors under tests/golden/ by running the REFERENCE implementation.

Run in the build container only (the reference does not exist on the GPU box):

    python tests/golden/make_golden.py            # writes tests/golden/*.npz, plans.json

What is recorded
----------------
* plans.json   - every plan constant the reference derives for PAL-BG / NTSC-M / SECAM IIIb at
                 720 px (filter (b, a, shift, phase_shift), carrier steps, start-phase and
                 parity tables, the 41-tap resampling FIR), full float64 repr.
* <case>.npz   - `inp` (float32, exactly what is fed to the reference after a cast to float64)
                 and `out` (float64, what the reference returned), for whole small frames run
                 through the row schedule of color_modem/image.py:47-55,75-83 and for explicit
                 (frame, line) row sequences at the full 576/480-line geometry.
* image_*.npz  - uint8 in/out of the reference's own ImageModem on a tiny PIL image.

Harness behaviour that is NOT reference code
--------------------------------------------
* scipy.signal.iirdesign shim: scipy >= 1.? validates `wp, ws > 0`; the
  reference (qam.py:17 via utils.py:55,62) asks for a band edge < 0 for NTSC-M.  The shim
  restores the pre-validation behaviour (buttord + iirfilter).  PAL/SECAM coefficients are
  bit-identical with and without it (asserted below).
* The float frame driver `run_*_frame` below is our restatement of the image.py row schedule
  without the uint8 conversion; the image_* cases pin the uint8 path with the real ImageModem.
"""

import json
import os
import sys
import warnings

import numpy
import scipy
import scipy.signal

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
sys.path.insert(0, ROOT)
sys.path.insert(0, '/root/reference')
warnings.simplefilter('ignore')

from color_modem_amd import testing  # noqa: E402

_orig_iirdesign = scipy.signal.iirdesign


def _legacy_iirdesign(wp, ws, gpass, gstop, analog=False, ftype='ellip', output='ba', fs=None):
    assert ftype == 'butter' and not analog and output == 'ba' and fs is None
    wp = numpy.atleast_1d(wp)
    ws = numpy.atleast_1d(ws)
    band_type = 2 * (len(wp) - 1) + 1
    if wp[0] >= ws[0]:
        band_type += 1
    btype = {1: 'lowpass', 2: 'highpass', 3: 'bandstop', 4: 'bandpass'}[band_type]
    n, wn = scipy.signal.buttord(wp, ws, gpass, gstop, analog=False)
    return scipy.signal.iirfilter(n, wn, rp=gpass, rs=gstop, analog=False, btype=btype, ftype='butter',
                                  output='ba')


def use_shim(on):
    scipy.signal.iirdesign = _legacy_iirdesign if on else _orig_iirdesign


use_shim(True)

from color_modem import comb, image, line  # noqa: E402
from color_modem.color import ntsc, pal, secam  # noqa: E402

LS = line.LineStandard


def filt(f):
    return {'b': [repr(float(v)) for v in f._b], 'a': [repr(float(v)) for v in f._a],
            'shift': int(f._shift), 'phase_shift': repr(float(f.phase_shift))}


def qam_plan(modem, lc, height, n_frames, extra_lines=4):
    q = modem.qam
    d = {
        'fs': repr(float(lc.fs)),
        'fsc': repr(float(modem.config.fsc)),
        'carrier_phase_step': repr(float(q.carrier_phase_step)),
        'line_shift': repr(float(modem.line_shift)),
        'frame_shift': repr(float(modem.frame_shift)),
        'frame_cycle': int(modem.frame_cycle),
        'precorrect': filt(q._chroma_precorrect_lowpass),
        'extract2x': filt(q._extract_chroma2x),
        'remove2x': filt(q._remove_chroma2x),
        'demod_lp': filt(q._demod_lowpass),
        'start_phase': [[repr(float(modem.start_phase(f, y))) for y in range(height + extra_lines)]
                        for f in range(n_frames)],
        'alt': [[bool(lc.is_alternate_line(f, y)) for y in range(height + extra_lines)] for f in range(n_frames)],
        'analog_line': [int(lc.analog_line(y)) for y in range(height + extra_lines)],
    }
    return d


def make_plans():
    plans = {'versions': {'numpy': numpy.__version__, 'scipy': scipy.__version__,
                          'note': 'scipy.signal.iirdesign shimmed (buttord+iirfilter), see module docstring'}}
    h = scipy.signal.firwin(41, 0.5, window=('kaiser', 5.0))
    plans['resample_fir'] = [repr(float(v)) for v in h]

    # the shim must not move PAL/SECAM coefficients
    use_shim(False)
    p0 = pal.PalDModem(line.LineConfig((720, 576)))
    s0 = secam.SecamModem(line.LineConfig((720, 576)))
    use_shim(True)
    p1 = pal.PalDModem(line.LineConfig((720, 576)))
    s1 = secam.SecamModem(line.LineConfig((720, 576)))
    for n in ('_chroma_precorrect_lowpass', '_extract_chroma2x', '_remove_chroma2x', '_demod_lowpass'):
        a, b = getattr(p0.backend.qam, n), getattr(p1.backend.qam, n)
        assert numpy.array_equal(a._b, b._b) and numpy.array_equal(a._a, b._a), n
    for n in ('_chroma_precorrect_lowpass', '_chroma_demod_bell'):
        a, b = getattr(s0, n), getattr(s1, n)
        assert numpy.array_equal(a._b, b._b) and numpy.array_equal(a._a, b._a), n

    lc = line.LineConfig((720, 576))
    m = pal.PalDModem(lc)
    d = qam_plan(m.backend, lc, 576, 6)
    d['pald_lp'] = filt(m._filter)
    d['sin_factor'] = repr(float(m._sin_factor))
    d['cos_factor'] = repr(float(m._cos_factor))
    plans['pal_720x576'] = d

    lc = line.LineConfig((720, 480))
    m = ntsc.NtscCombModem(lc)
    d = qam_plan(m.backend, lc, 480, 4)
    d['comb_factor'] = repr(float(m._factor))
    plans['ntsc_720x480'] = d

    lc = line.LineConfig((720, 576))
    m = secam.SecamModem(lc)
    d = {
        'fs': repr(float(lc.fs)),
        'fsc_dr': repr(float(m._fsc_dr)), 'fsc_db': repr(float(m._fsc_db)),
        'fdev_dr': repr(float(m._fdev_dr)), 'fdev_db': repr(float(m._fdev_db)),
        'flimit_min': repr(float(m._flimit_min)), 'flimit_max': repr(float(m._flimit_max)),
        'bell_f0': repr(float(m._bell_f0)),
        'm0': repr(float(m._variant.m0)), 'bell_kn': repr(float(m._variant.bell_kn)),
        'bell_kd': repr(float(m._variant.bell_kd)),
        'precorrect_lp': filt(m._chroma_precorrect_lowpass),
        'lf_precorrect': filt(m._chroma_precorrect),
        'lf_reverse': filt(m._reverse_chroma_precorrect),
        'bell': filt(m._chroma_demod_bell),
        'chroma_bp': filt(m._chroma_demod_chroma_filter),
        'luma_bs': filt(m._chroma_demod_luma_filter),
        'fm_lp': filt(m._chroma_demod._lowpass),
        'fm_fc': repr(float(m._chroma_demod._fc)),
        'start_phase_inverted': [[bool(m._start_phase_inverted(f, y)) for y in range(580)] for f in range(12)],
        'alt': [[bool(lc.is_alternate_line(f, y)) for y in range(580)] for f in range(4)],
    }
    plans['secam_720x576'] = d
    with open(os.path.join(HERE, 'plans.json'), 'w') as fh:
        json.dump(plans, fh, indent=1)


# ---------------------------------------------------------------------------------------------
# float frame drivers: the row schedule of image.py without the uint8 conversion

def run_demod_frame(modem, comp, frame):
    height, width = comp.shape
    delay = getattr(modem, 'demodulation_delay', 0)
    out = numpy.zeros((3, height, width))
    for field in range(2):
        for y in range(field, 2 * delay, 2):
            modem.demodulate(frame, y, comp[y])
        for y in range(field, height, 2):
            iy = y + 2 * delay
            while iy >= height:
                iy -= 2
            r, g, b = modem.demodulate(frame, y + 2 * delay, comp[iy])
            out[0, y], out[1, y], out[2, y] = r, g, b
    return out


def run_mod_frame(modem, rgb, frame):
    _, height, width = rgb.shape
    delay = getattr(modem, 'modulation_delay', 0)
    out = numpy.zeros((height, width))
    for field in range(2):
        for y in range(field, 2 * delay, 2):
            modem.modulate(frame, y, rgb[0, y], rgb[1, y], rgb[2, y])
        for y in range(field, height, 2):
            iy = y + 2 * delay
            while iy >= height:
                iy -= 2
            out[y] = modem.modulate(frame, y + 2 * delay, rgb[0, iy], rgb[1, iy], rgb[2, iy])
    return out


STACKS = {
    'pal_s': lambda lc: pal.PalSModem(lc),
    'pal_d': lambda lc: pal.PalDModem(lc),
    'pal_3d': lambda lc: pal.Pal3DModem(lc),
    'ntsc': lambda lc: ntsc.NtscModem(lc),
    'ntsc_comb': lambda lc: ntsc.NtscCombModem(lc),
    'ntsc_comb_simple': lambda lc: comb.SimpleCombModem(ntsc.NtscCombModem(lc)),
    'ntsc_comb_3d': lambda lc: comb.Simple3DCombModem(ntsc.NtscCombModem(lc)),
    'secam': lambda lc: secam.SecamModem(lc),
    'secam_avg': lambda lc: comb.ColorAveragingModem(secam.SecamModem(lc)),
}

# options and variants: luma notch, sign-aware minimum averaging, SECAM without bell /
# LF pre-emphasis, PAL-D on the PAL-M filter shapes, sub-carrier cycles of 4800 frames (4.43 MHz on 525 lines)
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
    'pal_d_60': lambda lc: pal.PalDModem(lc),
    'pal_s_60': lambda lc: pal.PalSModem(lc),
})

# one representative of every variant family that had no reference-generated vector ("what's weak" 2):
# PAL-N, NTSC-N and NTSC 3.61 on 625 / 525 lines, SECAM III, M, N, A
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

# comb wrappers around the PAL delay-line decoders (22 of the 140 cells of the support matrix)
STACKS.update({
    'simple3d_pald': lambda lc: comb.Simple3DCombModem(pal.PalDModem(lc)),
    'simple_pald': lambda lc: comb.SimpleCombModem(pal.PalDModem(lc)),
    'simple3d_pal3d': lambda lc: comb.Simple3DCombModem(pal.Pal3DModem(lc)),
    'simple3d_pald_minavg': lambda lc: comb.Simple3DCombModem(pal.PalDModem(lc), avg=comb.minavg),
    'simple3d_pald_notch': lambda lc: comb.Simple3DCombModem(pal.PalDModem(lc), notch=6.0),
    'simple_pal3d_notch': lambda lc: comb.SimpleCombModem(pal.Pal3DModem(lc), notch=3.0, avg=comb.minavg),
})


# avg= callables of the caller's own (comb.py:72, 81-84) - the same two functions in tests/stacks.py and tests/golden/make_golden.py
def weighted_avg(last, curr):
    return 0.25 * last + 0.75 * curr


def damped_avg(last, curr):      # non-linear and continuous (a discontinuous pick turns float32 rounding of its inputs into a different branch)
    return 0.5 * (last + curr) / (1.0 + 4.0 * abs(last - curr))


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
    # Pal3DModem's OWN average of its two estimates (pal.py:144-148, 176-179, 209-211) as a callable
    'pal_3d_favg': lambda lc: pal.Pal3DModem(lc, avg=damped_avg, notch=4.0),
    'pal_3d_wavg': lambda lc: pal.Pal3DModem(lc, avg=weighted_avg),
})
STANDARD = {'pal': 'GERBER_625', 'ntsc': 'NTSC_525', 'secam': 'GERBER_625', 'simple3d': 'GERBER_625', 'simple': 'GERBER_625'}
STANDARD_OF = {'pal_d_palm': 'NTSC_525', 'pal_s_palm': 'NTSC_525', 'pal_d_60': 'NTSC_525', 'pal_s_60': 'NTSC_525',
               'ntsc_comb_n': 'GERBER_625', 'ntsc_n': 'GERBER_625', 'ntsc_comb_i': 'GERBER_625', 'ntsc_i': 'GERBER_625',
               'secam_m': 'NTSC_525', 'secam_a': 'BAIRD_405', 'simple_ntsc_favg': 'NTSC_525', 'simple3d_ntsccomb_favg': 'NTSC_525'}


def line_config(stack, size):
    std = getattr(LS, STANDARD_OF.get(stack, STANDARD[stack.split('_')[0]]))
    return line.LineConfig(size, std)


def save(name, **arrays):
    path = os.path.join(HERE, name + '.npz')
    numpy.savez_compressed(path, **arrays)
    print('%-28s %8.1f KB' % (name, os.path.getsize(path) / 1024.0))


def frame_cases():
    W, H = 720, 8
    # (case, modulating stack for a valid input signal, demodulating stack, frames)
    demods = [
        ('pal_d', 'pal_s', [0, 1, 2, 3, 5]),
        ('pal_s', 'pal_s', [0, 3]),
        ('pal_3d', 'pal_s', [0, 1, 2, 3]),
        ('ntsc', 'ntsc', [0, 1]),
        ('ntsc_comb', 'ntsc', [0, 1, 2]),
        ('ntsc_comb_simple', 'ntsc', [0, 1]),
        ('ntsc_comb_3d', 'ntsc', [0, 1, 3]),
        ('secam', 'secam', [0, 1, 2, 7]),
    ]
    mods = [
        ('pal_s', [0, 1, 2, 3]),
        ('ntsc', [0, 1]),
        ('secam', [0, 1, 2, 3, 4, 5, 6]),
        ('secam_avg', [0, 1]),
    ]
    for stack, frames in mods:
        lc = line_config(stack, (W, H))
        modem = STACKS[stack](lc)
        rgb = testing.synthetic_rgb(len(frames), H, W, seed=100)
        out = numpy.stack([run_mod_frame(modem, rgb[i].astype(numpy.float64), f) for i, f in enumerate(frames)])
        save('frames_mod_' + stack, inp=rgb, out=out, frames=numpy.array(frames), size=numpy.array([W, H]))
    for stack, mod_stack, frames in demods:
        lc = line_config(stack, (W, H))
        enc = STACKS[mod_stack](lc)
        rgb = testing.synthetic_rgb(len(frames), H, W, seed=200)
        comp = numpy.stack([run_mod_frame(enc, rgb[i].astype(numpy.float64), f) for i, f in enumerate(frames)])
        comp = comp.astype(numpy.float32)
        modem = STACKS[stack](lc)
        out = numpy.stack([run_demod_frame(modem, comp[i].astype(numpy.float64), f) for i, f in enumerate(frames)])
        save('frames_demod_' + stack, inp=comp, out=out, frames=numpy.array(frames), size=numpy.array([W, H]))
    # noise input (not a valid colour signal) through the headline decoder, odd height, other width
    for stack, (w, h), frames in [('pal_d', (720, 8), [0, 1, 2, 3]), ('pal_d', (704, 7), [1, 2]),
                                  ('ntsc_comb_3d', (720, 7), [0, 1])]:
        lc = line_config(stack, (w, h))
        comp = testing.synthetic_composite(len(frames), h, w, seed=300)
        modem = STACKS[stack](lc)
        out = numpy.stack([run_demod_frame(modem, comp[i].astype(numpy.float64), f) for i, f in enumerate(frames)])
        save('frames_demod_%s_noise_%dx%d' % (stack, w, h), inp=comp, out=out, frames=numpy.array(frames),
             size=numpy.array([w, h]))


def option_cases(only=()):
    """Options and variants beyond the three headline systems, 720x8 frames."""
    W, H = 720, 8
    demods = [
        ('pal_d_notch', 'pal_s', [0, 3]),
        ('pal_3d_notch', 'pal_s', [1, 2]),
        ('ntsc_comb_3d_notch', 'ntsc', [0, 1]),
        ('pal_3d_minavg', 'pal_s', [0, 1, 2]),
        ('pal_3d_sin', 'pal_s', [1, 2]),
        ('pal_3d_cos', 'pal_s', [0, 3]),
        ('ntsc_simple_minavg', 'ntsc', [0, 1]),
        ('ntsc_comb_3d_minavg', 'ntsc', [0, 1]),
        ('secam_i', 'secam_i', [0, 1]),
        ('secam_ii', 'secam_ii', [0, 3]),
        ('pal_d_palm', 'pal_s_palm', [0, 1, 2, 3]),
        ('ntsc_comb_443', 'ntsc_443', [0, 1, 4799, 4802]),
        ('pal_d_60', 'pal_s_60', [1, 2402, 4799, 4800]),
        ('ntsc_comb_a', 'ntsc_a', [0, 1, 2]),      # NTSC-A: order-8 band-pass, odd shifts, two-section pre-correction
        ('ntsc_a', 'ntsc_a', [1, 4]),
    ]
    mods = [('secam_i', [0, 2]), ('secam_ii', [1, 5]), ('pal_s_60', [3, 4798]), ('ntsc_443', [0, 4797]), ('ntsc_a', [0, 3])]
    if only:
        mods = [m for m in mods if m[0] in only]
        demods = [d for d in demods if d[0] in only]
    for stack, frames in mods:
        lc = line_config(stack, (W, H))
        modem = STACKS[stack](lc)
        rgb = testing.synthetic_rgb(len(frames), H, W, seed=600)
        out = numpy.stack([run_mod_frame(modem, rgb[i].astype(numpy.float64), f) for i, f in enumerate(frames)])
        save('frames_mod_' + stack, inp=rgb, out=out, frames=numpy.array(frames), size=numpy.array([W, H]))
    for stack, mod_stack, frames in demods:
        lc = line_config(stack, (W, H))
        enc = STACKS[mod_stack](lc)
        rgb = testing.synthetic_rgb(len(frames), H, W, seed=700)
        comp = numpy.stack([run_mod_frame(enc, rgb[i].astype(numpy.float64), f) for i, f in enumerate(frames)])
        comp = comp.astype(numpy.float32)
        modem = STACKS[stack](lc)
        out = numpy.stack([run_demod_frame(modem, comp[i].astype(numpy.float64), f) for i, f in enumerate(frames)])
        save('frames_demod_' + stack, inp=comp, out=out, frames=numpy.array(frames), size=numpy.array([W, H]))


def width_cases():
    """Other sampling rates (image widths): different filter orders / shift parities than at 13.5 MHz, W x 6 frames."""
    H = 6
    for stack, mod_stack, W, frames in [('pal_d', 'pal_s', 768, [0, 3]), ('ntsc_comb', 'ntsc', 640, [0, 1]), ('pal_3d', 'pal_s', 1024, [1, 2])]:
        lc = line_config(stack, (W, H))
        enc = STACKS[mod_stack](lc)
        rgb = testing.synthetic_rgb(len(frames), H, W, seed=800 + W)
        comp = numpy.stack([run_mod_frame(enc, rgb[i].astype(numpy.float64), f) for i, f in enumerate(frames)])
        save('frames_mod_%s_w%d' % (mod_stack, W), inp=rgb, out=comp, frames=numpy.array(frames), size=numpy.array([W, H]))
        comp = comp.astype(numpy.float32)
        modem = STACKS[stack](lc)
        out = numpy.stack([run_demod_frame(modem, comp[i].astype(numpy.float64), f) for i, f in enumerate(frames)])
        save('frames_demod_%s_w%d' % (stack, W), inp=comp, out=out, frames=numpy.array(frames), size=numpy.array([W, H]))


def variant_cases(only=()):
    """Reference vectors for the variant families and image widths that were checked HIP-vs-oracle only: W x 6 frames, a
    valid signal from the matching encoder, both directions."""
    H = 6
    cases = [  # (decoder stack, encoder stack, width, frames)
        ('pal_d_paln', 'pal_s_paln', 720, [0, 1, 2, 3]),
        ('ntsc_comb_n', 'ntsc_n', 720, [0, 1, 2]),
        ('ntsc_comb_3d_361', 'ntsc_361', 720, [0, 1, 5]),
        ('ntsc_comb_i', 'ntsc_i', 720, [1, 2]),
        ('secam_iii', 'secam_iii', 720, [0, 5]),
        ('secam_m', 'secam_m', 720, [1, 2]),
        ('secam_n', 'secam_n', 720, [0, 3]),
        ('secam_a', 'secam_a', 720, [2, 3]),
        ('pal_d', 'pal_s', 480, [1, 2]),
        ('pal_d', 'pal_s', 960, [0, 3]),
        ('ntsc_comb_3d', 'ntsc', 1280, [0, 1]),
        ('pal_3d', 'pal_s', 1920, [2]),
        ('secam', 'secam', 960, [0, 1]),
        ('secam', 'secam', 1280, [3]),
        # round 6: the tuned instances of the wide rasters (csrc/cm_shapes_wide.h) against the reference itself, one width per stack
        ('pal_d', 'pal_s', 1024, [0]),
        ('pal_d', 'pal_s', 1280, [1, 2]),
        ('pal_d', 'pal_s', 1920, [3]),
        ('pal_3d', 'pal_s', 1440, [0]),
        ('pal_s', 'pal_s', 1600, [1]),
        ('pal_d', 'pal_s', 800, [2]),
        ('ntsc_comb', 'ntsc', 960, [0, 1]),
        ('ntsc_comb', 'ntsc', 1920, [1]),
        ('ntsc_comb_3d', 'ntsc', 1440, [0]),
        ('ntsc', 'ntsc', 1600, [1]),
        ('ntsc', 'ntsc', 1024, [0]),
        ('secam', 'secam', 1920, [0]),
    ]
    for stack, mod_stack, W, frames in cases:
        tag = stack if W == 720 else '%s_w%d' % (stack, W)
        mtag = mod_stack if W == 720 else '%s_w%d' % (mod_stack, W)
        if only and tag not in only:
            continue
        lc = line_config(stack, (W, H))
        enc = STACKS[mod_stack](lc)
        rgb = testing.synthetic_rgb(len(frames), H, W, seed=900 + W + len(stack))
        comp = numpy.stack([run_mod_frame(enc, rgb[i].astype(numpy.float64), f) for i, f in enumerate(frames)])
        save('frames_mod_' + mtag, inp=rgb, out=comp, frames=numpy.array(frames), size=numpy.array([W, H]))
        comp = comp.astype(numpy.float32)
        modem = STACKS[stack](lc)
        out = numpy.stack([run_demod_frame(modem, comp[i].astype(numpy.float64), f) for i, f in enumerate(frames)])
        save('frames_demod_' + tag, inp=comp, out=out, frames=numpy.array(frames), size=numpy.array([W, H]))


def wrapper_cases(only=()):
    """SimpleCombModem / Simple3DCombModem around PalDModem and Pal3DModem: 720 x 10 frames of a valid PAL signal."""
    W, H = 720, 10
    for stack, frames in (('simple3d_pald', [1, 2]), ('simple_pald', [1, 2]), ('simple3d_pal3d', [1, 2]), ('simple3d_pald_minavg', [0, 3]),
                          ('simple3d_pald_notch', [0, 3]), ('simple_pal3d_notch', [1, 2])):
        if only and stack not in only:
            continue
        lc = line_config(stack, (W, H))
        enc = STACKS['pal_s'](lc)
        rgb = testing.synthetic_rgb(len(frames), H, W, seed=321)
        comp = numpy.stack([run_mod_frame(enc, rgb[i].astype(numpy.float64), f) for i, f in enumerate(frames)]).astype(numpy.float32)
        out = numpy.stack([run_demod_frame(STACKS[stack](lc), comp[i].astype(numpy.float64), f) for i, f in enumerate(frames)])
        save('frames_demod_' + stack, inp=comp, out=out, frames=numpy.array(frames), size=numpy.array([W, H]))


def callable_cases(only=()):
    """SimpleCombModem / Simple3DCombModem with an avg= callable of the caller's own (comb.py:72, 81-84, 103-104)."""
    W, H = 720, 10
    for stack, frames in (('simple3d_pald_favg', [1, 2]), ('simple_pal3d_favg', [0, 3]), ('simple_ntsc_favg', [1, 2]), ('simple3d_ntsccomb_favg', [0, 1]),
                          ('pal_3d_favg', [0, 3]), ('pal_3d_wavg', [1, 2])):
        if only and stack not in only:
            continue
        lc = line_config(stack, (W, H))
        enc = STACKS['ntsc' if 'ntsc' in stack else 'pal_s'](lc)
        rgb = testing.synthetic_rgb(len(frames), H, W, seed=654)
        comp = numpy.stack([run_mod_frame(enc, rgb[i].astype(numpy.float64), f) for i, f in enumerate(frames)]).astype(numpy.float32)
        out = numpy.stack([run_demod_frame(STACKS[stack](lc), comp[i].astype(numpy.float64), f) for i, f in enumerate(frames)])
        save('frames_demod_' + stack, inp=comp, out=out, frames=numpy.array(frames), size=numpy.array([W, H]))


def notch_shift_cases():
    """notch= values whose FilterFunction comes out with a shift other than 0."""
    W, H = 720, 10
    for stack, frames in (('pal_d_notchq1', [1, 2]), ('pal_3d_notchq07', [0, 3]), ('ntsc_comb_3d_notchq1', [0, 1]), ('simple_pald_notchq1', [2, 3])):
        lc = line_config(stack, (W, H))
        modem = STACKS[stack](lc)
        notch = getattr(modem, 'notch', None) or getattr(modem, '_notch', None)
        print('   %s: notch shift %d' % (stack, notch._shift))
        assert notch._shift != 0
        enc = STACKS['ntsc' if 'ntsc' in stack else 'pal_s'](lc)
        rgb = testing.synthetic_rgb(len(frames), H, W, seed=777)
        comp = numpy.stack([run_mod_frame(enc, rgb[i].astype(numpy.float64), f) for i, f in enumerate(frames)]).astype(numpy.float32)
        out = numpy.stack([run_demod_frame(STACKS[stack](lc), comp[i].astype(numpy.float64), f) for i, f in enumerate(frames)])
        save('frames_demod_' + stack, inp=comp, out=out, frames=numpy.array(frames), size=numpy.array([W, H]))


def degenerate_pictures(W, H):
    """The inputs of the degenerate_* sets (shared with make_golden_am.py / make_golden_mac.py): black / white / mid-grey / saturated red pictures
    and all-zero / constant composites - where an algorithm divides by an amplitude or takes the angle of a vanishing pair."""
    one, zero = numpy.ones((H, W), numpy.float32), numpy.zeros((H, W), numpy.float32)
    pics = numpy.stack([numpy.stack([zero, zero, zero]), numpy.stack([one, one, one]),
                        numpy.full((3, H, W), numpy.float32(200 / 255.0)), numpy.stack([one, zero, zero])])
    comps = numpy.stack([zero, numpy.full((H, W), numpy.float32(0.3))])
    return pics, ['black', 'white', 'grey 200/255', 'red'], comps, ['zero', 'constant 0.3']


def degenerate_cases():
    """One tiny reference-generated set per family on degenerate inputs (until round 4 these were checked GPU-vs-oracle
    only, and the oracle borrows its coefficients from the product): every picture through the encoder, every composite through the decoder,
    720 x 12, frame 1, the image.py row schedule."""
    W, H, frame = 720, 12, 1
    pics, pic_names, comps, comp_names = degenerate_pictures(W, H)
    for stack in ('pal_s', 'pal_d', 'pal_3d', 'ntsc', 'ntsc_comb_3d', 'secam', 'secam_avg', 'simple3d_pald'):
        lc = line_config(stack, (W, H))
        mod_out = numpy.stack([run_mod_frame(STACKS[stack](lc), pics[i].astype(numpy.float64), frame) for i in range(len(pics))])
        with numpy.errstate(all='ignore'):
            demod_out = numpy.stack([run_demod_frame(STACKS[stack](lc), comps[i].astype(numpy.float64), frame) for i in range(len(comps))])
        save('degenerate_' + stack, pics=pics, pic_names=numpy.array(pic_names), mod_out=mod_out, comps=comps, comp_names=numpy.array(comp_names),
             demod_out=demod_out, frame=numpy.array(frame), size=numpy.array([W, H]))


def row_cases():
    """Explicit (frame, line) sequences at the full-height geometry, fed to one modem object in order."""
    seqs = {
        'pal_d': ((720, 576), [(1, 0), (1, 2), (1, 4), (1, 571), (1, 573), (1, 575), (2, 1), (2, 3)]),
        'pal_3d': ((720, 576), [(1, 0), (1, 2), (1, 4), (1, 6), (3, 573), (3, 575), (3, 577)]),
        'ntsc_comb_3d': ((720, 480), [(1, 1), (1, 3), (1, 5), (1, 7), (0, 476), (0, 478), (0, 480)]),
        'ntsc_comb': ((720, 480), [(0, 0), (0, 2), (0, 4), (1, 477), (1, 479)]),
        'secam': ((720, 576), [(0, 0), (0, 2), (0, 4), (3, 571), (3, 573), (3, 575)]),
    }
    enc_of = {'pal_d': 'pal_s', 'pal_3d': 'pal_s', 'ntsc_comb_3d': 'ntsc', 'ntsc_comb': 'ntsc', 'secam': 'secam'}
    for stack, (size, seq) in seqs.items():
        lc = line.LineConfig(size)
        enc = STACKS[enc_of[stack]](lc)
        modem = STACKS[stack](lc)
        rgb = testing.synthetic_rgb(1, len(seq), size[0], seed=400)[0]
        comp = numpy.stack([enc.modulate(f, y, *[rgb[c, i].astype(numpy.float64) for c in range(3)])
                            for i, (f, y) in enumerate(seq)]).astype(numpy.float32)
        out = numpy.stack([numpy.stack(modem.demodulate(f, y, comp[i].astype(numpy.float64)))
                           for i, (f, y) in enumerate(seq)])
        save('rows_demod_' + stack, inp=comp, out=out, seq=numpy.array(seq), size=numpy.array(size))


def image_cases():
    """uint8 through the reference's own ImageModem (image.py:27-84) on a 720x8 image."""
    from PIL import Image
    W, H = 720, 8
    rgb = testing.synthetic_rgb(1, H, W, seed=500)[0]
    rgb8 = numpy.uint8(numpy.rint(255.0 * rgb)).transpose(1, 2, 0).copy()
    img = Image.frombytes('RGB', (W, H), rgb8.tobytes())
    for stack in ('pal_d', 'ntsc_comb_3d', 'secam_avg'):
        lc = line_config(stack, (W, H))
        im = image.ImageModem(STACKS[stack](lc))
        comp_img = im.modulate(img, 1)
        back = im.demodulate(comp_img, 1)
        save('image_' + stack, rgb8=rgb8, comp8=numpy.frombuffer(comp_img.tobytes(), dtype=numpy.uint8).reshape(H, W),
             back8=numpy.frombuffer(back.tobytes(), dtype=numpy.uint8).reshape(H, W, 3), frame=numpy.array(1))


def test_picture(width, height):
    """A deterministic picture with the structure of real ones (and one that compresses): colour bars over a vertical luminance ramp,
    a zone plate (fine detail at every orientation), a saturated-transition checker, 0.5 % of hash noise; uint8 [H, W, 3]."""
    x = numpy.arange(width)[None, :] / float(width)
    y = numpy.arange(height)[:, None] / float(height)
    bars = numpy.array([[1, 1, 1], [1, 1, 0], [0, 1, 1], [0, 1, 0], [1, 0, 1], [1, 0, 0], [0, 0, 1], [0, 0, 0]], dtype=numpy.float64)
    rgb = bars[numpy.minimum((x * 8).astype(int), 7)[0]][None, :, :] * (0.25 + 0.75 * (1.0 - y))[:, :, None]
    cx, cy = x - 0.5, (y - 0.6) * height / float(width)
    zone = 0.5 + 0.5 * numpy.cos(900.0 * (cx * cx + cy * cy))
    inside = ((cx * cx + cy * cy) < 0.03)
    rgb = numpy.where(inside[:, :, None], zone[:, :, None] * numpy.array([1.0, 0.8, 0.6])[None, None, :], rgb * numpy.ones((height, 1, 1)))
    check = ((numpy.arange(width)[None, :] // 12 + numpy.arange(height)[:, None] // 9) % 2).astype(numpy.float64)
    band = (y > 0.88) * numpy.ones((1, width))
    rgb = numpy.where(band[:, :, None] > 0, numpy.stack([check, 1.0 - check, 0.5 * check], axis=2), rgb)
    rgb = 0.995 * rgb + 0.005 * testing.hash_uniform((height, width, 3), 977)
    return numpy.uint8(numpy.rint(255.0 * numpy.clip(rgb, 0.0, 1.0)))


def full_image_cases():
    """uint8 through the reference's own ImageModem at FULL height (720x576: PAL-D and SECAM): the byte
    boundary pinned at a realistic size, both directions."""
    from PIL import Image
    for stack, (W, H) in (('pal_d', (720, 576)), ('secam', (720, 576)), ('ntsc_comb_3d', (720, 480))):      # (round 6: BASELINE configs[2] at its full size)
        rgb8 = test_picture(W, H)
        img = Image.frombytes('RGB', (W, H), rgb8.tobytes())
        lc = line_config(stack, (W, H))
        im = image.ImageModem(STACKS[stack](lc))
        comp_img = im.modulate(img, 2)
        back = im.demodulate(comp_img, 2)
        save('imagefull_' + stack, rgb8=rgb8, comp8=numpy.frombuffer(comp_img.tobytes(), dtype=numpy.uint8).reshape(H, W),
             back8=numpy.frombuffer(back.tobytes(), dtype=numpy.uint8).reshape(H, W, 3), frame=numpy.array(2))


def full_frame_case():
    """BASELINE configs[2] at its FULL size as floats: one 720x480 frame of a valid NTSC signal through
    Simple3DCombModem(NtscCombModem) on the image.py row schedule.  The whole input is kept; of the reference's output the rows `rows`
    (the top and bottom eight and every 16th: the line geometry of the full height at a tenth of the bytes)."""
    W, H, frame = 720, 480, 3
    stack = 'ntsc_comb_3d'
    lc = line_config(stack, (W, H))
    rgb = test_picture(W, H).astype(numpy.float64).transpose(2, 0, 1) / 255.0
    comp = run_mod_frame(STACKS['ntsc'](lc), rgb, frame).astype(numpy.float32)
    out = run_demod_frame(STACKS[stack](lc), comp.astype(numpy.float64), frame)
    rows = numpy.array(sorted(set(list(range(8)) + list(range(0, H, 16)) + list(range(H - 8, H)))))
    save('framefull_demod_' + stack, inp=comp[None], out_rows=out[:, rows][None], rows=rows, frames=numpy.array([frame]), size=numpy.array([W, H]))


if __name__ == '__main__':
    if sys.argv[1:2] == ['full_images']:
        full_image_cases()
        full_frame_case()
        sys.exit(0)
    if sys.argv[1:2] == ['options']:     # only the option / variant cases (the rest is unchanged), optionally some
        option_cases(sys.argv[2:])
        sys.exit(0)
    if sys.argv[1:2] == ['widths']:
        width_cases()
        sys.exit(0)
    if sys.argv[1:2] == ['wrappers']:
        wrapper_cases(sys.argv[2:])
        sys.exit(0)
    if sys.argv[1:2] == ['variants']:
        variant_cases(sys.argv[2:])
        sys.exit(0)
    if sys.argv[1:2] == ['degenerate']:
        degenerate_cases()
        sys.exit(0)
    if sys.argv[1:2] == ['notch_shift']:
        notch_shift_cases()
        sys.exit(0)
    if sys.argv[1:2] == ['callables']:
        callable_cases(sys.argv[2:])
        sys.exit(0)
    make_plans()
    frame_cases()
    option_cases()
    width_cases()
    variant_cases()
    wrapper_cases()
    callable_cases()
    notch_shift_cases()
    degenerate_cases()
    row_cases()
    image_cases()
    full_image_cases()
    full_frame_case()
