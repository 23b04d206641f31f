# -*- coding: utf-8 -*-
"""Golden vectors of the NESTED stacks (round 6), produced by running the REFERENCE in the build container:

    python tests/golden/make_golden_nested.py      # writes tests/golden/nested_*.npz

The reference's wrappers sit on any backend with demodulate_components / modulate_components (comb.py:90-113, 131-155): a comb
wrapper around ColorAveragingModem (comb.py:105 folds the backend's modulation_delay into the strip line), a wrapper inside a
wrapper, wrappers around Pal3DModem(avg=f) and around the NIIR modems (cli.py:52-53).  Rounds 1 - 5 refused these; they now run level
by level (color_modem_amd/generic.py) and these sets pin both the oracle (oracle/cm_oracle_generic.py) and the device.

(The wrapper's notch= needs a backend with .config / .line_config - comb.py:18-20 - which a wrapper is not: the reference raises
AttributeError for notch= around another wrapper, so only the set around Pal3DModem carries one.)

Recorded as in make_golden.py: `inp` (float32, fed to the reference after a cast to float64), `out` (float64, what it returned) for whole
small frames through the row schedule of image.py:47-55, 75-83, explicit (frame, line) row sequences with a reset in the run, and one
uint8 round trip through the reference's own ImageModem.  The scipy.signal.iirdesign shim of make_golden.py applies (NTSC).
"""

import os
import sys

import numpy

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, HERE)

import make_golden as mg  # noqa: E402  (installs the shim, puts the reference on the path)
from make_golden import comb, line, ntsc, pal, secam, testing, save, run_demod_frame, run_mod_frame  # noqa: E402
from color_modem.color import niir  # noqa: E402

LS = line.LineStandard

# name -> (line standard, decoder / encoder under test, encoder that makes a valid input signal for a decoder)
STACKS = {
    'simple_avg_pals': ('GERBER_625', lambda lc: comb.SimpleCombModem(comb.ColorAveragingModem(pal.PalSModem(lc))), lambda lc: pal.PalSModem(lc)),
    'simple3d_avg_pald_minavg': ('GERBER_625', lambda lc: comb.Simple3DCombModem(comb.ColorAveragingModem(pal.PalDModem(lc)), avg=comb.minavg),
                                lambda lc: pal.PalSModem(lc)),
    'simple_simple_ntsc': ('NTSC_525', lambda lc: comb.SimpleCombModem(comb.SimpleCombModem(ntsc.NtscModem(lc))), lambda lc: ntsc.NtscModem(lc)),
    'simple3d_simple_ntsccomb': ('NTSC_525', lambda lc: comb.Simple3DCombModem(comb.SimpleCombModem(ntsc.NtscCombModem(lc)), avg=comb.minavg),
                                 lambda lc: ntsc.NtscModem(lc)),
    'simple3d_pal3d_favg': ('GERBER_625', lambda lc: comb.Simple3DCombModem(pal.Pal3DModem(lc, avg=mg.damped_avg), notch=4.0), lambda lc: pal.PalSModem(lc)),
    'avg_pal3d_favg': ('GERBER_625', lambda lc: comb.ColorAveragingModem(pal.Pal3DModem(lc, avg=mg.weighted_avg)), lambda lc: pal.PalSModem(lc)),
    'simple_niir_hue': ('GERBER_625', lambda lc: comb.SimpleCombModem(niir.HueCorrectingNiirModem(lc)), lambda lc: niir.HueCorrectingNiirModem(lc)),
    'simple3d_niir': ('GERBER_625', lambda lc: comb.Simple3DCombModem(niir.NiirModem(lc), avg=mg.weighted_avg), lambda lc: niir.NiirModem(lc)),
    'avg_avg_secam': ('GERBER_625', lambda lc: comb.ColorAveragingModem(comb.ColorAveragingModem(secam.SecamModem(lc))), None),
    'avg_niir': ('GERBER_625', lambda lc: comb.ColorAveragingModem(niir.NiirModem(lc)), None),
    'avg_avg_pals': ('GERBER_625', lambda lc: comb.ColorAveragingModem(comb.ColorAveragingModem(pal.PalSModem(lc))), None),
}


def lc_of(name, size):
    return line.LineConfig(size, getattr(LS, STACKS[name][0]))


def demod_cases():
    W, H = 720, 10
    for name, frames in (('simple_avg_pals', [1, 2]), ('simple3d_avg_pald_minavg', [0, 3]), ('simple_simple_ntsc', [0, 1]),
                         ('simple3d_simple_ntsccomb', [1, 2]), ('simple3d_pal3d_favg', [1, 2]), ('avg_pal3d_favg', [0, 3]),
                         ('simple_niir_hue', [1, 2]), ('simple3d_niir', [0, 3])):
        lc = lc_of(name, (W, H))
        enc = STACKS[name][2](lc)
        rgb = testing.synthetic_rgb(len(frames), H, W, seed=860)
        comp = numpy.stack([run_mod_frame(enc, rgb[i].astype(numpy.float64), f) for i, f in enumerate(frames)]).astype(numpy.float32)
        modem = STACKS[name][1](lc)
        out = numpy.stack([run_demod_frame(modem, comp[i].astype(numpy.float64), f) for i, f in enumerate(frames)])
        save('nested_demod_' + name, inp=comp, out=out, frames=numpy.array(frames), size=numpy.array([W, H]))


def mod_cases():
    W, H = 720, 10
    for name, frames in (('avg_avg_secam', [0, 1, 5]), ('avg_niir', [1, 2]), ('avg_avg_pals', [0, 3]), ('simple_avg_pals', [1, 2])):
        lc = lc_of(name, (W, H))
        modem = STACKS[name][1](lc)
        rgb = testing.synthetic_rgb(len(frames), H, W, seed=870)
        out = numpy.stack([run_mod_frame(modem, rgb[i].astype(numpy.float64), f) for i, f in enumerate(frames)])
        save('nested_mod_' + name, inp=rgb, out=out, frames=numpy.array(frames), size=numpy.array([W, H]))


def row_cases():
    """(frame, line) sequences at the full-height geometry fed to ONE modem object in order, with a reset in the run (a repeated line:
    the wrapper starts over while a stateful backend modulator sees its strip lines continue)."""
    seq = [(1, 0), (1, 2), (1, 4), (1, 6), (1, 6), (1, 8), (1, 10), (2, 571), (2, 573), (2, 575)]
    for name in ('simple_avg_pals', 'simple_simple_ntsc', 'simple_niir_hue', 'simple3d_pal3d_favg'):
        size = (720, 480) if STACKS[name][0] == 'NTSC_525' else (720, 576)
        lc = lc_of(name, size)
        enc, modem = STACKS[name][2](lc), STACKS[name][1](lc)
        rgb = testing.synthetic_rgb(1, len(seq), size[0], seed=880)[0]
        comp = numpy.stack([enc.modulate(f, y, *[rgb[c, i].astype(numpy.float64) for c in range(3)]) for i, (f, y) in enumerate(seq)]).astype(numpy.float32)
        out = numpy.stack([numpy.stack(modem.demodulate(f, y, comp[i].astype(numpy.float64))) for i, (f, y) in enumerate(seq)])
        comps = numpy.stack([numpy.stack(STACKS[name][1](lc).demodulate_components(f, y, comp[i].astype(numpy.float64), strip_chroma=False))
                             for i, (f, y) in enumerate(seq[:1])])
        save('nested_rows_' + name, inp=comp, out=out, first_unstripped=comps, seq=numpy.array(seq), size=numpy.array(size))


def image_case():
    """uint8 through the reference's own ImageModem (image.py:27-84), 720 x 12."""
    from PIL import Image
    from color_modem import image
    W, H = 720, 12
    rgb = testing.synthetic_rgb(1, H, W, seed=890)[0]
    rgb8 = numpy.uint8(numpy.rint(255.0 * rgb)).transpose(1, 2, 0).copy()
    img = Image.frombytes('RGB', (W, H), rgb8.tobytes())
    name = 'simple_avg_pals'
    im = image.ImageModem(STACKS[name][1](lc_of(name, (W, H))))
    comp_img = im.modulate(img, 1)
    back = im.demodulate(comp_img, 1)
    save('nested_image_' + name, rgb8=rgb8, comp8=numpy.frombuffer(comp_img.tobytes(), dtype=numpy.uint8).reshape(H, W),
         back8=numpy.frombuffer(back.tobytes(), dtype=numpy.uint8).reshape(H, W, 3), frame=numpy.array(1))


if __name__ == '__main__':
    demod_cases()
    mod_cases()
    row_cases()
    image_case()
