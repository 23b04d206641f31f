# -*- coding: utf-8 -*-
"""Golden vectors of the amplitude-modulated line-sequential standards (Proto-SECAM 1957 and
NIIR / SECAM-IV), produced by running the REFERENCE (color_modem/color/protosecam.py, niir.py, comb.py:130-167) in the
build container:

    python tests/golden/make_golden_am.py      # writes tests/golden/am_*.npz

Recorded: `inp` (float32, fed to the reference after a cast to float64) and `out` (float64, what it returned) for whole
small frames run through the row schedule of image.py:47-55, 75-83 (our float restatement of that loop, as in
make_golden.py), plus explicit (frame, line) row sequences at the full-height geometry.  NIIR runs with noise_level 0
(the reference's default; its noise is numpy.random and cannot be pinned).  numpy / scipy versions as in plans.json.
"""

import os
import sys
import warnings

import numpy

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
sys.path.insert(0, ROOT)
sys.path.insert(0, '/root/reference')
warnings.simplefilter('ignore')

from color_modem_amd import testing  # noqa: E402
from color_modem import comb, line  # noqa: E402
from color_modem.color import niir, pal, protosecam  # noqa: E402

LS = line.LineStandard

# name -> (line standard, factory)
STACKS = {
    'proto': ('FRENCH_819', lambda lc: protosecam.ProtoSecamModem(lc)),
    'proto_avg': ('FRENCH_819', lambda lc: comb.ColorAveragingModem(protosecam.ProtoSecamModem(lc))),
    'proto_nofilter': ('BELGIAN_819', lambda lc: protosecam.ProtoSecamModem(lc, premod_luma_filter=False)),
    'proto_625': ('GERBER_625', lambda lc: protosecam.ProtoSecamModem(lc)),
    'niir': ('GERBER_625', lambda lc: niir.NiirModem(lc)),
    'niir_hue': ('GERBER_625', lambda lc: niir.HueCorrectingNiirModem(lc)),
    'niir_525': ('NTSC_525', lambda lc: niir.NiirModem(lc)),     # PAL sub-carrier on 525 lines: a phase cycle of 4800 frames
}


def run_mod_frame(modem, rgb, frame):
    _, height, width = rgb.shape
    delay = getattr(modem, 'modulation_delay', 0)
    out = numpy.zeros((height, width))
    for field in range(2):
        for y in range(field, 2 * delay, 2):
            modem.modulate(frame, y, *[rgb[p, y].astype(numpy.float64) for p in range(3)])
        for y in range(field, height, 2):
            iy = y + 2 * delay
            while iy >= height:
                iy -= 2
            out[y] = modem.modulate(frame, y + 2 * delay, *[rgb[p, iy].astype(numpy.float64) for p in range(3)])
    return out


def run_demod_frame(modem, comp, frame):
    height, width = comp.shape
    out = numpy.zeros((3, height, width))
    for field in range(2):
        for y in range(field, height, 2):
            out[:, y] = numpy.stack(modem.demodulate(frame, y, comp[y].astype(numpy.float64)))
    return out


def save(name, **arrays):
    path = os.path.join(HERE, name + '.npz')
    numpy.savez_compressed(path, **arrays)
    print('%-28s %8.1f KB' % (name, os.path.getsize(path) / 1024.0))


def noise_cases():
    """NiirModem / HueCorrectingNiirModem with noise_level != 0 (niir.py:45-46, 193-194): the noise is numpy.random.random_sample,
    pinned here by numpy.random.seed(7) before each frame batch (the engine and the oracle draw in the same order)."""
    for name, make in (('niir_noise', lambda lc: niir.NiirModem(lc, noise_level=0.05)),
                       ('niir_hue_noise', lambda lc: niir.HueCorrectingNiirModem(lc, noise_level=0.08))):
        W, H, frames = 720, 7, [0, 3]
        lc = line.LineConfig((W, H), LS.GERBER_625)
        rgb = testing.synthetic_rgb(len(frames), H, W, seed=1600)
        outs = []
        for i, f in enumerate(frames):        # one seeded batch per frame: a test can replay any of them alone
            numpy.random.seed(7 + i)
            outs.append(run_mod_frame(make(lc), rgb[i], f))
        save('am_mod_' + name, inp=rgb, out=numpy.stack(outs), frames=numpy.array(frames), size=numpy.array([W, H]),
             standard=numpy.array('GERBER_625'), seeds=numpy.array([7, 8]))


def grey_cases():
    """NiirModem / HueCorrectingNiirModem on pictures with EXACTLY grey and nearly grey areas (byte / 255 levels, as ImageModem feeds them):
    there (db, dr) = niir.py:35-36 are rounding residues of ~1e-17 or small numbers, and the hue of the 0.1 pedestal (niir.py:42-49,
    187-198) is THEIR angle - the device evaluates these sums in float64 in the reference's own operation order where the saturation is
    small (cm_am_stages.h: niir_chroma_f64)."""
    W, H, frames = 720, 8, [0, 3]
    lc = line.LineConfig((W, H), LS.GERBER_625)
    rng = numpy.random.default_rng(1700)
    rgb8 = numpy.zeros((len(frames), 3, H, W), dtype=numpy.uint8)
    x = 0
    while x < W:                                   # runs of grey, nearly grey (+-1 .. 2 LSB per channel), coloured and black pixels
        n = int(rng.integers(4, 40))
        kind = int(rng.integers(4))
        v = rng.integers(0, 256, size=(len(frames), 1, H, 1))
        if kind == 0:
            blk = numpy.broadcast_to(v, (len(frames), 3, H, n)).copy()
        elif kind == 1:
            blk = numpy.clip(v + rng.integers(-2, 3, size=(len(frames), 3, H, n)), 0, 255)
        elif kind == 2:
            blk = rng.integers(0, 256, size=(len(frames), 3, H, n))
        else:
            blk = numpy.zeros((len(frames), 3, H, n), dtype=int)
        rgb8[:, :, :, x:x + n] = blk[:, :, :, :W - x]
        x += n
    rgb = (rgb8.astype(numpy.float64) / 255.0).astype(numpy.float32)
    for name, make in (('niir_grey', lambda lc: niir.NiirModem(lc)), ('niir_hue_grey', lambda lc: niir.HueCorrectingNiirModem(lc))):
        comp = numpy.stack([run_mod_frame(make(lc), rgb[i], f) for i, f in enumerate(frames)])
        save('am_mod_' + name, inp=rgb, out=comp, frames=numpy.array(frames), size=numpy.array([W, H]), standard=numpy.array('GERBER_625'))


def degenerate_cases():
    """Proto-SECAM / NIIR / hue-correcting NIIR on the degenerate inputs of make_golden.py: degenerate_pictures (black / white / grey / red pictures,
    all-zero / constant composites; 720 x 12, frame 1).  The NIIR decoder divides 0 / 0 on such composites: the reference's NaNs are recorded."""
    sys.path.insert(0, HERE)
    import make_golden
    W, H, frame = 720, 12, 1
    pics, pic_names, comps, comp_names = make_golden.degenerate_pictures(W, H)
    for name in ('proto', 'niir', 'niir_hue'):
        make = STACKS[name][1]
        lc = line.LineConfig((W, H), LS.GERBER_625)
        mod_out = numpy.stack([run_mod_frame(make(lc), pics[i], frame) for i in range(len(pics))])
        with numpy.errstate(all='ignore'):
            demod_out = numpy.stack([run_demod_frame(make(lc), comps[i], frame) for i in range(len(comps))])
        save('degenerate_am_' + name, pics=pics, pic_names=numpy.array(pic_names), mod_out=mod_out, comps=comps, comp_names=numpy.array(comp_names),
             demod_out=demod_out, frame=numpy.array(frame), size=numpy.array([W, H]))


def main():
    if len(sys.argv) > 1 and sys.argv[1] == 'grey':      # only the sets added last (the others stay as they were made)
        grey_cases()
        return
    cases = [  # (stack, decoder stack, width, height, frames)
        ('proto', 'proto', 720, 8, [0, 1]),
        ('proto_avg', 'proto', 720, 7, [1, 2]),
        ('proto_nofilter', 'proto_nofilter', 720, 6, [0]),
        ('proto_625', 'proto_625', 1024, 6, [3]),
        ('niir', 'niir', 720, 8, [0, 1, 2, 3]),
        ('niir_hue', 'niir_hue', 720, 7, [1, 2]),
        ('niir_525', 'niir_525', 768, 6, [0, 4799]),
    ]
    for stack, dec, W, H, frames in cases:
        std_name, make = STACKS[stack]
        lc = line.LineConfig((W, H), getattr(LS, std_name))
        rgb = testing.synthetic_rgb(len(frames), H, W, seed=1300 + W + H)
        comp = numpy.stack([run_mod_frame(make(lc), rgb[i], f) for i, f in enumerate(frames)])
        save('am_mod_' + stack, inp=rgb, out=comp, frames=numpy.array(frames), size=numpy.array([W, H]),
             standard=numpy.array(std_name))
        comp32 = comp.astype(numpy.float32)
        dmake = STACKS[dec][1]
        back = numpy.stack([run_demod_frame(dmake(lc), comp32[i], f) for i, f in enumerate(frames)])
        save('am_demod_' + stack, inp=comp32, out=back, frames=numpy.array(frames), size=numpy.array([W, H]),
             standard=numpy.array(std_name))
    # explicit (frame, line) sequences at the full-height geometry, with a break in the run and a frame change
    seqs = {
        'proto': ((720, 720), 'FRENCH_819', [(0, 0), (0, 2), (0, 4), (1, 715), (1, 717), (1, 719), (2, 1), (2, 3)]),
        'niir': ((720, 576), 'GERBER_625', [(1, 0), (1, 2), (1, 4), (3, 571), (3, 573), (3, 575), (2, 1), (2, 3)]),
    }
    for stack, (size, std_name, seq) in seqs.items():
        lc = line.LineConfig(size, getattr(LS, std_name))
        enc, modem = STACKS[stack][1](lc), STACKS[stack][1](lc)
        rgb = testing.synthetic_rgb(1, len(seq), size[0], seed=1400)[0]
        comp = numpy.stack([enc.modulate(f, y, *[rgb[c, i].astype(numpy.float64) for c in range(3)])
                            for i, (f, y) in enumerate(seq)]).astype(numpy.float32)
        out = numpy.stack([numpy.stack(modem.demodulate(f, y, comp[i].astype(numpy.float64))) for i, (f, y) in enumerate(seq)])
        save('am_rows_' + stack, inp=comp, out=out, seq=numpy.array(seq), size=numpy.array(size), standard=numpy.array(std_name))
    noise_cases()
    grey_cases()
    # NIIR component protocol with the chroma left in the luma (strip_chroma=False) and noise input
    lc = line.LineConfig((720, 6), LS.GERBER_625)
    noise = testing.synthetic_composite(2, 6, 720, seed=1500)
    m = niir.NiirModem(lc)
    rows = []
    for i, f in enumerate((0, 3)):
        for field in range(2):
            for y in range(field, 6, 2):
                rows.append(numpy.stack(m.demodulate_components(f, y, noise[i, y].astype(numpy.float64), strip_chroma=False)))
    save('am_niir_components_noise', inp=noise, out=numpy.stack(rows), frames=numpy.array([0, 3]), size=numpy.array([720, 6]),
         standard=numpy.array('GERBER_625'))


if __name__ == '__main__':
    if sys.argv[1:2] == ['degenerate']:
        degenerate_cases()
    elif sys.argv[1:2] == ['noise']:
        noise_cases()
    else:
        main()
