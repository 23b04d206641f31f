# -*- coding: utf-8 -*-
"""Golden vectors of the D2-MAC style time-multiplex path, produced by running the REFERENCE
(color_modem/color/mac.py, comb.py:130-167, image.py) in the build container:

    python tests/golden/make_golden_mac.py      # writes tests/golden/mac_*.npz

Recorded: `inp` (float32, fed to the reference after a cast to float64) and `out` (float64, what it returned) for whole
small frames run through the row schedule of image.py:47-55, 75-83 (our float restatement of that loop, as in
make_golden.py, here with the row length changing between the two sides: 720 <-> 1080), and one uint8 round trip through
the reference's own ImageModem.  numpy / scipy versions as in plans.json.
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
from color_modem import comb, image, line  # noqa: E402
from color_modem.color import mac  # noqa: E402


def run_mod_frame(modem, rgb, frame):
    _, height, _ = rgb.shape
    delay = getattr(modem, 'modulation_delay', 0)
    rows = [None] * height
    for field in range(2):
        for y in range(field, 2 * delay, 2):
            modem.modulate(frame, y, *[rgb[p, y].astype(numpy.float64) for p in range(3)])
        for y in range(field, height, 2):
            iy = y + 2 * delay
            while iy >= height:
                iy -= 2
            rows[y] = modem.modulate(frame, y + 2 * delay, *[rgb[p, iy].astype(numpy.float64) for p in range(3)])
    return numpy.stack(rows)


def run_demod_frame(modem, comp, frame):
    height = comp.shape[0]
    rows = [None] * height
    for field in range(2):
        for y in range(field, height, 2):
            rows[y] = numpy.stack(modem.demodulate(frame, y, comp[y].astype(numpy.float64)))
    return numpy.stack(rows).transpose(1, 0, 2)   # [3][H][720]


def save(name, **arrays):
    path = os.path.join(HERE, name + '.npz')
    numpy.savez_compressed(path, **arrays)
    print('%-20s %8.1f KB' % (name, os.path.getsize(path) / 1024.0))


def main():
    W = 720
    std = line.LineStandard.GERBER_625
    for name, H, frames, make in (
            ('mac', 12, [0, 1], lambda lc: mac.MacModem(lc)),
            ('mac_avg', 12, [0, 3], lambda lc: comb.ColorAveragingModem(mac.MacModem(lc))),
            ('mac_avg_h7', 7, [1], lambda lc: comb.ColorAveragingModem(mac.MacModem(lc)))):
        lc = line.LineConfig((W, H), std)
        rgb = testing.synthetic_rgb(len(frames), H, W, seed=900 + H)
        out = numpy.stack([run_mod_frame(make(lc), rgb[i], f) for i, f in enumerate(frames)])
        save('mac_mod_' + name.replace('mac_', '').replace('mac', 'plain'), inp=rgb, out=out, frames=numpy.array(frames),
             height=numpy.array(H))
        if name == 'mac':
            comp = out.astype(numpy.float32)
            back = numpy.stack([run_demod_frame(make(lc), comp[i], f) for i, f in enumerate(frames)])
            save('mac_demod_plain', inp=comp, out=back, frames=numpy.array(frames), height=numpy.array(H))
    # the cases that resample (mac.py:49-55, 71-74, 88-91): other row lengths, the 720-sample D2MAC_7MHZ line, an odd width
    for name, Wi, cw, H, frames, avg in (('w768_7mhz', 768, mac.MacVariant.D2MAC_7MHZ, 8, [0, 1], False),
                                         ('w768_7mhz_avg', 768, mac.MacVariant.D2MAC_7MHZ, 7, [2], True),
                                         ('w640_12mhz', 640, mac.MacVariant.D2MAC_12MHZ, 6, [1], False),
                                         ('w720_7mhz', 720, mac.MacVariant.D2MAC_7MHZ, 6, [0], False),
                                         ('w1000_900', 1000, 900, 5, [3], True)):
        lc = line.LineConfig((Wi, H), std)
        mk = (lambda: comb.ColorAveragingModem(mac.MacModem(lc, cw))) if avg else (lambda: mac.MacModem(lc, cw))
        rgb = testing.synthetic_rgb(len(frames), H, Wi, seed=1200 + Wi)
        out = numpy.stack([run_mod_frame(mk(), rgb[i], f) for i, f in enumerate(frames)])
        comp = out.astype(numpy.float32)
        back = numpy.stack([run_demod_frame(mac.MacModem(lc, cw), comp[i], f) for i, f in enumerate(frames)])
        save('mac_resampled_' + name, rgb=rgb, comp=out, back=back, frames=numpy.array(frames), height=numpy.array(H),
             width=numpy.array(Wi), line_width=numpy.array(out.shape[2]), averaging=numpy.array(1 if avg else 0))
    # not a valid MAC signal: noise rows (every sample of the row, incl. the guard areas the encoder leaves at 0.5)
    H = 9
    lc = line.LineConfig((W, H), std)
    comp = testing.synthetic_composite(2, H, 1080, seed=77)
    back = numpy.stack([run_demod_frame(mac.MacModem(lc), comp[i], f) for i, f in enumerate([2, 5])])
    save('mac_demod_noise', inp=comp, out=back, frames=numpy.array([2, 5]), height=numpy.array(H))
    # uint8 through the reference's own ImageModem
    from PIL import Image
    H = 6
    lc = line.LineConfig((W, H), std)
    rgb = testing.synthetic_rgb(1, H, W, seed=501)[0]
    rgb8 = numpy.uint8(numpy.rint(255.0 * rgb)).transpose(1, 2, 0).copy()
    img = Image.frombytes('RGB', (W, H), rgb8.tobytes())
    im = image.ImageModem(comb.ColorAveragingModem(mac.MacModem(lc)))
    comp_img = im.modulate(img, 1)
    back = im.demodulate(comp_img, 1)
    assert comp_img.size == (1080, H) and back.size == (720, H)
    save('mac_image_avg', rgb8=rgb8, comp8=numpy.frombuffer(comp_img.tobytes(), dtype=numpy.uint8).reshape(H, 1080),
         back8=numpy.frombuffer(back.tobytes(), dtype=numpy.uint8).reshape(H, 720, 3), frame=numpy.array(1))


def degenerate_cases():
    """MacModem on the degenerate pictures of make_golden.py: degenerate_pictures and on all-zero / constant lines (720 x 12, frame 1)."""
    sys.path.insert(0, HERE)
    import make_golden
    W, H, frame = 720, 12, 1
    pics, pic_names, _, comp_names = make_golden.degenerate_pictures(W, H)
    lc = line.LineConfig((W, H))
    mod_out = numpy.stack([run_mod_frame(mac.MacModem(lc), pics[i], frame) for i in range(len(pics))])
    comps = numpy.stack([numpy.zeros((H, 1080), numpy.float32), numpy.full((H, 1080), numpy.float32(0.3))])
    demod_out = numpy.stack([run_demod_frame(mac.MacModem(lc), comps[i], frame) for i in range(len(comps))])
    save('degenerate_mac', pics=pics, pic_names=numpy.array(pic_names), mod_out=mod_out, comps=comps, comp_names=numpy.array(comp_names),
         demod_out=demod_out, frame=numpy.array(frame), size=numpy.array([W, H]))


if __name__ == '__main__':
    if sys.argv[1:2] == ['degenerate']:
        degenerate_cases()
    else:
        main()
