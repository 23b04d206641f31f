# -*- coding: utf-8 -*-
"""Encode a picture to composite video and decode it again - the reference's README example with the imports switched.

    python examples/roundtrip.py [in.png [out.png]]      (needs an MI355X; without arguments a test card is generated)

Reference:                                   This package:
    from color_modem.line import LineConfig      from color_modem_amd.line import LineConfig
    from color_modem.color.pal import PalDModem  from color_modem_amd.color.pal import PalDModem
    from color_modem.image import ImageModem     from color_modem_amd.image import ImageModem
"""
import os
import sys

import numpy
from PIL import Image

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))

from color_modem_amd.color.mac import MacModem  # noqa: E402
from color_modem_amd.color.pal import PalDModem, PalSModem  # noqa: E402
from color_modem_amd.color.secam import SecamModem  # noqa: E402
from color_modem_amd.comb import ColorAveragingModem  # noqa: E402
from color_modem_amd.image import ImageModem  # noqa: E402
from color_modem_amd.line import LineConfig  # noqa: E402


def test_card(width=720, height=576):
    x = numpy.linspace(0.0, 1.0, width)[None, :]
    y = numpy.linspace(0.0, 1.0, height)[:, None]
    bars = numpy.array([[1, 1, 1], [1, 1, 0], [0, 1, 1], [0, 1, 0], [1, 0, 1], [1, 0, 0], [0, 0, 1], [0, 0, 0]], dtype=float)
    img = bars[numpy.minimum((x * 8).astype(int), 7)[0]][None, :, :] * (0.25 + 0.75 * (1.0 - y))[:, :, None]
    return Image.fromarray(numpy.uint8(numpy.rint(255 * img)), 'RGB')


def main():
    img = Image.open(sys.argv[1]).convert('RGB') if len(sys.argv) > 1 else test_card()
    line_config = LineConfig(img.size)
    composite = ImageModem(PalSModem(line_config)).modulate(img, frame=0)        # mode 'L'
    decoded = ImageModem(PalDModem(line_config)).demodulate(composite, frame=0)  # mode 'RGB', 2D comb
    err = numpy.abs(numpy.asarray(decoded, dtype=float) - numpy.asarray(img, dtype=float)).mean()
    print('%dx%d: composite %s, decoded %s, mean |decoded - original| = %.2f of 255' % (img.size + (composite.mode, decoded.mode, err)))
    if len(sys.argv) > 2:
        decoded.save(sys.argv[2])
    # the reference's cli.py default (SECAM with encoder-side averaging) and the D2-MAC style time multiplex
    for name, enc, dec in (('SECAM', ColorAveragingModem(SecamModem(line_config)), SecamModem(line_config)),
                           ('MAC', ColorAveragingModem(MacModem(line_config)), MacModem(line_config))):
        composite = ImageModem(enc).modulate(img, frame=0)
        decoded = ImageModem(dec).demodulate(composite, frame=0)
        ref = numpy.asarray(img.resize(decoded.size) if decoded.size != img.size else img, dtype=float)
        err = numpy.abs(numpy.asarray(decoded, dtype=float) - ref).mean()
        print('%-5s composite %dx%d, decoded %dx%d, mean |decoded - original| = %.2f of 255' % ((name,) + composite.size + decoded.size + (err,)))


if __name__ == '__main__':
    main()
