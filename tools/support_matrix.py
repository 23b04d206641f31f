# -*- coding: utf-8 -*-
"""Which (modem stack x colour-system variant) pairs have a kernel instance in this build.

    python tools/support_matrix.py            # needs a GPU: creates every plan and runs one tiny batch

For each pair: 'ok' (plan created, 1 frame modulated + demodulated, finite) or the error the library gives.
"""
import os
import sys

import numpy

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)

from color_modem_amd import comb, image, line, testing  # noqa: E402
from color_modem_amd.color import ntsc, pal, secam  # noqa: E402


def variants(cls):
    return sorted(k for k, v in vars(cls).items() if isinstance(v, cls))


def stacks_for(system):
    if system == 'pal':
        return [('PalS', lambda lc, v: pal.PalSModem(lc, v)),
                ('PalD', lambda lc, v: pal.PalDModem(lc, v)),
                ('Pal3D', lambda lc, v: pal.Pal3DModem(lc, v)),
                ('Simple(PalS)', lambda lc, v: comb.SimpleCombModem(pal.PalSModem(lc, v))),
                ('Simple3D(PalD)', lambda lc, v: comb.Simple3DCombModem(pal.PalDModem(lc, v))),
                ('Simple(Pal3D)', lambda lc, v: comb.SimpleCombModem(pal.Pal3DModem(lc, v))),
                ('Avg(PalS)', lambda lc, v: comb.ColorAveragingModem(pal.PalSModem(lc, v)))]
    if system == 'ntsc':
        return [('Ntsc', lambda lc, v: ntsc.NtscModem(lc, v)),
                ('NtscComb', lambda lc, v: ntsc.NtscCombModem(lc, v)),
                ('Simple(Ntsc)', lambda lc, v: comb.SimpleCombModem(ntsc.NtscModem(lc, v))),
                ('Simple3D(NtscComb)', lambda lc, v: comb.Simple3DCombModem(ntsc.NtscCombModem(lc, v))),
                ('Avg(Ntsc)', lambda lc, v: comb.ColorAveragingModem(ntsc.NtscModem(lc, v)))]
    return [('Secam', lambda lc, v: secam.SecamModem(lc, v)),
            ('Avg(Secam)', lambda lc, v: comb.ColorAveragingModem(secam.SecamModem(lc, v)))]


def try_pair(make, variant, size):
    try:
        modem = make(line.LineConfig(size), variant)
        im = image.ImageModem(modem)
        rgb = testing.synthetic_rgb(1, size[1], size[0], seed=5)
        msg = []
        try:
            comp = im.modulate_frames(rgb, first_frame=1)
            msg.append('mod ok' if numpy.isfinite(comp).all() else 'mod NaN')
        except NotImplementedError as e:
            msg.append('mod: %s' % e)
            comp = testing.synthetic_composite(1, size[1], size[0], seed=5)
        try:
            out = im.demodulate_frames(comp, first_frame=1)
            msg.append('demod ok' if numpy.isfinite(out).all() else 'demod NaN')
        except NotImplementedError as e:
            msg.append('demod: %s' % e)
        return '; '.join(msg)
    except NotImplementedError as e:
        return 'construct: %s' % e
    except Exception as e:  # noqa
        return 'ERROR %s: %s' % (type(e).__name__, e)


def widths():
    """The default variant of each system at other image widths (= sampling rates): the run-time-shape instances."""
    for system, v in (('pal', pal.PalVariant.PAL), ('ntsc', ntsc.NtscVariant.NTSC), ('secam', secam.SecamVariant.SECAM)):
        for w in (480, 544, 640, 704, 768, 960, 1024, 1280, 1440, 1920):
            size = (w, 480 if system == 'ntsc' else 576)
            for sname, make in stacks_for(system):
                print('%-8s %-10s %-9s %-20s %s' % (system, 'default', '%dx%d' % size, sname, try_pair(make, v, size)))
                sys.stdout.flush()


def main():
    if sys.argv[1:2] == ['widths']:
        return widths()
    for system, cls in (('pal', pal.PalVariant), ('ntsc', ntsc.NtscVariant), ('secam', secam.SecamVariant)):
        for vname in variants(cls):
            v = getattr(cls, vname)
            for size in ((720, 576), (720, 480)):
                for sname, make in stacks_for(system):
                    print('%-8s %-10s %-9s %-20s %s' % (system, vname, '%dx%d' % size, sname, try_pair(make, v, size)))
                    sys.stdout.flush()
    # the amplitude-modulated line-sequential standards and D2-MAC (one variant each)
    from color_modem_amd.color import mac, niir, protosecam
    extra = [('ProtoSecam', (720, 736), lambda lc, v: protosecam.ProtoSecamModem(lc)),
             ('Avg(ProtoSecam)', (720, 736), lambda lc, v: comb.ColorAveragingModem(protosecam.ProtoSecamModem(lc))),
             ('Niir', (720, 576), lambda lc, v: niir.NiirModem(lc)),
             ('HueCorrectingNiir', (720, 576), lambda lc, v: niir.HueCorrectingNiirModem(lc)),
             ('Mac', (720, 576), lambda lc, v: mac.MacModem(lc)),
             ('Avg(Mac)', (720, 576), lambda lc, v: comb.ColorAveragingModem(mac.MacModem(lc)))]
    for sname, size, make in extra:
        print('%-8s %-10s %-9s %-20s %s' % ('other', 'default', '%dx%d' % size, sname, try_pair(make, None, size)))
        sys.stdout.flush()


if __name__ == '__main__':
    main()
