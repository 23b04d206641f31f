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


def nested():
    """Round 6: every wrapper around every backend kind, two levels deep - what the reference's classes accept (comb.py:71-167 sit on any
    backend with demodulate_components / modulate_components) must run here; where the REFERENCE itself raises (a comb wrapper needs
    demodulate_components: SecamModem, ProtoSecamModem and MacModem have none - AttributeError at the first row) the same refusal is
    expected.  One line per stack: ok / the error."""
    from color_modem_amd.color import mac, niir, protosecam
    lc, ln = line.LineConfig((720, 20), line.LineStandard.GERBER_625), line.LineConfig((720, 20), line.LineStandard.NTSC_525)
    leaves = [('PalS', lc, lambda c: pal.PalSModem(c), True), ('PalD', lc, lambda c: pal.PalDModem(c), True), ('Pal3D', lc, lambda c: pal.Pal3DModem(c), True),
              ('Pal3D(avg=f)', lc, lambda c: pal.Pal3DModem(c, avg=lambda a, b: 0.25 * a + 0.75 * b), True),
              ('Ntsc', ln, lambda c: ntsc.NtscModem(c), True), ('NtscComb', ln, lambda c: ntsc.NtscCombModem(c), True),
              ('Niir', lc, lambda c: niir.NiirModem(c), True), ('HueCorrectingNiir', lc, lambda c: niir.HueCorrectingNiirModem(c), True),
              ('Secam', lc, lambda c: secam.SecamModem(c), False), ('ProtoSecam', lc, lambda c: protosecam.ProtoSecamModem(c), False),
              ('Mac', lc, lambda c: mac.MacModem(c), False)]
    wrappers = [('Simple', lambda m: comb.SimpleCombModem(m)), ('Simple3D', lambda m: comb.Simple3DCombModem(m)),
                ('Simple(minavg)', lambda m: comb.SimpleCombModem(m, avg=comb.minavg)), ('Avg', lambda m: comb.ColorAveragingModem(m))]
    rows = bad = 0
    for lname, cfg, leaf, has_components in leaves:
        for n1, w1 in wrappers:
            for n2, w2 in [('', None)] + wrappers:
                name = ('%s(%s(%s))' % (n2, n1, lname)) if w2 else '%s(%s)' % (n1, lname)
                modem = w1(leaf(cfg))
                if w2:
                    modem = w2(modem)
                # the reference raises AttributeError where a comb wrapper meets a backend without demodulate_components
                comb_on_top = ('Simple' in n1) or ('Simple' in n2)
                expect_refusal = comb_on_top and not has_components
                res = try_pair(lambda c, v: modem, None, cfg.size if hasattr(cfg, 'size') else (720, 20))
                ok = ('ERROR' not in res and 'construct' not in res and 'demod:' not in res) if not expect_refusal else ('AttributeError' in res or 'demod' in res or 'construct' in res)
                rows += 1
                bad += 0 if ok else 1
                print('%-44s %s%s' % (name, res[:110], '' if ok else '    <-- UNEXPECTED'))
                sys.stdout.flush()
    print('%d nested stacks, %d unexpected results' % (rows, bad))


def main():
    if sys.argv[1:2] == ['widths']:
        return widths()
    if sys.argv[1:2] == ['nested']:
        return nested()
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
