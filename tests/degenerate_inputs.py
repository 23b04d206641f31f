"""Degenerate inputs against vectors the REFERENCE produced (tests/golden/degenerate_*.npz, made by make_golden*.py degenerate): black / white /
grey / saturated pictures through every encoder, all-zero / constant composite rows through every decoder - the places where an algorithm
divides by an amplitude or takes the angle of a vanishing pair.  `run('device')`: the HIP path; `run('oracle')`: the float64 oracle (which takes
its coefficients from the product's host classes: these vectors are what pins it there).  TEST TOOL.  python tests/degenerate_inputs.py [oracle]"""
import os, sys, warnings, numpy
sys.path.insert(0, '.'); sys.path.insert(0, 'tests')
warnings.filterwarnings('ignore')
import stacks, am_stacks
from color_modem_amd import line

QAM = ('pal_s', 'pal_d', 'pal_3d', 'ntsc', 'ntsc_comb_3d', 'secam', 'secam_avg', 'simple3d_pald')
AM = ('proto', 'niir', 'niir_hue')


def err(a, b):
    """max |a - b| / max(1, max |b|), NaN = NaN (the reference's own 0 / 0), anything else against a NaN = inf"""
    a, b = numpy.asarray(a, dtype=numpy.float64), numpy.asarray(b, dtype=numpy.float64)
    both_nan = numpy.isnan(a) & numpy.isnan(b)
    d = numpy.where(both_nan, 0.0, numpy.abs(a - b))
    return float(numpy.nan_to_num(d, nan=numpy.inf).max() / max(1.0, numpy.nanmax(numpy.abs(b)) if numpy.isfinite(b).any() else 1.0))


def load(name):
    return numpy.load(os.path.join(stacks.GOLDEN, name + '.npz'))


def run(what='device'):
    rows = []
    device = what == 'device'
    if device:
        from color_modem_amd import image
    from oracle import cm_oracle, cm_oracle_am as oa, cm_oracle_mac as om
    for name in QAM:
        g = load('degenerate_' + name)
        W, H = [int(v) for v in g['size']]
        frame = int(g['frame'])
        modem = stacks.make(name, (W, H))
        for i, tag in enumerate(g['pic_names']):
            rgb = g['pics'][i:i + 1]
            got = image.ImageModem(modem).modulate_frames(rgb, first_frame=frame) if device else cm_oracle.OracleModem(modem).modulate_frame(frame, rgb[0].astype(numpy.float64))[None]
            rows.append((name, 'encode', str(tag), err(got[0], g['mod_out'][i]), ''))
        for i, tag in enumerate(g['comp_names']):
            comp = g['comps'][i:i + 1]
            got = image.ImageModem(modem).demodulate_frames(comp, first_frame=frame) if device else cm_oracle.OracleModem(modem).demodulate_frame(frame, comp[0].astype(numpy.float64))[None]
            rows.append((name, 'decode', str(tag), err(got[0], g['demod_out'][i]), ''))
    for name in AM:
        g = load('degenerate_am_' + name)
        W, H = [int(v) for v in g['size']]
        frame = int(g['frame'])
        modem = am_stacks.STACKS[name](line.LineConfig((W, H), line.LineStandard.GERBER_625))
        for i, tag in enumerate(g['pic_names']):
            rgb = g['pics'][i:i + 1]
            got = image.ImageModem(modem).modulate_frames(rgb, first_frame=frame) if device else oa.modulate_frames(modem, rgb.astype(numpy.float64), frame)
            rows.append((name, 'encode', str(tag), err(got[0], g['mod_out'][i]), ''))
        for i, tag in enumerate(g['comp_names']):
            comp = g['comps'][i:i + 1]
            got = image.ImageModem(modem).demodulate_frames(comp, first_frame=frame) if device else oa.demodulate_frames(modem, comp.astype(numpy.float64), frame)
            want = g['demod_out'][i]
            note = ' (reference: %d NaN samples, here: %d)' % (numpy.isnan(want).sum(), numpy.isnan(numpy.asarray(got[0], dtype=numpy.float64)).sum())
            rows.append((name, 'decode', str(tag), err(got[0], want), note))
    g = load('degenerate_mac')
    W, H = [int(v) for v in g['size']]
    frame = int(g['frame'])
    lc = line.LineConfig((W, H))
    from color_modem_amd.color import mac
    for i, tag in enumerate(g['pic_names']):
        rgb = g['pics'][i:i + 1]
        got = image.ImageModem(mac.MacModem(lc)).modulate_frames(rgb, first_frame=frame) if device else om.modulate_frames(lc, rgb.astype(numpy.float64), frame, False, 1080)
        rows.append(('mac', 'encode', str(tag), err(got[0], g['mod_out'][i]), ''))
    for i, tag in enumerate(g['comp_names']):
        comp = g['comps'][i:i + 1]
        got = image.ImageModem(mac.MacModem(lc)).demodulate_frames(comp, first_frame=frame) if device else om.demodulate_frames(lc, comp.astype(numpy.float64), frame)
        rows.append(('mac', 'decode', str(tag), err(got[0], g['demod_out'][i]), ''))
    return rows


# no sub-carrier: behind its band-pass the SECAM discriminator takes the angle of the filters' decaying rounding residues - the reference's float64
# ones, another restatement's other ones: neither is a signal, no implementation but the reference's own reproduces them
KNOWN = {('secam', 'decode', 'constant 0.3'), ('secam_avg', 'decode', 'constant 0.3')}

if __name__ == '__main__':
    what = sys.argv[1] if len(sys.argv) > 1 else 'device'
    bad = 0
    for name, direction, tag, e, note in run(what):
        fail = e >= (1e-5 if what == 'device' else 1e-9)
        bad += fail and (name, direction, tag) not in KNOWN
        print('%-14s %s %-13s %.2e%s%s' % (name, direction, tag, e, note, '   <-- FAIL' if fail else ''))
    print('failures beyond the known case', bad)
    sys.exit(1 if bad else 0)
