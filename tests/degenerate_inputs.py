"""Degenerate inputs against the float64 oracle (TEST TOOL, uses oracle/): all-zero / constant / grey pictures through every encoder, all-zero /
constant composite rows through every decoder - the places where an algorithm divides by an amplitude or takes the angle of a vanishing pair.
python tests/degenerate_inputs.py"""
import sys, warnings, numpy
sys.path.insert(0, '.'); sys.path.insert(0, 'tests')
warnings.filterwarnings('ignore')
import stacks, am_stacks
from color_modem_amd import image, line
from color_modem_amd.color import mac
from oracle import cm_oracle, cm_oracle_am as oa, cm_oracle_mac as om

def run():
    W, H = 720, 12
    def err(a, b):
        a, b = numpy.asarray(a, dtype=numpy.float64), numpy.asarray(b, dtype=numpy.float64)
        both_nan = numpy.isnan(a) & numpy.isnan(b)
        d = numpy.where(both_nan, 0.0, numpy.abs(a - b))
        return float(numpy.nan_to_num(d, nan=numpy.inf).max() / max(1.0, numpy.nanmax(numpy.abs(b)) if numpy.isfinite(b).any() else 1.0))
    pics = {'black': numpy.zeros((1, 3, H, W), numpy.float32), 'white': numpy.ones((1, 3, H, W), numpy.float32),
            'grey 200/255': numpy.full((1, 3, H, W), numpy.float32(200 / 255.0)), 'red': numpy.stack([numpy.ones((1, H, W), numpy.float32), numpy.zeros((1, H, W), numpy.float32), numpy.zeros((1, H, W), numpy.float32)], axis=1)}
    comps = {'zero': numpy.zeros((1, H, W), numpy.float32), 'constant 0.3': numpy.full((1, H, W), numpy.float32(0.3))}
    rows = []
    for name in ('pal_s', 'pal_d', 'pal_3d', 'ntsc', 'ntsc_comb_3d', 'secam', 'secam_avg', 'simple3d_pald'):
        modem = stacks.make(name, (W, H))
        im = image.ImageModem(modem)
        for tag, rgb in pics.items():
            e = err(im.modulate_frames(rgb, first_frame=1), cm_oracle.modulate_frames_f32(modem, rgb, first_frame=1, n_threads=4))
            rows.append((name, 'encode', tag, e, ''))
        for tag, comp in comps.items():
            e = err(im.demodulate_frames(comp, first_frame=1), cm_oracle.demodulate_frames_f32(modem, comp, first_frame=1, n_threads=4))
            rows.append((name, 'decode', tag, e, ''))
    for name in ('proto', 'niir', 'niir_hue'):
        lc = line.LineConfig((W, H), line.LineStandard.GERBER_625)
        modem = am_stacks.STACKS[name](lc)
        im = image.ImageModem(modem)
        for tag, rgb in pics.items():
            e = err(im.modulate_frames(rgb, first_frame=1), oa.modulate_frames(modem, rgb.astype(numpy.float64), 1))
            rows.append((name, 'encode', tag, e, ''))
        for tag, comp in comps.items():
            got, want = im.demodulate_frames(comp, first_frame=1), oa.demodulate_frames(modem, comp.astype(numpy.float64), 1)
            e = err(got, want)
            note = ' (oracle: %d NaN samples, device: %d)' % (numpy.isnan(want).sum(), numpy.isnan(got).sum())
            rows.append((name, 'decode', tag, e, note))
    lc = line.LineConfig((W, H))
    for avg in (False,):
        enc = mac.MacModem(lc)
        im = image.ImageModem(enc)
        for tag, rgb in pics.items():
            e = err(im.modulate_frames(rgb, first_frame=1), om.modulate_frames(lc, rgb.astype(numpy.float64), 1, avg, 1080))
            rows.append(('mac', 'encode', tag, e, ''))
    return rows


KNOWN = {('secam', 'decode', 'constant 0.3'), ('secam_avg', 'decode', 'constant 0.3')}      # no sub-carrier: the angle of rounding residues

if __name__ == '__main__':
    bad = 0
    for name, direction, tag, e, note in run():
        fail = e >= 1e-5
        bad += fail and (name, direction, tag) not in KNOWN
        print('%-14s %s %-13s %.2e%s%s' % (name, direction, tag, e, note, '   <-- FAIL' if fail else ''))
    print('failures beyond the known case', bad)
    sys.exit(1 if bad else 0)
