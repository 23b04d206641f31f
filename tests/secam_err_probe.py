"""Where the SECAM decoder's float32 error sits: python tests/secam_err_probe.py [widths] [variants] [seed] (GPU; uses the oracle: test tool)"""
import sys, warnings
import numpy
sys.path.insert(0, '.'); sys.path.insert(0, 'tests')
warnings.filterwarnings('ignore')
from color_modem_amd import image, line, testing
from color_modem_amd.color import secam
from oracle import cm_oracle
WIDTHS = [int(a) for a in sys.argv[1].split(',')] if len(sys.argv) > 1 else (720, 1280, 1920)
VARIANTS = sys.argv[2].split(',') if len(sys.argv) > 2 else ('SECAM', 'SECAM_II', 'SECAM_M')
SEED = int(sys.argv[3]) if len(sys.argv) > 3 else 5
for w in WIDTHS:
    for vn in VARIANTS:
        lc = line.LineConfig((w, 120), line.LineStandard.detect(576))
        modem = secam.SecamModem(lc, getattr(secam.SecamVariant, vn))
        im = image.ImageModem(modem)
        rgb = testing.synthetic_rgb(2, 120, w, seed=SEED)
        comp = cm_oracle.modulate_frames_f32(modem, rgb, first_frame=3, n_threads=8)
        got = im.demodulate_frames(comp, first_frame=3).astype(numpy.float64)
        want = cm_oracle.demodulate_frames_f32(modem, comp, first_frame=3, n_threads=8)
        err = numpy.abs(got - want) / numpy.abs(want).max()
        f, p, y, x = numpy.unravel_index(err.argmax(), err.shape)
        col = err.max(axis=(0, 1, 2))
        print('%4d %-9s max %.2e at frame %d plane %d row %d col %d | max over cols 0-15 %.1e, 16..W-17 %.1e, last 16 %.1e | 99.9th pct %.1e'
              % (w, vn, err.max(), f, p, y, x, col[:16].max(), col[16:-16].max(), col[-16:].max(), numpy.quantile(err, 0.999)))
