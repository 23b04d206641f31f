"""Per-row error of Pal3D wrappers on another PAL variant: python tools/diag_wrap2.py PAL_M NTSC_525 720 14 1900 which(0..3)"""
import sys, numpy, torch
sys.path.insert(0, '.'); sys.path.insert(0, 'tests')
import stacks
from color_modem_amd import image, testing, comb, line
from color_modem_amd.color import pal
from oracle import cm_oracle
variant, std, w, h, frames, which = sys.argv[1], sys.argv[2], int(sys.argv[3]), int(sys.argv[4]), int(sys.argv[5]), int(sys.argv[6])
lc = line.LineConfig((w, h), getattr(line.LineStandard, std))
v = getattr(pal.PalVariant, variant)
make = [lambda: comb.Simple3DCombModem(pal.PalDModem(lc, v)), lambda: comb.SimpleCombModem(pal.PalDModem(lc, v), avg=comb.minavg),
        lambda: comb.Simple3DCombModem(pal.Pal3DModem(lc, v)), lambda: comb.SimpleCombModem(pal.Pal3DModem(lc, v, avg=comb.minavg), avg=comb.minavg)][which]
modem = make()
eng = image.ImageModem(modem)._engine()
print(eng.describe()[:200])
few = testing.synthetic_rgb(2, h, w, seed=7 + h)
comp2 = cm_oracle.modulate_frames_f32(pal.PalSModem(lc, v), few, first_frame=0, n_threads=4)
comp = torch.from_numpy(comp2).cuda().repeat((frames + 1) // 2, 1, 1)[:frames].contiguous()
for first in (0, 4797):
    got = eng.demodulate_frames(comp, first_frame=first)
    for i in (0, 1, 5, frames - 1):
        want = cm_oracle.demodulate_frames_f32(modem, comp2[i % 2][None], first_frame=first + i, n_threads=4)[0]
        g = got[i].cpu().numpy().astype(numpy.float64)
        errp = numpy.abs(g - want).max(axis=2) / numpy.abs(want).max()
        print('first', first, 'frame', i, 'rows > 1e-5:', [(r, ['%.1g' % errp[p, r] for p in range(3)]) for r in range(h) if errp[:, r].max() > 1e-5])
