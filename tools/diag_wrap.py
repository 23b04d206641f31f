"""Per-row error of a wrapped comb's long-batch path against the oracle: python tools/diag_wrap.py stack W H frames"""
import sys, numpy, torch
sys.path.insert(0, '.'); sys.path.insert(0, 'tests')
import stacks
from color_modem_amd import image, testing
from oracle import cm_oracle
name, w, h, frames = sys.argv[1], int(sys.argv[2]), int(sys.argv[3]), int(sys.argv[4])
modem = stacks.make(name, (w, h))
eng = image.ImageModem(modem)._engine()
print(eng.describe())
few = testing.synthetic_rgb(3, h, w, seed=31 + h)
comp3 = cm_oracle.modulate_frames_f32(stacks.make('pal_s', (w, h)), few, first_frame=0, n_threads=4)
comp = torch.from_numpy(comp3).cuda().repeat((frames + 2) // 3, 1, 1)[:frames].contiguous()
first = 5
got = eng.demodulate_frames(comp, first_frame=first)
for i in (0, 1, 2, frames // 2, frames - 1):
    want = cm_oracle.demodulate_frames_f32(modem, comp3[i % 3][None], first_frame=first + i, n_threads=4)[0]
    g = got[i].cpu().numpy().astype(numpy.float64)
    err = numpy.abs(g - want).max(axis=(0, 2)) / numpy.abs(want).max()
    errp = numpy.abs(g - want).max(axis=2) / numpy.abs(want).max()
    print('frame', i, 'rows with err > 1e-5:', [(r, '%.2g' % err[r], ['%.1g' % errp[p, r] for p in range(3)]) for r in range(h) if err[r] > 1e-5])
