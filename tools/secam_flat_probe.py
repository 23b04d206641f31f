"""SECAM encoder error on flat saturated pictures and random ones, streaming kernel pinned: python tools/secam_flat_probe.py
(CM_LIB picks an experimental build, e.g. -DCM_EXPERIMENTS -DCM_EXP_SECAM_MOD_F32: profiles/r06_secam_mod_bound.txt)"""
import sys, numpy
sys.path.insert(0, '.'); sys.path.insert(0, 'tests')
import stacks
from color_modem_amd import image, testing
from oracle import cm_oracle
for stack, size in (('secam', (720, 16)), ('secam', (1280, 12)), ('secam_avg', (720, 16))):
    m = stacks.make(stack, size)
    pics = []
    for rgb in ((1, 0, 0), (0, 0, 1), (0, 1, 0), (1, 1, 0), (0.75, 0.1, 0.6)):
        p = numpy.zeros((3, size[1], size[0]), numpy.float32)
        for c in range(3):
            p[c] = rgb[c]
        pics.append(p)
    pics = numpy.concatenate([numpy.stack(pics), testing.synthetic_rgb(6, size[1], size[0], seed=9)])
    eng = image.ImageModem(m)._engine()
    eng.set_small_batch('rows')          # the streaming encoder (small batches take the scan encoder otherwise)
    got = eng.modulate_frames(pics, first_frame=1)
    want = cm_oracle.modulate_frames_f32(m, pics, first_frame=1)
    errs = [stacks.rel_err(got[i], want[i]) for i in range(len(pics))]
    print('%-10s %4dx%-3d flat pictures %s   random pictures worst %.3g' % (stack, size[0], size[1], ' '.join('%.2e' % e for e in errs[:5]), max(errs[5:])))
