"""Where the two waves of a wave-pair workgroup spend their cycles (library built with -DCM_DIAG).
usage: CM_LIB=build_ab/libdiag.so python tools/diag_pair.py"""
import sys, os, ctypes, numpy, torch
sys.path.insert(0, '.'); sys.path.insert(0, 'tests')
from color_modem_amd import _native
import stacks
from color_modem_amd import image
F = 1000
modem = stacks.make('pal_d', (720, 576)); eng = image.ImageModem(modem)._engine()
comp = torch.rand((F, 576, 720), device='cuda'); out = torch.empty((F, 3, 576, 720), device='cuda')
nb = 9300
dbg = torch.zeros((nb, 16), dtype=torch.int64, device='cuda')
L = _native.lib(); L.cm_diag_set_buffer.argtypes = [ctypes.c_void_p]; L.cm_diag_set_buffer(dbg.data_ptr())
eng.demodulate_frames(comp, 0, out=out); torch.cuda.synchronize()
dbg.zero_(); eng.demodulate_frames(comp, 0, out=out); torch.cuda.synchronize()
d = dbg.cpu().numpy().astype(numpy.float64); d = d[d[:, 0] > 0]
for role, off, names in (('A', 0, ['total', 'barrier wait', 'tile fill wait']), ('B', 8, ['total', 'barrier wait', 'flush'])):
    tot = d[:, off]
    print('stage %s: workgroups %d, mean lifetime %.0f cycles (min %.0f max %.0f)' % (role, len(d), tot.mean(), tot.min(), tot.max()))
    for i, n in enumerate(names):
        print('  %-18s %10.0f  %.1f %%' % (n, d[:, off + i].mean(), 100 * d[:, off + i].mean() / tot.mean()))
rt = d[:, 11]
print('in-kernel clock: %.3f GHz (median over workgroups of d s_memtime / d s_memrealtime x 100 MHz); lifetime %.1f us' % (numpy.median(d[:, 8] / rt) * 0.1, numpy.median(rt) / 100))
# which SIMD hosts which stage (HW_ID: SIMD_ID = bits 5:4, CU_ID = bits 11:8)
import collections
for role, off in (('A', 4), ('B', 12)):
    hw = d[:, off].astype(numpy.int64)
    print('stage %s waves per SIMD id:' % role, dict(sorted(collections.Counter(((hw >> 4) & 3).tolist()).items())))
a, b = d[:, 4].astype(numpy.int64), d[:, 12].astype(numpy.int64)
print('pairs (SIMD of A, SIMD of B):', dict(sorted(collections.Counter(zip(((a >> 4) & 3).tolist(), ((b >> 4) & 3).tolist())).items())))
print('same CU for both waves: %.3f' % numpy.mean(((a >> 8) & 15) == ((b >> 8) & 15)))
