"""Where a wave of the scan kernel spends its time (library built with -DCM_DIAG): CM_LIB=... python tools/scan_diag.py"""
import sys, os, ctypes, numpy, torch
sys.path.insert(0, '.'); sys.path.insert(0, 'tests')
from color_modem_amd import _native
_native.LIB_PATH = os.environ['CM_LIB']
import stacks
from color_modem_amd import image, testing
modem = stacks.make('pal_d', (720, 576)); eng = image.ImageModem(modem)._engine()
eng.set_small_batch('scan')
comp = torch.from_numpy(testing.synthetic_composite(1, 576, 720)).cuda(); out = torch.empty((1, 3, 576, 720), device='cuda')
dbg = torch.zeros((512 * 4, 16), dtype=torch.int64, device='cuda')
L = _native.lib(); L.cm_diag_set_buffer.argtypes = [ctypes.c_void_p]; L.cm_diag_set_buffer(dbg.data_ptr())
for _ in range(3): eng.demodulate_frames(comp, 0, out=out)
torch.cuda.synchronize(); dbg.zero_(); eng.demodulate_frames(comp, 0, out=out); torch.cuda.synchronize()
d = dbg.cpu().numpy().astype(numpy.float64); d = d[d[:, 15] >= 13]
names = ['row in LDS', 'up2', 'band-pass', 'put, dn2, up2 of e', 'detector products', 'low-pass', 'put, dn2 x 2, base out', 'barrier + lane constants',
         'combination', 'pre-correction + put', 're-modulation', 'matrix + stores']
steps = numpy.diff(d[:, :13], axis=1)
tot = d[:, 12] - d[:, 0]
print('waves', len(d), 'cycles per wave: mean %.0f (min %.0f, max %.0f); lifetime %.2f us (s_memrealtime)' % (tot.mean(), tot.min(), tot.max(), numpy.median(d[:, 14]) / 100))
for i, n in enumerate(names):
    print('  %-28s %8.0f  %5.1f %%' % (n, steps[:, i].mean(), 100 * steps[:, i].mean() / tot.mean()))
