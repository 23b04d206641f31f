import sys, os, ctypes, numpy, torch
sys.path.insert(0, '.'); sys.path.insert(0, 'tests')
from color_modem_amd import _native
_native.LIB_PATH = os.environ['CM_LIB']
import stacks
from color_modem_amd import image, testing
F = 1000
modem = stacks.make('pal_d', (720, 576)); eng = image.ImageModem(modem)._engine()
comp = torch.rand((F, 576, 720), device='cuda'); out = torch.empty((F, 3, 576, 720), device='cuda')
nb = 9200
dbg = torch.zeros((nb, 8), dtype=torch.int64, device='cuda')
L = _native.lib(); L.cm_diag_set_buffer.argtypes = [ctypes.c_void_p]; L.cm_diag_set_buffer(dbg.data_ptr())
eng.demodulate_frames(comp, 0, out=out); torch.cuda.synchronize()
dbg.zero_(); eng.demodulate_frames(comp, 0, out=out); torch.cuda.synchronize()
d = dbg.cpu().numpy().astype(numpy.float64); d = d[d[:, 0] > 0]
tot = d[:, 0]
print('workgroups', len(d), 'mean total cycles %.0f (min %.0f max %.0f)' % (tot.mean(), tot.min(), tot.max()))
for i, n in enumerate(['total', 'flush(+drain)', 'fill issue + tile wait', 'x read (LDS latency)', 'luma vmcnt wait', 'substeps']):
    print('  %-26s %10.0f  %.1f %%' % (n, d[:, i].mean(), 100 * d[:, i].mean() / tot.mean()))
if d[:, 6].max() > 0:
    print('in-kernel clock: %.3f GHz (d s_memtime / d s_memrealtime x 100 MHz, median); lifetime %.1f us' % (numpy.median(d[:, 0] / d[:, 6]) * 0.1, numpy.median(d[:, 6]) / 100))
