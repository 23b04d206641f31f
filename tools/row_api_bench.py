import sys, time, numpy
sys.path.insert(0, '.'); sys.path.insert(0, 'tests')
import stacks
from color_modem_amd import testing
m = stacks.make('pal_d', (720, 576))
comp = testing.synthetic_composite(1, 576, 720)[0]
m.demodulate(0, 0, comp[0])
t0 = time.time()
for y in range(0, 576, 2): m.demodulate(0, y, comp[y])
for y in range(1, 576, 2): m.demodulate(0, y, comp[y])
dt = time.time() - t0
print('per-row API: %.1f ms per frame, %.3f ms per row' % (dt * 1e3, dt / 576 * 1e3))
enc = stacks.make('pal_s', (720, 576))
rgb = testing.synthetic_rgb(1, 576, 720)[0]
t0 = time.time()
for y in range(0, 576, 2): enc.modulate(0, y, rgb[0, y], rgb[1, y], rgb[2, y])
dt = time.time() - t0
print('per-row modulate: %.3f ms per row' % (dt / 288 * 1e3))
