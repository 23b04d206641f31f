# -*- coding: utf-8 -*-
"""The reference's row-level protocol on the GPU, and what this package adds beside it.

    python examples/protocol_additions.py        (needs an MI355X)

1. ``modem.demodulate(frame, line, row)`` - the reference's one-row-per-call protocol (comb.py:47-59), stateful like the reference.
2. ``modem.demodulate_rows(frame, line, rows)`` - the same results for a whole group of rows of a field from ONE launch (an addition).
3. ``SimpleCombModem(..., avg=f)`` with a function of the caller's own (comb.py:72): ``f`` gets float32 torch tensors on the device.
4. ``notch=1.0`` - a notch whose FilterFunction comes out with a shift of 1 sample (comb.py:18-20 over utils.py:9-26).
5. A ``FilterFunction`` is callable as in the reference (utils.py:28-36); the design code is the package's own (no scipy at run time).
6. Round 5: ``Pal3DModem(avg=f)`` with the caller's own function (pal.py:144-148) - also one written with numpy ufuncs; comb wrappers around
   Pal3DModem (long batches: one launch); ``ImageModem(modem, batch_invariant=True)``: results independent of the batch, bit for bit.
"""
import os
import sys
import time

import numpy

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))

from color_modem_amd import comb  # noqa: E402
from color_modem_amd.color.pal import Pal3DModem, PalDModem, PalSModem  # noqa: E402
from color_modem_amd.image import ImageModem  # noqa: E402
from color_modem_amd.line import LineConfig  # noqa: E402


def main():
    lc = LineConfig((720, 576))
    rng = numpy.random.default_rng(1)
    rgb = rng.uniform(0.0, 1.0, (1, 3, 576, 720 + 3)).astype(numpy.float32)
    rgb = 0.25 * (rgb[..., :-3] + rgb[..., 1:-2] + rgb[..., 2:-1] + rgb[..., 3:])
    composite = ImageModem(PalSModem(lc)).modulate_frames(rgb, first_frame=0)[0]       # [576, 720]
    field = composite[0::2]                                                              # lines 0, 2, 4, ...

    modem, grouper = PalDModem(lc), PalDModem(lc)
    modem.demodulate(1, 0, field[0])            # (the first call builds the device plan: keep it out of the timing)
    grouper.demodulate_rows(1, 0, field[:2])
    t0 = time.time()
    one_by_one = numpy.stack([numpy.stack(modem.demodulate(0, 2 * i, field[i])) for i in range(len(field))])
    t1 = time.time()
    grouped = grouper.demodulate_rows(0, 0, field)
    t2 = time.time()
    print('a field of %d rows: one call per row %.1f ms, one call for the field %.1f ms, largest difference %.1e'
          % (len(field), (t1 - t0) * 1e3, (t2 - t1) * 1e3, float(numpy.abs(one_by_one - grouped).max())))

    def favour_the_new_line(last, curr):        # any elementwise function of two arrays
        return 0.25 * last + 0.75 * curr

    weighted = comb.Simple3DCombModem(PalDModem(lc), avg=favour_the_new_line)
    plain = comb.Simple3DCombModem(PalDModem(lc))
    a = ImageModem(weighted).demodulate_frames(composite[None], first_frame=0)[0]
    b = ImageModem(plain).demodulate_frames(composite[None], first_frame=0)[0]
    print('Simple3DCombModem(PalDModem, avg=f): differs from comb.avg by up to %.3f of full scale' % float(numpy.abs(a - b).max()))

    notched = PalDModem(lc, notch=1.0)
    print('PalDModem(notch=1.0): FilterFunction shift %d; decoded frame finite: %s'
          % (notched.notch.shift, bool(numpy.isfinite(ImageModem(notched).demodulate_frames(composite[None], first_frame=0)).all())))

    lowpass = PalSModem(lc).qam._chroma_precorrect_lowpass
    step = numpy.concatenate([numpy.zeros(20), numpy.ones(40)])
    print('FilterFunction(step)[18:26] =', numpy.round(lowpass(step)[18:26], 4), ' (shift %d)' % lowpass.shift)

    def numpy_style(a, b):                      # numpy ufuncs cannot take device tensors: this one is called with float64 numpy arrays
        return 0.5 * (a + b) * numpy.exp(-2.0 * numpy.abs(a - b))

    c = ImageModem(Pal3DModem(lc, avg=numpy_style)).demodulate_frames(composite[None], first_frame=0)[0]
    d = ImageModem(Pal3DModem(lc)).demodulate_frames(composite[None], first_frame=0)[0]
    print('Pal3DModem(avg=f): differs from comb.avg by up to %.3f of full scale' % float(numpy.abs(c - d).max()))

    batch = numpy.repeat(composite[None], 12, axis=0)
    wrapped = ImageModem(comb.Simple3DCombModem(Pal3DModem(lc)), batch_invariant=True)
    whole = wrapped.demodulate_frames(batch, first_frame=0)
    one = wrapped.demodulate_frames(batch[4:5], first_frame=4)
    print('Simple3DCombModem(Pal3DModem), batch_invariant=True: frame 4 alone equals frame 4 of the batch bit for bit: %s'
          % bool(numpy.array_equal(one[0], whole[4])))


if __name__ == '__main__':
    main()
