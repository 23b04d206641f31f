# -*- coding: utf-8 -*-
"""color_modem_amd/design.py (the package's own filter design, no scipy at run time) against scipy.signal, the library the
reference designs with (/root/reference/color_modem/utils.py:9-64, comb.py:18-20, the FIR behind resample_poly).

scipy is TEST infrastructure here: the product modules must construct every modem and build every plan descriptor with scipy
blocked (test_product_designs_without_scipy)."""
import os
import subprocess
import sys
import warnings

import numpy
import pytest
import scipy.signal

from color_modem_amd import design

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def close(got, want, tol):
    got, want = numpy.asarray(got), numpy.asarray(want)
    assert got.shape == want.shape
    if got.size:
        scale = max(1.0, float(numpy.max(numpy.abs(want))))
        assert float(numpy.max(numpy.abs(got - want))) <= tol * scale


@pytest.mark.parametrize('order', range(1, 9))
def test_prototypes(order):
    for mine, theirs, args in ((design.buttap, scipy.signal.buttap, ()), (design.cheb1ap, scipy.signal.cheb1ap, (3.0,)),
                               (design.cheb1ap, scipy.signal.cheb1ap, (0.5,)), (design.cheb2ap, scipy.signal.cheb2ap, (48.0,)),
                               (design.cheb2ap, scipy.signal.cheb2ap, (20.0,)), (design.besselap, scipy.signal.besselap, ())):
        z, p, k = mine(order, *args)
        zs, ps, ks = theirs(order, *args)
        close(z, zs, 1e-15)
        close(p, ps, 4e-15)      # besselap: Newton on the polynomial here, on K_v in scipy - the last bits
        close(k, ks, 1e-15)


@pytest.mark.parametrize('ftype,kw', [('butter', {}), ('cheby1', {'rp': 3.0}), ('cheby2', {'rs': 48.0}), ('bessel', {}),
                                      ('cheby1', {'rp': 0.5}), ('cheby2', {'rs': 20.0})])
def test_iirfilter_sweep(ftype, kw):
    rng = numpy.random.default_rng(hash(ftype) % 1000 + len(kw))
    bit_equal = total = 0
    for order in range(1, 9):
        for btype in ('lowpass', 'highpass', 'bandpass', 'bandstop'):
            for _ in range(4):
                if btype in ('lowpass', 'highpass'):
                    wn = rng.uniform(0.02, 0.95)
                else:
                    lo = rng.uniform(0.02, 0.8)
                    wn = [lo, lo + rng.uniform(0.02, 0.95 - lo)]
                b, a = design.iirfilter(order, wn, btype=btype, ftype=ftype, **kw)
                bs, as_ = scipy.signal.iirfilter(order, wn, btype=btype, ftype=ftype, **kw)
                close(b, bs, 1e-11)
                close(a, as_, 1e-11)
                sos = design.iirfilter(order, wn, btype=btype, ftype=ftype, output='sos', **kw)
                sos_s = scipy.signal.iirfilter(order, wn, btype=btype, ftype=ftype, output='sos', **kw)
                close(sos, sos_s, 1e-12)     # same pairing, same section order
                total += 1
                bit_equal += numpy.array_equal(sos, sos_s)
                z, p, k = design.iirfilter(order, wn, btype=btype, ftype=ftype, output='zpk', **kw)
                zs, ps, ks = scipy.signal.iirfilter(order, wn, btype=btype, ftype=ftype, output='zpk', **kw)
                close(z, zs, 1e-13)
                close(p, ps, 1e-13)
                close(k, ks, 1e-13)
    if ftype != 'bessel':
        assert bit_equal >= 0.9 * total      # the same arithmetic in the same order: equal to the bit almost everywhere


def test_buttord_every_band_type():
    rng = numpy.random.default_rng(5)
    for rep in range(400):
        kind = rep % 4
        if kind == 0:
            wp = rng.uniform(0.05, 0.5)
            ws = wp + rng.uniform(0.02, 0.4)
        elif kind == 1:
            ws = rng.uniform(0.05, 0.5)
            wp = ws + rng.uniform(0.02, 0.4)
        else:
            c = rng.uniform(0.2, 0.7)
            inner = rng.uniform(0.01, 0.1)
            outer = inner + rng.uniform(0.01, 0.1)
            wp, ws = ([c - inner, c + inner], [c - outer, c + outer]) if kind == 2 else ([c - outer, c + outer], [c - inner, c + inner])
        gpass, gstop = rng.uniform(0.5, 3.5), rng.uniform(10, 50)
        order, wn = design.buttord(wp, ws, gpass, gstop)
        order_s, wn_s = scipy.signal.buttord(wp, ws, gpass, gstop)
        assert order == order_s
        close(wn, wn_s, 1e-15)       # the band-stop case runs the bounded minimiser: the same iterates


def test_notch_responses_and_fir():
    rng = numpy.random.default_rng(9)
    for w0, q in ((0.3, 2.0), (0.6568, 0.5), (0.53, 10.0), (0.657, 5.0)):
        b, a = design.iirnotch(w0, q)
        bs, as_ = scipy.signal.iirnotch(w0, q)
        close(b, bs, 1e-15)
        close(a, as_, 1e-15)
        close(design.tf2sos(b, a), scipy.signal.tf2sos(bs, as_), 1e-14)
    for _ in range(40):
        order = int(rng.integers(1, 5))
        lo = rng.uniform(0.1, 0.5)
        b, a = scipy.signal.iirfilter(order, [lo, lo + 0.2], btype='bandpass')
        w = rng.uniform(lo, lo + 0.2)        # in the pass band, where FilterFunction evaluates them
        close(design.freqz_at(b, a, w), scipy.signal.freqz(b, a, worN=[w], fs=2.0)[1][0], 1e-12)
        close(design.group_delay_at(b, a, w), scipy.signal.group_delay((b, a), [w], fs=2.0)[1][0], 1e-10)
    for rate in (2, 3, 4, 9):
        close(design.resample_poly_fir(rate), scipy.signal.firwin(2 * 10 * rate + 1, 1.0 / rate, window=('kaiser', 5.0)), 2e-16)


def test_every_design_the_modems_request():
    """The (b, a, shift, phase_shift) of every FilterFunction of every stack, rebuilt with scipy from the same request."""
    sys.path.insert(0, os.path.join(ROOT, 'tests'))
    import stacks
    from color_modem_amd import utils
    seen = []
    real_init = utils.FilterFunction.__init__

    def spy(self, b, a, wp, btype, shift, sos=None):
        real_init(self, b, a, wp, btype, shift, sos=sos)
        seen.append((self, wp, btype, shift))

    utils.FilterFunction.__init__ = spy
    try:
        for name in sorted(stacks.STACKS):
            for size in ((720, 576), (960, 576), (1280, 480)):
                try:
                    stacks.make(name, size)
                except Exception:
                    continue      # shapes a stack does not take (tested elsewhere)
    finally:
        utils.FilterFunction.__init__ = real_init
    assert len(seen) > 300
    for f, wp, btype, shift in seen:
        # scipy's own factoring of (b, a), its group delay and response at the centre FilterFunction picks
        centre = f.shift_frequency
        if shift:
            with warnings.catch_warnings():
                warnings.simplefilter('ignore')
                gd = scipy.signal.group_delay((f.b, f.a), [centre], fs=2.0)[1][0]
            assert f.shift == int(numpy.round(gd))
        resp = scipy.signal.freqz(f.b, f.a, worN=[centre], fs=2.0)[1][0]
        want = float((numpy.angle(resp) + f.shift * numpy.pi * centre) % (2.0 * numpy.pi))
        d = abs(f.phase_shift - want)
        assert min(d, 2.0 * numpy.pi - d) < 1e-9     # (b, a) polynomials of order-8 band-passes near DC: both evaluations round at 1e-11
        # the sections multiply back to (b, a)
        sos = f.sos()
        b = numpy.array([1.0])
        a = numpy.array([1.0])
        for s in sos:
            b = numpy.convolve(b, s[:3])
            a = numpy.convolve(a, s[3:])
        n = max(len(f.b), len(f.a))
        close(numpy.trim_zeros(b, 'b')[:n], numpy.trim_zeros(numpy.asarray(f.b), 'b'), 1e-9)
        close(numpy.trim_zeros(a, 'b')[:n], numpy.trim_zeros(numpy.asarray(f.a), 'b'), 1e-9)


def test_product_designs_without_scipy():
    """Every modem constructs and every QAM / SECAM plan descriptor builds with scipy un-importable."""
    code = r'''
import sys
class Block(object):
    def find_spec(self, name, path=None, target=None):
        if name == 'scipy' or name.startswith('scipy.'):
            raise ImportError('scipy is blocked in this test')
        return None
sys.meta_path.insert(0, Block())
sys.path.insert(0, %r)
sys.path.insert(0, %r)
import stacks
from color_modem_amd import plan, line
from color_modem_amd.color import niir, protosecam, mac
n = 0
for name in sorted(stacks.STACKS):
    m = stacks.make(name, (720, 576))
    n += 1
lc = line.LineConfig((720, 576))
for m in (niir.NiirModem(lc), niir.HueCorrectingNiirModem(lc), protosecam.ProtoSecamModem(line.LineConfig((720, 736))), mac.MacModem(lc)):
    n += 1
assert 'scipy' not in sys.modules and 'scipy.signal' not in sys.modules
print('ok', n)
''' % (ROOT, os.path.join(ROOT, 'tests'))
    out = subprocess.run([sys.executable, '-c', code], capture_output=True, text=True, timeout=300)
    assert out.returncode == 0, out.stderr[-2000:]
    assert out.stdout.strip().startswith('ok')
