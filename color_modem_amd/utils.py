# -*- coding: utf-8 -*-
"""IIR design wrappers and carrier bookkeeping (host side).

Mirrors the names of the reference's ``color_modem/utils.py`` (``FilterFunction``,
``iirfilter``, ``iirdesign``, ``iirdesign_wc``, ``iirsplitter``, ``ConstantFrequencyCarrier``;
/root/reference/color_modem/utils.py:9-88).  Design runs once per modem on the host with the
package's own design code (``color_modem_amd/design.py``: the algorithms behind the
scipy.signal calls the reference makes, held to scipy's results by tests/test_design.py - no
scipy at run time since round 4); what differs is what a ``FilterFunction`` *is* here: a design
record (b, a, shift, phase_shift, second-order sections) that the device plan consumes.
Filtering itself happens in the HIP kernels.

``iirdesign`` keeps the behaviour of the scipy release the reference was written against:
scipy >= 1.12 rejects band edges <= 0, which the reference's NTSC-M set-up relies on
(SURVEY.md D6).  ``design.buttord`` + ``design.iirfilter`` take the request as it is - identical
coefficients wherever current scipy accepts it.
"""

import fractions

import numpy

from color_modem_amd import design

_BANDSTOP_NAMES = frozenset(('bs', 'bandstop', 'bands', 'stop'))


class FilterFunction(object):
    """Design record of one IIR filter plus the delay compensation of ref utils.py:9-26."""

    def __init__(self, b, a, wp, btype, shift, sos=None):
        self.b = numpy.array(b, dtype=numpy.float64)
        self.a = numpy.array(a, dtype=numpy.float64)
        self._sos = None if sos is None else numpy.array(sos, dtype=numpy.float64)
        wp = numpy.atleast_1d(wp)
        if len(wp) > 1 and btype.lower() not in _BANDSTOP_NAMES:
            centre = float(numpy.average(wp))
        else:
            centre = 0.0
        self.shift_frequency = centre
        if shift:
            self.shift = int(numpy.round(design.group_delay_at(self.b, self.a, centre)))
        else:
            self.shift = 0
        response = design.freqz_at(self.b, self.a, centre)
        self.phase_shift = float((numpy.angle(response) + self.shift * numpy.pi * centre) % (2.0 * numpy.pi))

    # the reference keeps these as private attributes; expose both spellings
    @property
    def _b(self):
        return self.b

    @property
    def _a(self):
        return self.a

    @property
    def _shift(self):
        return self.shift

    @property
    def order(self):
        return max(len(self.a), len(self.b)) - 1

    def sos(self):
        """Second-order sections [n, 6].

        Taken from the zero/pole form of the same design when the filter came from one of the
        wrappers below (exact zeros at z = +-1 / on the unit circle; factoring (b, a) back would
        smear repeated roots by ~1e-8), otherwise factored from (b, a)."""
        if self._sos is not None:
            return self._sos
        return design.tf2sos(self.b, self.a)

    def __call__(self, x):
        """The filter applied to a row (or to every row of a 2-D array) with the reference's delay compensation (ref utils.py:28-36:
        the row padded with ``shift`` copies of its last sample, the first ``shift`` results dropped), float64 on the GPU
        (``cm_filter_rows_f64``: lfilter's own recurrence, one lane per row).  The modems do not come through here - their filters
        are stages of the fused kernels; this is the reference's REPL-level callable."""
        import ctypes
        from color_modem_amd import _native
        from color_modem_amd.engine import _torch
        torch = _torch()
        rows = numpy.atleast_2d(numpy.ascontiguousarray(x, dtype=numpy.float64))
        if rows.ndim != 2 or rows.shape[1] < 1:
            raise ValueError('a row or an array of rows expected')
        dev_in = torch.from_numpy(rows).cuda()
        dev_out = torch.empty_like(dev_in)
        b = numpy.ascontiguousarray(self.b, dtype=numpy.float64)
        a = numpy.ascontiguousarray(self.a, dtype=numpy.float64)
        dp = ctypes.POINTER(ctypes.c_double)
        stream = torch.cuda.current_stream()
        _native.check(_native.lib().cm_filter_rows_f64(b.ctypes.data_as(dp), len(b), a.ctypes.data_as(dp), len(a), int(self.shift),
                                                       dev_in.data_ptr(), dev_out.data_ptr(), rows.shape[0], rows.shape[1],
                                                       stream.cuda_stream))
        out = dev_out.cpu().numpy()
        return out[0] if numpy.ndim(x) == 1 else out


def notch(qam_modem, q):
    """Luma notch at the sub-carrier (ref comb.py:18-20)."""
    b, a = design.iirnotch(2.0 * qam_modem.config.fsc / qam_modem.line_config.fs, q)
    f = FilterFunction(b, a, wp=0.0, btype='bandstop', shift=True)
    f.q = float(q)          # the request itself (the oracle designs its own copy from it)
    return f


def iirfilter(N, Wn, rp=None, rs=None, btype='band', ftype='butter', shift=True):
    b, a = design.iirfilter(N, Wn, rp, rs, btype, ftype=ftype)
    sos = design.iirfilter(N, Wn, rp, rs, btype, ftype=ftype, output='sos')
    return FilterFunction(b, a, Wn, btype, shift, sos=sos)


def _legacy_scipy_iirdesign(wp, ws, gpass, gstop, ftype):
    if ftype != 'butter':
        raise ValueError('only Butterworth designs are requested on this path')
    wp = numpy.atleast_1d(wp)
    ws = numpy.atleast_1d(ws)
    if len(wp) == 1:
        btype = 'lowpass' if wp[0] < ws[0] else 'highpass'
    else:
        btype = 'bandstop' if wp[0] < ws[0] else 'bandpass'
    order, natural = design.buttord(wp, ws, gpass, gstop)
    ba = design.iirfilter(order, natural, rp=gpass, rs=gstop, btype=btype, ftype='butter', output='ba')
    sos = design.iirfilter(order, natural, rp=gpass, rs=gstop, btype=btype, ftype='butter', output='sos')
    return ba[0], ba[1], sos


def iirdesign(wp, ws, gpass, gstop, ftype='butter', shift=True):
    tiny = numpy.nextafter(0.0, 1.0)
    below_one = numpy.nextafter(1.0, 0.0)
    b, a, sos = _legacy_scipy_iirdesign(numpy.maximum(wp, tiny), numpy.minimum(ws, below_one), gpass, gstop, ftype)
    wp_arr = numpy.atleast_1d(wp)
    ws_arr = numpy.atleast_1d(ws)
    btype = 'bandstop' if (len(wp_arr) > 1 and len(ws_arr) > 1 and ws_arr[0] > wp_arr[0]) else 'band'
    return FilterFunction(b, a, wp, btype, shift, sos=sos)


def iirdesign_wc(wc, wp, ws, gpass, gstop, ftype='butter', shift=True):
    return iirdesign([wc - wp, wc + wp], [wc - ws, wc + ws], gpass, gstop, ftype, shift)


def _complement_db(db):
    return -(20.0 * numpy.log10(1.0 - 10.0 ** (-db / 20.0)))


def iirsplitter(wc, wp, ws, gpass, gstop, ftype='butter', shift=True):
    """Complementary band-pass / band-stop pair around `wc` (ref utils.py:58-64)."""
    bandpass = iirdesign_wc(wc, wp, ws, gpass, gstop, ftype, shift)
    bandstop = iirdesign_wc(wc, ws, wp, _complement_db(gstop), _complement_db(gpass), ftype, shift)
    return bandpass, bandstop


class ConstantFrequencyCarrier(object):
    """Subcarrier phase bookkeeping; needs ``self.config.fsc`` and ``self.line_config`` (ref utils.py:67-88)."""

    @property
    def line_shift(self):
        std = self.line_config.line_standard
        return 2.0 * numpy.pi * ((self.config.fsc / (std.frame_rate * std.total_lines)) % 1.0)

    @property
    def frame_shift(self):
        return 2.0 * numpy.pi * ((self.config.fsc / self.line_config.line_standard.frame_rate) % 1.0)

    @property
    def frame_cycle(self):
        ratio = fractions.Fraction(self.config.fsc / self.line_config.line_standard.frame_rate)
        return ratio.limit_denominator().denominator

    def start_phase(self, frame, line):
        std = self.line_config.line_standard
        first_line = min(std.odd_field_first_active_line, std.even_field_first_active_line)
        two_pi = 2.0 * numpy.pi
        from_frame = ((frame % self.frame_cycle) * self.frame_shift) % two_pi
        from_line = ((self.line_config.analog_line(line) - first_line) * self.line_shift) % two_pi
        return (from_frame + from_line) % two_pi
