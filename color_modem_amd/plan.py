# -*- coding: utf-8 -*-
"""Turn a modem stack into the flat plan descriptor of include/color_modem_hip.h.

The reference decoders are stateful row-at-a-time objects.  Every one of them is, per
output row, a *linear* function of at most three consecutive input rows of one field
(no inter-line recurrence).  With

    B_k = (Bs, Bc)[k] = base demodulation of call k's own input row, detector phase theta_k

(ref qam.py:47-54 for the QAM front end, pal.py:71-77 on qam.py:34-37 for PAL-D) and the
identity  demod(theta + d) = rot(demod(theta), d),  each decoder becomes

    u = sum_j cu[j] . B_{k-j},   v = sum_j cv[j] . B_{k-j},   y = x_{k-dy} - remod(phi; u, v)

with coefficients that depend only on (frame mod cycle, line, k regime).  This module derives
those coefficients from the same formulas the reference executes on signals, applied to small
symbolic objects (`Lin`), in float64.
"""

import ctypes

import numpy
from color_modem_amd import design

CM_ABI_VERSION = 8
CM_SECAM_PRESENT, CM_SECAM_FLOAT64 = 1, 2      # cm_secam_desc.present (include/color_modem_hip.h)
CM_PIPE_QAM, CM_PIPE_PAL_D, CM_PIPE_SECAM = 1, 2, 3
CM_MAX_SECTIONS = 4
CM_LANE_DOUBLES = 32
CM_AVG_FOLDED, CM_AVG_MIN = 0, 1
MAX_TABLE_CYCLE = 64          # longer sub-carrier cycles: two parity frames + a per-frame rotation
MAX_ROTATION_CYCLE = 1 << 21


class IirDesc(ctypes.Structure):
    _fields_ = [('n_sections', ctypes.c_int32), ('shift', ctypes.c_int32),
                ('sos', (ctypes.c_double * 6) * CM_MAX_SECTIONS)]


class LaneTable(ctypes.Structure):
    _fields_ = [('frame_cycle', ctypes.c_int32), ('n_lines', ctypes.c_int32),
                ('table', ctypes.POINTER(ctypes.c_double)),
                ('luma_from_prev', ctypes.c_int32), ('wrap_mode', ctypes.c_int32)]


class SecamDesc(ctypes.Structure):
    _fields_ = [('present', ctypes.c_int32), ('preroll', ctypes.c_int32),
                ('flimit_min', ctypes.c_double), ('flimit_max', ctypes.c_double), ('bell_f0', ctypes.c_double),
                ('m0', ctypes.c_double), ('bell_kn', ctypes.c_double), ('bell_kd', ctypes.c_double),
                ('fm_fc', ctypes.c_double),
                ('pre_lp', IirDesc), ('lf_pre', IirDesc), ('lf_rev', IirDesc), ('bell', IirDesc),
                ('chroma_bp', IirDesc), ('luma_bs', IirDesc), ('fm_lp', IirDesc)]


class PlanDesc(ctypes.Structure):
    _fields_ = [('abi_version', ctypes.c_int32), ('pipeline', ctypes.c_int32),
                ('width', ctypes.c_int32), ('height', ctypes.c_int32),
                ('demodulation_delay', ctypes.c_int32), ('modulation_delay', ctypes.c_int32),
                ('depth', ctypes.c_int32), ('first_is_plain', ctypes.c_int32),
                ('main_luma_bandstop', ctypes.c_int32), ('skip_calls', ctypes.c_int32),
                ('carrier_phase_step', ctypes.c_double),
                ('resample_fir', ctypes.c_double * 41),
                ('extract2x', IirDesc), ('remove2x', IirDesc), ('demod_lp', IirDesc),
                ('pald_lp', IirDesc), ('precorrect', IirDesc),
                ('decode_matrix', ctypes.c_double * 9), ('encode_matrix', ctypes.c_double * 9),
                ('demod_main', LaneTable), ('demod_first', LaneTable), ('secam', SecamDesc),
                ('mod_main', LaneTable),
                ('frame_rotation', ctypes.POINTER(ctypes.c_double)), ('frame_rotation_cycle', ctypes.c_int32),
                ('chroma_average', ctypes.c_int32), ('notch', IirDesc)]


# ---------------------------------------------------------------------------------------------
# symbolic linear combinations of base pairs

class Lin(object):
    """sum_j c[j, 0] * Bs[k - j] + c[j, 1] * Bc[k - j], j = 0..2."""
    __slots__ = ('c',)

    def __init__(self, c=None):
        self.c = numpy.zeros((3, 2)) if c is None else numpy.array(c, dtype=numpy.float64)

    def __add__(self, other):
        return Lin(self.c + other.c)

    def __sub__(self, other):
        return Lin(self.c - other.c)

    def __mul__(self, scalar):
        return Lin(self.c * float(scalar))

    __rmul__ = __mul__

    def __neg__(self):
        return Lin(-self.c)

    def shifted(self, rows):
        """The same combination evaluated `rows` calls earlier, seen from the current call."""
        out = numpy.zeros((3, 2))
        if rows:
            assert not self.c[3 - rows:].any(), 'combination reaches further back than 2 lines'
            out[rows:] = self.c[:3 - rows]
        else:
            out[:] = self.c
        return Lin(out)


def base(j):
    s, c = Lin(), Lin()
    s.c[j, 0] = 1.0
    c.c[j, 1] = 1.0
    return s, c


def rot(pair, delta):
    """Detector output for a phase advanced by `delta`: sin(a+d) = sin a cos d + cos a sin d, ..."""
    s, c = pair
    cd, sd = numpy.cos(delta), numpy.sin(delta)
    return cd * s + sd * c, cd * c - sd * s


def pair_sub(a, b):
    return a[0] - b[0], a[1] - b[1]


def pair_add(a, b):
    return a[0] + b[0], a[1] + b[1]


# ---------------------------------------------------------------------------------------------
# IIR descriptors

def bandpass_sections(sos):
    """Re-pair the zeros of a Butterworth/Chebyshev band-pass so every section has the numerator
    g * (1 - z^-2) (one zero at z = +1, one at z = -1).  scipy pairs (+1, +1) and (-1, -1); the
    cascade's transfer function is unchanged, each section just becomes a 3-operation update."""
    sos = numpy.array(sos, dtype=numpy.float64)
    zeros = numpy.concatenate([numpy.roots(row[:3]) for row in sos])
    n = len(sos)
    plus = int(numpy.sum(numpy.abs(zeros - 1.0) < 1e-6))
    minus = int(numpy.sum(numpy.abs(zeros + 1.0) < 1e-6))
    if plus != n or minus != n:
        raise NotImplementedError('band-pass with zeros %r is not of the (1 - z^-2)^n family' % (zeros,))
    gain = float(numpy.prod(sos[:, 0]))
    out = sos.copy()
    out[:, 0:3] = (1.0, 0.0, -1.0)
    out[0, 0:3] *= gain
    return out


def iir_desc(filt, bandpass=False):
    """FilterFunction -> IirDesc (sections from the design's zpk form when available)."""
    d = IirDesc()
    if filt is None:
        d.n_sections = 0
        d.shift = 0
        return d
    sos = filt.sos()
    if bandpass:
        sos = bandpass_sections(sos)
    if len(sos) > CM_MAX_SECTIONS:
        raise NotImplementedError('filter order %d exceeds the %d sections this build carries'
                                  % (filt.order, CM_MAX_SECTIONS))
    d.n_sections = len(sos)
    d.shift = int(filt.shift)
    for j, row in enumerate(sos):
        for i in range(6):
            d.sos[j][i] = float(row[i])
    return d


def resample_fir():
    # the filter scipy.signal.resample_poly designs for up/down = 2 (ref qam.py:35 etc.)
    return design.resample_poly_fir(2)


# ---------------------------------------------------------------------------------------------

class QamTables(object):
    """Per-line coefficient derivation for the QAM-family stacks."""

    def __init__(self, stack, strip_chroma=True, min_lines=0):
        self.stack = stack
        self.strip_chroma = strip_chroma         # False: demodulate_components(..., strip_chroma=False)
        self.kind = stack['kind']
        self.backend = stack['backend']          # PalSModem / NtscModem
        self.comb = stack.get('comb')            # PalDModem / Pal3DModem / NtscCombModem or None
        self.demod_wrapper = stack.get('demod_wrapper')
        self.mod_wrapper = stack.get('mod_wrapper')
        # SimpleCombModem / Simple3DCombModem around PalDModem, calls k >= 2 of a run only (wrapped.py: the fused plan): there both
        # chroma estimates of comb.py:103-104 are PAL-D decodes, i.e. combinations of this front end's base pairs over three lines;
        # the calls k < 2 mix in the plain decode of the first line and are left to the composition (skip_calls = 2)
        self.fused_main = bool(stack.get('fused_main'))
        # SimpleCombModem / Simple3DCombModem around Pal3DModem as a TWO-LEVEL comb (round 5): the lane tables are Pal3DModem's own
        # (components, strip_chroma=False - what the wrapper asks its backend for), and the kernel averages that result with the one the
        # neighbouring lane (the previous call) formed the same way - comb.py:103-104 as written - before it strips and notches at the
        # wrapper's line.  One more lane of halo, no fourth line in the tables.  (cm_lane_table::wrap_mode)
        self.two_level = bool(stack.get('two_level'))
        self.wrap_mode = 0
        b = self.backend
        self.lc = b.line_config
        self.LS = b.line_shift
        self.ps = b.qam.extract_chroma_phase_shift
        self.carrier_cycle = b.frame_cycle       # utils.py:78-80
        self.cycle = self.carrier_cycle
        if b.v_switch and self.cycle % 2:
            self.cycle *= 2  # the V switch alternates with frame parity (line.py:64-65)
        # Long cycles (4.43 MHz colour on 525 lines: 4800 frames) are not tabulated frame by frame: the tables hold
        # frames 0 and 1 (both parities of is_alternate_line) and frame F is frame F % 2 with every phase advanced
        # by frame_rotation()[F % rotation_cycle], applied by the kernels (cm_plan_desc::frame_rotation).
        self.rotating = self.cycle > MAX_TABLE_CYCLE
        self.rotation_cycle = 0
        self.table_frames = self.cycle
        if self.rotating:
            self.rotation_cycle = self.carrier_cycle * (2 if self.carrier_cycle % 2 else 1)
            if self.rotation_cycle > MAX_ROTATION_CYCLE:
                raise NotImplementedError('sub-carrier phase cycle of %d frames exceeds the rotation table limit'
                                          % self.carrier_cycle)
            self.table_frames = 2
        self.width, self.height = self.lc.size
        self.min_lines = int(min_lines)
        # line_offset o (round 6, the level-by-level engines of generic.py): the MODULATOR table's row i describes line i - o, so that an
        # encoder run can start above the picture - ColorAveragingModem sends its first call of a field to the backend at line - 2
        # (comb.py:152), two of them nested at line - 4.  The engine adds o to the line numbers it passes (Engine.modulate_run).
        self.line_offset = int(stack.get('line_offset', 0))
        # comb.py:9-15: which averaging function combines the two chroma estimates (wrapper, or Pal3DModem's two paths)
        from color_modem_amd import comb as comb_module
        fn = None
        if self.two_level:
            if self.kind != 'pal_3d' or not self.demod_wrapper:
                raise NotImplementedError('the two-level comb serves SimpleCombModem / Simple3DCombModem around Pal3DModem')
            wfn = stack.get('wrapper_avg')
            if wfn is None or wfn is comb_module.avg:
                self.wrap_mode = 1
            elif wfn is comb_module.minavg:
                self.wrap_mode = 2
            else:
                raise NotImplementedError('avg=%r around Pal3DModem: callables go through the composition (wrapped.py)' % (wfn,))
            if self.comb._use_sin and self.comb._use_cos:
                fn = self.comb._avg           # the tables' second estimate is Pal3DModem's own (pal.py:209-211)
        elif self.demod_wrapper:
            fn = stack.get('wrapper_avg')
        elif self.kind == 'pal_3d' and self.comb._use_sin and self.comb._use_cos:
            fn = self.comb._avg
        if fn is None or fn is comb_module.avg:
            self.minavg = False
        elif fn is comb_module.minavg:
            self.minavg = True
        else:
            raise NotImplementedError('avg=%r: the device path implements comb.avg and comb.minavg, not arbitrary '
                                      'callables' % (fn,))

    # -- helpers ---------------------------------------------------------------------------
    def phi(self, frame, line):
        return self.backend.start_phase(frame, line)

    def frame_rotation(self):
        """{cos, sin} of phi(F, line) - phi(F % 2, line) for F < rotation_cycle (utils.py:82-88: the frame's
        share of the start phase is ((F % cycle) * frame_shift) % 2 pi for every line)."""
        two_pi = 2.0 * numpy.pi
        fs = self.backend.frame_shift
        rot = numpy.zeros((self.rotation_cycle, 2))
        for i in range(self.rotation_cycle):
            delta = ((i % self.carrier_cycle) * fs) % two_pi - (((i % 2) % self.carrier_cycle) * fs) % two_pi
            rot[i] = numpy.cos(delta), numpy.sin(delta)
        return rot

    def vsign(self, frame, line):
        if self.backend.v_switch and self.lc.is_alternate_line(frame, line):
            return -1.0
        return 1.0

    def q_row(self, j, d):
        """qam.demodulate(phi_k + d, x_{k-j}) in terms of the base pairs (lines of one run are LS apart)."""
        return rot(base(j), d + j * self.LS)

    # -- base decoders: (u, v) of backend.demodulate_components at call k --------------------
    def uv_plain(self, frame, line):
        s, c = base(0)
        return s, self.vsign(frame, line) * c

    def uv_ntsc_comb(self, frame, line, k):
        if k == 0:  # comb.py:48-49 -> ntsc.py:47-49
            return self.uv_plain(frame, line)
        # ntsc.py:74-81: demodulate (curr - last) half a line shift back, swap, scale
        f = self.comb._factor
        if not numpy.isfinite(f):   # ntsc.py:62-63: lines nearly in phase, the comb falls back to the plain decode
            return self.uv_plain(frame, line)
        q = pair_sub(self.q_row(0, -0.5 * self.LS), self.q_row(1, -0.5 * self.LS))
        return f * q[1], -f * q[0]

    def uv_pal_d(self, frame, line, k):
        if k == 0:
            raise AssertionError('first line of a PAL-D run is decoded by the plain pass')
        # base pair here is the PAL-D front end at theta = phi + ps - LS/2 (pal.py:113-114)
        p0, p1 = base(0), rot(base(1), self.LS)
        total = p0[0] + p1[0]          # AM(sum, sin carrier), pal.py:116,119
        diff = p0[1] - p1[1]           # AM(diff, cos carrier), pal.py:117,120
        sf, cf = self.comb._sin_factor, self.comb._cos_factor
        u = sf * diff + cf * total     # pal.py:122
        v = cf * diff - sf * total     # pal.py:123
        return u, self.vsign(frame, line) * v  # pal.py:124-125

    def uv_pal_3d(self, frame, line, k):
        c3 = self.comb
        if k == 0:                     # pal.py:191-195 (plain decode, luma left unstripped)
            return self.uv_plain(frame, line)
        if k == 1:                     # pal.py:199-201: replay the first line's own decode
            u, v = self.uv_plain(frame, line - 2)
            return u.shifted(1), v.shifted(1)
        # pal.py:203-207: demodulate at the phase of line - 2
        d = -self.LS
        x0, x1, x2 = self.q_row(0, d), self.q_row(1, d), self.q_row(2, d)
        sumsig = pair_sub(x0, x2)                                  # curr_diff + last_diff
        diffsig = pair_add(pair_sub(x0, (2.0 * x1[0], 2.0 * x1[1])), x2)  # curr_diff - last_diff
        if c3._use_sin and c3._use_cos and self.minavg:            # pal.py:209-211 with comb.minavg: two estimates
            s = self.vsign(frame, line - 2)                        # pal.py:219-220 (minavg is odd: the sign commutes)
            return ((c3._sin_sum_factor * sumsig[1], s * (c3._sin_sum_factor * sumsig[0])),
                    (c3._cos_u_factor * diffsig[0], s * (c3._cos_v_factor * diffsig[1])))
        if c3._use_sin and c3._use_cos:                            # pal.py:209-211 (arithmetic mean)
            u = 0.5 * (c3._sin_sum_factor * sumsig[1] + c3._cos_u_factor * diffsig[0])
            v = 0.5 * (c3._sin_sum_factor * sumsig[0] + c3._cos_v_factor * diffsig[1])
        elif c3._use_sin:
            u, v = c3._sin_sum_factor * sumsig[1], c3._sin_sum_factor * sumsig[0]
        else:
            u, v = c3._cos_u_factor * diffsig[0], c3._cos_v_factor * diffsig[1]
        return u, self.vsign(frame, line - 2) * v                  # pal.py:219-220

    def backend_uv(self, frame, line, k):
        if self.kind in ('pal_s', 'ntsc'):
            return self.uv_plain(frame, line)
        if self.kind == 'ntsc_comb':
            return self.uv_ntsc_comb(frame, line, k)
        if self.kind == 'pal_d':
            return self.uv_pal_d(frame, line, k)
        if self.kind == 'pal_3d':
            return self.uv_pal_3d(frame, line, k)
        raise NotImplementedError(self.kind)

    # -- full decoder at call k ---------------------------------------------------------------
    def decode(self, frame, line, k):
        """Returns u, v, remod_line (None: luma unstripped or band-stop), luma_prev (calls back: bool or 0 / 1 / 2), u2, v2.
        u2, v2 are None unless the decoder combines two estimates with comb.minavg: then the output is
        minavg(u, u2), minavg(v, v2)."""
        u, v, remod_line, luma_prev, u2, v2 = self._decode(frame, line, k)
        if not self.strip_chroma:
            remod_line = None   # qam.py:55-57, comb.py:52, 105, pal.py:225: luma stays the (previous) composite row
        return u, v, remod_line, luma_prev, u2, v2

    def _decode(self, frame, line, k):
        w = self.demod_wrapper
        if w is None:
            if self.kind in ('pal_s', 'ntsc'):
                u, v = self.uv_plain(frame, line)
                return u, v, None, False, None, None
            if self.kind in ('ntsc_comb', 'pal_d'):
                u, v = self.backend_uv(frame, line, k)
                return u, v, line, False, None, None  # comb.py:52-53
            if self.kind == 'pal_3d':
                r = self.uv_pal_3d(frame, line, k)
                u2 = v2 = None
                if self.minavg and k >= 2:
                    (u, v), (u2, v2) = r
                else:
                    u, v = r
                if k == 0:
                    return u, v, None, False, None, None   # pal.py:195: returned before the strip
                return u, v, line - 2, True, u2, v2         # pal.py:223,226
        else:
            own_delay = 1 if w == 'simple_3d' else 0
            if self.kind == 'pal_3d':
                if not self.two_level:
                    raise NotImplementedError('SimpleCombModem around Pal3DModem reaches back 3 lines: wrapped.py composes it (and runs long '
                                              'batches as a two-level comb)')
                # the inner decoder's own combination at this call (pal.py:180-234, strip_chroma=False); the wrapper level is the kernel's
                r = self.uv_pal_3d(frame, line, k)
                u2 = v2 = None
                if self.minavg and k >= 2:
                    (u, v), (u2, v2) = r
                else:
                    u, v = r
                if k == 0:                                # comb.py:97-99: the inner result as it is, luma unstripped
                    return u, v, None, 0, u2, v2
                # comb.py:102: y = the previous call's inner luma (own_delay) or this call's; Pal3DModem's luma at its call j is the row
                # of call 0 for j <= 1 (pal.py:191-201) and of call j - 1 after that (pal.py:223)
                back = (1 if k == 1 else 2) if own_delay else 1
                return u, v, line - 2 * own_delay, back, u2, v2      # comb.py:105-106 (modulation_delay = 0)
            if self.kind == 'pal_d':
                if not self.fused_main:
                    raise NotImplementedError('SimpleCombModem around PalDModem mixes two front ends on the first two calls of a run; '
                                              'wrapped.py composes it (and fuses the calls k >= 2)')
                if k < 2:         # never stored (skip_calls = 2): no combination to tabulate
                    return Lin(), Lin(), None, False, None, None
            cu, cv = self.backend_uv(frame, line, k)
            if k == 0:                                # comb.py:97-99: luma left unstripped
                return cu, cv, None, False, None, None
            lu, lv = self.backend_uv(frame, line - 2, k - 1)
            remod, luma_prev = line - 2 * own_delay, bool(own_delay)   # comb.py:102,106 (modulation_delay = 0)
            if self.minavg:                           # comb.py:103-104 with comb.minavg
                return lu.shifted(1), lv.shifted(1), remod, luma_prev, cu, cv
            u = 0.5 * (lu.shifted(1) + cu)            # comb.py:103-104
            v = 0.5 * (lv.shifted(1) + cv)
            return u, v, remod, luma_prev, None, None
        raise NotImplementedError((self.kind, w))

    @property
    def demodulation_delay(self):
        d = 1 if self.kind == 'pal_3d' else 0
        if self.demod_wrapper == 'simple_3d':
            d += 1
        return d

    @property
    def modulation_delay(self):
        return 1 if self.mod_wrapper == 'color_averaging' else 0

    @property
    def depth(self):
        d = {'pal_s': 0, 'ntsc': 0, 'pal_d': 1, 'ntsc_comb': 1, 'pal_3d': 2}[self.kind]
        if self.demod_wrapper:
            d += 1
        return d

    @property
    def plain_stack(self):
        # luma from the band-stop path (qam.py:57); with strip_chroma=False luma is the composite row itself
        return self.demod_wrapper is None and self.kind in ('pal_s', 'ntsc') and self.strip_chroma

    @property
    def first_is_plain(self):
        # k == 0 goes through backend.demodulate_components(strip_chroma=True): band-stop luma
        return self.demod_wrapper is None and self.kind in ('pal_d', 'ntsc_comb')

    def detector_phase(self, frame, line):
        theta = self.phi(frame, line) + self.ps
        if self.kind == 'pal_d':
            theta -= 0.5 * self.LS
        return theta

    def n_lines(self):
        # the frame entry points need height + 2 delay lines; the per-row protocol takes any line number the caller
        # passes (the reference's size only sets fs and the line shift): min_lines grows the tables for it
        return max(self.height + 2 * max(self.demodulation_delay, self.modulation_delay) + 4, self.min_lines) + self.line_offset

    def phase_free(self, lin, frame, line):
        """Re-express a combination over the base pairs B_{k-j} (detected at each line's own phase theta_j)
        over the phase-free pairs R_{k-j}:  Bs = cos(theta) Rs + sin(theta) Rc,  Bc = -sin(theta) Rs + cos(theta) Rc."""
        out = numpy.zeros((3, 2))
        for j in range(3):
            if not lin.c[j].any():
                continue
            th = self.detector_phase(frame, line - 2 * j)
            c0, c1 = lin.c[j]
            out[j, 0] = c0 * numpy.cos(th) - c1 * numpy.sin(th)
            out[j, 1] = c0 * numpy.sin(th) + c1 * numpy.cos(th)
        return out

    def demod_main_table(self):
        n_lines = self.n_lines()
        tab = numpy.zeros((self.table_frames, 3, n_lines, CM_LANE_DOUBLES))
        luma_prev_bits = 0
        for f in range(self.table_frames):
            for k in range(3):
                for line in range(n_lines):
                    if line - 2 * k < -1:
                        continue  # a run never starts above the top of the frame
                    e = tab[f, k, line]
                    theta = self.detector_phase(f, line)
                    e[0], e[1] = numpy.sin(theta), numpy.cos(theta)
                    e[16] = 1.0
                    if self.first_is_plain and k == 0:
                        continue  # output produced by the plain pass; the base pair still feeds call 1
                    u, v, remod_line, luma_prev, u2, v2 = self.decode(f, line, k)
                    if remod_line is not None:
                        p = self.phi(f, remod_line)
                        e[2], e[3] = numpy.sin(p), numpy.cos(p)
                        e[16] = self.vsign(f, remod_line)  # pal.py:50-51
                    e[4:10] = self.phase_free(u, f, line).reshape(-1)
                    e[10:16] = self.phase_free(v, f, line).reshape(-1)
                    if self.minavg:   # no second estimate at this call: minavg(a, a) = a
                        e[20:26] = self.phase_free(u2 if u2 is not None else u, f, line).reshape(-1)
                        e[26:32] = self.phase_free(v2 if v2 is not None else v, f, line).reshape(-1)
                    if int(luma_prev) >= 1:     # bit k: the luma source lies one call back, bit 8 + k: one more (two-level combs)
                        luma_prev_bits |= 1 << k
                    if int(luma_prev) >= 2:
                        luma_prev_bits |= 1 << (8 + k)
        return tab, luma_prev_bits

    def demod_first_table(self):
        n_lines = self.n_lines()
        tab = numpy.zeros((self.table_frames, 3, n_lines, CM_LANE_DOUBLES))
        for f in range(self.table_frames):
            for line in range(n_lines):
                u, v = self.uv_plain(f, line)
                e = tab[f, 0, line]
                theta = self.phi(f, line) + self.ps
                e[0], e[1] = numpy.sin(theta), numpy.cos(theta)
                for lin, lo in ((u, 4), (v, 10)):   # plain decoder: the QAM front end's phase, not the PAL-D one
                    c0, c1 = lin.c[0]
                    e[lo] = c0 * numpy.cos(theta) - c1 * numpy.sin(theta)
                    e[lo + 1] = c0 * numpy.sin(theta) + c1 * numpy.cos(theta)
                e[16] = 1.0
        return tab

    def mod_table(self):
        """[0] sin, [1] cos of the start phase of the modulated line; [2..5] row weights (luma: current,
        previous; chroma: current, previous), ref comb.py:141-152; [6] V-switch sign (pal.py:50-51)."""
        n_lines = self.n_lines()
        tab = numpy.zeros((self.table_frames, 3, n_lines, CM_LANE_DOUBLES))
        avg = self.mod_wrapper == 'color_averaging'
        for f in range(self.table_frames):
            for k in range(3):
                for line in range(n_lines):
                    target = (line - self.line_offset) - 2 if avg else line - self.line_offset
                    p = self.phi(f, target)
                    e = tab[f, k, line]
                    e[0] = numpy.sin(p)
                    e[1] = numpy.cos(p)
                    e[6] = self.vsign(f, target)
                    if avg and k >= 1:
                        e[2:6] = (0.0, 1.0, 0.5, 0.5)
                    else:
                        e[2:6] = (1.0, 0.0, 1.0, 0.0)
        return tab


class BuiltPlan(object):
    """PlanDesc plus the numpy arrays it points into (kept alive here)."""

    def __init__(self, desc, keep, tables):
        self.desc = desc
        self._keep = keep
        self.tables = tables


def _lane_table(arr, luma_prev_bits=0):
    t = LaneTable()
    if arr is None:
        t.frame_cycle, t.n_lines, t.table = 0, 0, None
        return t
    t.frame_cycle, t.n_lines = arr.shape[0], arr.shape[2]
    t.table = arr.ctypes.data_as(ctypes.POINTER(ctypes.c_double))
    t.luma_from_prev = luma_prev_bits
    return t


def build_qam_plan(stack, components=False, strip_chroma=True, min_lines=0):
    tb = QamTables(stack, strip_chroma, min_lines)
    b = tb.backend
    d = PlanDesc()
    d.abi_version = CM_ABI_VERSION
    d.pipeline = CM_PIPE_PAL_D if tb.kind == 'pal_d' else CM_PIPE_QAM
    d.width, d.height = tb.width, tb.height
    d.demodulation_delay = tb.demodulation_delay
    d.modulation_delay = tb.modulation_delay
    d.depth = tb.depth
    d.first_is_plain = 1 if tb.first_is_plain else 0
    d.skip_calls = 2 if tb.fused_main else 0
    d.main_luma_bandstop = 1 if tb.plain_stack else 0
    d.carrier_phase_step = float(b.qam.carrier_phase_step)
    d.resample_fir[:] = list(resample_fir())
    d.extract2x = iir_desc(b.qam._extract_chroma2x, bandpass=True)
    d.remove2x = iir_desc(b.qam._remove_chroma2x)
    d.demod_lp = iir_desc(b.qam._demod_lowpass)
    d.pald_lp = iir_desc(tb.comb._filter if tb.kind == 'pal_d' else None)
    d.precorrect = iir_desc(b.qam._chroma_precorrect_lowpass)
    # components=True: the *_components protocol - (y, u, v) cross the boundary instead of (r, g, b)
    eye = numpy.eye(3)
    d.decode_matrix[:] = list(numpy.asarray(eye if components else b.decode_matrix).reshape(-1))
    d.encode_matrix[:] = list(numpy.asarray(eye if components else b.encode_matrix).reshape(-1))
    main, bits = tb.demod_main_table()
    first = tb.demod_first_table() if tb.first_is_plain else None
    mod = tb.mod_table()
    main = numpy.ascontiguousarray(main)
    mod = numpy.ascontiguousarray(mod)
    d.demod_main = _lane_table(main, bits)
    d.demod_main.wrap_mode = tb.wrap_mode          # 0 / 1: avg / 2: minavg of consecutive calls' results in the kernel (two-level comb)
    d.demod_first = _lane_table(first)
    d.mod_main = _lane_table(mod)
    rot = None
    if tb.rotating:
        rot = numpy.ascontiguousarray(tb.frame_rotation())
        d.frame_rotation = rot.ctypes.data_as(ctypes.POINTER(ctypes.c_double))
        d.frame_rotation_cycle = tb.rotation_cycle
    d.chroma_average = CM_AVG_MIN if tb.minavg else CM_AVG_FOLDED
    # comb.py:52-55 / 107-110 / pal.py:225-228: the notch follows the chroma strip of whichever layer strips; a
    # wrapped comb is called with strip_chroma=False, so only the wrapper's notch acts then
    notch = stack.get('wrapper_notch') if tb.demod_wrapper else stack.get('comb_notch')
    if not strip_chroma:
        notch = None
    if notch is not None and notch.shift != 0:      # (engine.make_engine sends such stacks to notched.ShiftedNotchEngine, which builds this plan without the notch)
        raise NotImplementedError('the fused kernels carry the notch at FilterFunction shift 0; shift %d goes through color_modem_amd/notched.py' % notch.shift)
    d.notch = iir_desc(notch)
    return BuiltPlan(d, [main, first, mod, rot], tb)


def build_plan(modem, components=False, strip_chroma=True, min_lines=0):
    """components: (y, u, v) instead of (r, g, b) at the boundary (modulate_components / demodulate_components);
    strip_chroma: the flag of demodulate_components (False: luma is returned unstripped); min_lines: line numbers
    0 .. min_lines - 1 get table entries even beyond the image height (the per-row protocol)."""
    stack = modem._stack()
    if stack['kind'] == 'secam':
        from color_modem_amd import plan_secam
        return plan_secam.build_secam_plan(stack, components, min_lines)
    return build_qam_plan(stack, components, strip_chroma, min_lines)
