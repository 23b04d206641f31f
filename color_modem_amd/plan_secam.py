# -*- coding: utf-8 -*-
"""Plan descriptor of the SECAM stacks (SecamModem, ColorAveragingModem(SecamModem))."""

import numpy

from color_modem_amd import plan


def build_secam_plan(stack, components=False, min_lines=0):
    m = stack['backend']
    if stack.get('demod_wrapper'):
        # the reference would fail at the first row: SecamModem has no demodulate_components
        raise AttributeError("'SecamModem' object has no attribute 'demodulate_components'")
    avg = stack.get('mod_wrapper') == 'color_averaging'
    lc = m.line_config
    width, height = lc.size
    d = plan.PlanDesc()
    d.abi_version = plan.CM_ABI_VERSION
    d.pipeline = plan.CM_PIPE_SECAM
    d.width, d.height = width, height
    d.demodulation_delay = 0
    d.modulation_delay = 1 if avg else 0
    d.depth = 1                      # last_chroma: one line of history (secam.py:297-300)
    d.first_is_plain = 0
    d.main_luma_bandstop = 0
    d.resample_fir[:] = list(plan.resample_fir())
    eye = numpy.eye(3)   # components: (luma, dr, db) cross the boundary (secam.py:258 modulate_components)
    d.decode_matrix[:] = list(numpy.asarray(eye if components else m.decode_matrix).reshape(-1))
    d.encode_matrix[:] = list(numpy.asarray(eye if components else m.encode_matrix).reshape(-1))
    s = d.secam
    s.present = plan.CM_SECAM_PRESENT | (plan.CM_SECAM_FLOAT64 if getattr(m, 'float64_front_end', False) else 0)
    s.preroll = width // 40 - 1
    if s.preroll < 0:
        raise NotImplementedError('rows shorter than 40 samples have no chroma pre-roll')
    s.flimit_min, s.flimit_max, s.bell_f0 = m._flimit_min, m._flimit_max, m._bell_f0
    s.m0, s.bell_kn, s.bell_kd = m._variant.m0, m._variant.bell_kn, m._variant.bell_kd
    s.fm_fc = m._chroma_demod._fc
    s.pre_lp = plan.iir_desc(m._chroma_precorrect_lowpass)
    s.lf_pre = plan.iir_desc(m._chroma_precorrect)
    s.lf_rev = plan.iir_desc(m._reverse_chroma_precorrect)
    s.bell = plan.iir_desc(m._chroma_demod_bell, bandpass=True)
    s.chroma_bp = plan.iir_desc(m._chroma_demod_chroma_filter, bandpass=True)
    s.luma_bs = plan.iir_desc(m._chroma_demod_luma_filter)
    s.fm_lp = plan.iir_desc(m._chroma_demod._lowpass)

    offset = int(stack.get('line_offset', 0))     # the modulator table's row i describes line i - offset (plan.QamTables: line_offset)
    n_lines = max(height + 2 * d.modulation_delay + 4, int(min_lines)) + offset
    demod = numpy.zeros((2, 3, n_lines, plan.CM_LANE_DOUBLES))
    for f in range(2):
        for k in range(3):
            for line in range(n_lines):
                alt = lc.is_alternate_line(f, line)
                e = demod[f, k, line]
                e[0] = m._fsc_db if alt else m._fsc_dr
                e[1] = m._fdev_db if alt else m._fdev_dr
                e[2] = 1.0 if alt else 0.0
                e[3] = 0.0 if k == 0 else 1.0
    mod = numpy.zeros((6, 3, n_lines, plan.CM_LANE_DOUBLES))
    for f in range(6):
        for k in range(3):
            for line in range(n_lines):
                target = (line - offset) - 2 if avg else line - offset      # comb.py:152
                alt = lc.is_alternate_line(f, target)
                e = mod[f, k, line]
                e[0] = m._fsc_db if alt else m._fsc_dr
                e[1] = m._fdev_db if alt else m._fdev_dr
                e[2] = 1.0 if alt else 0.0
                e[3] = numpy.pi if m._start_phase_inverted(f, target) else 0.0
                e[4:8] = (0.0, 1.0, 0.5, 0.5) if (avg and k >= 1) else (1.0, 0.0, 1.0, 0.0)
    demod = numpy.ascontiguousarray(demod)
    mod = numpy.ascontiguousarray(mod)
    d.demod_main = plan._lane_table(demod)
    d.demod_first = plan._lane_table(None)
    d.mod_main = plan._lane_table(mod)
    return plan.BuiltPlan(d, [demod, mod], None)
