# -*- coding: utf-8 -*-
"""cm_am_desc (include/color_modem_hip.h) of the amplitude-modulated line-sequential stacks: ProtoSecamModem,
ColorAveragingModem(ProtoSecamModem), NiirModem, HueCorrectingNiirModem."""

import ctypes

import numpy
from color_modem_amd import design

from color_modem_amd import plan

CM_AM_FLOAT64 = 1        # cm_am_desc.flags (include/color_modem_hip.h)
CM_AM_PROTO_SECAM, CM_AM_NIIR = 1, 2


class AmDesc(ctypes.Structure):
    """cm_am_desc"""
    _fields_ = [('abi_version', ctypes.c_int32), ('kind', ctypes.c_int32), ('width', ctypes.c_int32), ('height', ctypes.c_int32),
                ('line_shift', ctypes.c_int32), ('even_first', ctypes.c_int32), ('odd_first', ctypes.c_int32),
                ('averaging', ctypes.c_int32), ('premod_luma_filter', ctypes.c_int32), ('frame_cycle', ctypes.c_int32),
                ('strip_chroma', ctypes.c_int32), ('flags', ctypes.c_int32),
                ('frame_phase_shift', ctypes.c_double), ('line_phase_shift', ctypes.c_double),
                ('carrier_phase_step', ctypes.c_double), ('resample_fir3', ctypes.c_double * 61),
                ('precorrect', plan.IirDesc), ('bandpass_up', plan.IirDesc), ('bandstop_up', plan.IirDesc),
                ('lowpass_up', plan.IirDesc), ('bandpass_phase_shift', ctypes.c_double),
                ('decode_matrix', ctypes.c_double * 9), ('encode_matrix', ctypes.c_double * 9)]


def resample_fir3():
    """the filter scipy.signal.resample_poly designs for up / down = 3 (protosecam.py:83, 85, 96, 101, 102; niir.py:109 ...)"""
    return design.resample_poly_fir(3)


def build_am_desc(modem, components=False, strip_chroma=True):
    stack = modem._stack()
    kind = stack['kind']
    m = stack['backend']
    if stack.get('demod_wrapper'):
        raise NotImplementedError('comb wrappers around %s are not built' % type(m).__name__)
    lc = m.line_config
    std = lc.line_standard
    d = AmDesc()
    d.abi_version = plan.CM_ABI_VERSION
    d.width, d.height = int(lc.size[0]), int(lc.size[1])
    d.line_shift = int(lc._line_shift)
    d.even_first = int(std.even_field_first_active_line)
    d.odd_first = int(std.odd_field_first_active_line)
    d.frame_cycle = int(m.frame_cycle)
    d.strip_chroma = 1 if strip_chroma else 0
    d.frame_phase_shift = float(m.frame_shift)
    d.line_phase_shift = float(m.line_shift)
    d.resample_fir3[:] = list(resample_fir3())
    eye = numpy.eye(3)
    if kind == 'protosecam':
        from color_modem_amd.color import protosecam
        d.kind = CM_AM_PROTO_SECAM
        d.averaging = 1 if stack.get('mod_wrapper') == 'color_averaging' else 0
        d.premod_luma_filter = 1 if m._premod_luma_filter else 0
        d.carrier_phase_step = 2.0 * float(m._carrier_phase_step)          # protosecam.py:88: 2 * _carrier_phase_step per sample
        d.precorrect = plan.iir_desc(m._chroma_precorrect_lowpass)
        d.bandpass_up = plan.iir_desc(m._extract_chroma_up, bandpass=True)
        d.bandstop_up = plan.iir_desc(m._remove_chroma_up)
        d.lowpass_up = plan.iir_desc(m._chroma_up_post_demod_filter)
        dec, enc = protosecam.DECODE, protosecam.ENCODE
    elif kind == 'niir':
        from color_modem_amd.color import niir
        if stack.get('mod_wrapper'):
            raise NotImplementedError('ColorAveragingModem around NiirModem is not built (HueCorrectingNiirModem is its own averaging encoder)')
        d.kind = CM_AM_NIIR
        d.averaging = 1 if stack.get('hue_correcting') else 0
        d.carrier_phase_step = float(m._carrier_phase_step)
        d.precorrect = plan.iir_desc(m._chroma_precorrect_lowpass)
        d.bandpass_up = plan.iir_desc(m._demodulate_upsampled_filter, bandpass=True)
        d.bandstop_up = plan.iir_desc(None)
        d.lowpass_up = plan.iir_desc(m._demodulate_upsampled_baseband_filter)
        d.bandpass_phase_shift = float(m._demodulate_upsampled_filter.phase_shift)
        # (modem.float64_front_end, round 3's opt-in: every NIIR decoder runs its hue path in float64 now; the flag is accepted and ignored)
        d.flags = CM_AM_FLOAT64 if (getattr(modem, 'float64_front_end', False) or getattr(m, 'float64_front_end', False)) else 0
        dec, enc = niir.DECODE, niir.ENCODE
    else:
        raise ValueError(kind)
    d.decode_matrix[:] = list(numpy.asarray(eye if components else dec).reshape(-1))
    d.encode_matrix[:] = list(numpy.asarray(eye if components else enc).reshape(-1))
    return d
