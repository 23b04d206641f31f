# -*- coding: utf-8 -*-
"""float64 restatement of the reference's amplitude-modulated line-sequential standards - TEST INFRASTRUCTURE.

Follows /root/reference/color_modem/color/protosecam.py (ProtoSecamModem) and niir.py (NiirModem,
HueCorrectingNiirModem) in plain numpy: ``resample_poly`` written out (cm_oracle_mac.resample_poly), ``lfilter`` as the
direct-form recurrence, ``FilterFunction.__call__`` (utils.py:28-36) with its tail padding.  The filter coefficients are the
oracle's own scipy designs (oracle/cm_oracle_design.py: the reference's design calls restated); the host classes of color_modem_amd supply the variant
constants and the line geometry only.  Pinned against vectors the
reference itself produced (tests/golden/am_*.npz, made by tests/golden/make_golden_am.py) in tests/test_am_oracle.py.
May be imported only by tests/ and by tools run by hand - never by color_modem_amd.
"""

import numpy

from oracle import cm_oracle_design as design
from oracle.cm_oracle_mac import resample_poly


def lfilter(b, a, x):
    """scipy.signal.lfilter(b, a, x) from a zero state (direct form II transposed, like scipy)."""
    b = numpy.asarray(b, dtype=numpy.float64) / a[0]
    a = numpy.asarray(a, dtype=numpy.float64) / a[0]
    n = max(len(a), len(b))
    b = numpy.concatenate([b, numpy.zeros(n - len(b))])
    a = numpy.concatenate([a, numpy.zeros(n - len(a))])
    z = numpy.zeros(n)
    y = numpy.empty(len(x))
    for i, xi in enumerate(numpy.asarray(x, dtype=numpy.float64)):
        yi = b[0] * xi + z[0]
        for j in range(1, n):
            z[j - 1] = b[j] * xi + z[j] - a[j] * yi
        y[i] = yi
    return y


try:    # the same recurrence, compiled (scipy is test infrastructure too; the loop above is the definition)
    import scipy.signal as _sig

    def lfilter(b, a, x):   # noqa: F811
        return _sig.lfilter(b, a, numpy.asarray(x, dtype=numpy.float64))
except ImportError:   # pragma: no cover
    pass


def apply_filter(f, x):
    """FilterFunction.__call__ (utils.py:28-36): the tail is padded with `shift` copies of the last sample, the first
    `shift` outputs are dropped."""
    x = numpy.asarray(x, dtype=numpy.float64)
    s = int(f.shift)
    if s == 0:
        return lfilter(f.b, f.a, x)
    if s > 0:
        return lfilter(f.b, f.a, numpy.concatenate((x, x[-1] * numpy.ones(s))))[s:]
    return lfilter(f.b, f.a, numpy.concatenate((x[0] * numpy.ones(-s), x)))[:s]


def _phase_ramp(start, step, n):
    """numpy.linspace(start, start + n step, n, endpoint=False) % 2 pi"""
    return numpy.linspace(start, start + n * step, n, endpoint=False) % (2.0 * numpy.pi)


class ProtoSecamOracle(object):
    """protosecam.py:27-112 on a color_modem_amd.color.protosecam.ProtoSecamModem (filter designs, geometry); the colour matrices are restated here
    term by term in the reference's operation order (protosecam.py:54-69), not taken from the product."""

    @staticmethod
    def encode_components(r, g, b):                                                      # protosecam.py:54-60
        r, g, b = [numpy.asarray(c, dtype=numpy.float64) for c in (r, g, b)]
        luma = 0.3 * r + 0.59 * g + 0.11 * b
        dr = 1.001 * r - 0.8437 * g - 0.1573 * b
        db = -0.336 * r - 0.6608 * g + 0.9968 * b
        return luma, dr, db

    @staticmethod
    def decode_components(luma, dr, db):                                                 # protosecam.py:62-69
        r = luma + 0.6993006993006993 * dr
        g = luma - 0.3555766267630674 * dr - 0.1664648910411622 * db
        b = luma + 0.8928571428571429 * db
        return r, g, b

    def __init__(self, modem):
        self.m = modem
        # the oracle's own scipy designs of protosecam.py:32-50 (oracle/cm_oracle_design.py), not the product's filter objects
        self.f = design.proto_filters(modem.line_config.fs, modem.config)
        self._last_frame = -1
        self._last_line = -1
        self._last_chroma = None

    def modulate(self, frame, line, r, g, b):
        return self.modulate_components(frame, line, *self.encode_components(r, g, b))

    def modulate_components(self, frame, line, luma, dr, db):
        m = self.m
        chroma = db if m.line_config.is_alternate_line(frame, line) else dr          # protosecam.py:75-78
        chroma = 0.125 * (1.0 + apply_filter(self.f['pre'], chroma))      # :79-80
        if m._premod_luma_filter:                                                        # :82-85
            up = resample_poly(luma, 3, 1)
            luma = resample_poly(apply_filter(self.f['remove_up'], up), 1, 3)
        start = m.start_phase(frame, line)
        phase = _phase_ramp(start, 2.0 * self.f['carrier_phase_step'], len(chroma))            # :87-89
        return luma + numpy.cos(phase) * chroma

    def demodulate(self, frame, line, composite):
        m = self.m
        composite = numpy.asarray(composite, dtype=numpy.float64)
        if frame != self._last_frame or line != self._last_line + 2 or self._last_chroma is None:
            self._last_chroma = numpy.zeros(len(composite))                             # :93-94
        up = resample_poly(composite, 3, 1)                                              # :96
        chroma_up = apply_filter(self.f['extract_up'], up)
        chroma_up = 0.5 * numpy.pi * numpy.abs(chroma_up)                                # :98
        chroma_up = apply_filter(self.f['post_demod'], chroma_up)
        luma = resample_poly(apply_filter(self.f['remove_up'], up), 1, 3)                # :100-101
        chroma = 8.0 * resample_poly(chroma_up, 1, 3) - 1.0                               # :102-103
        if not m.line_config.is_alternate_line(frame, line):                             # :105-108
            self._last_chroma, dr, db = chroma, chroma, self._last_chroma
        else:
            self._last_chroma, dr, db = chroma, self._last_chroma, chroma
        self._last_frame, self._last_line = frame, line
        return self.decode_components(luma, dr, db)


class NiirOracle(object):
    """niir.py:10-202 on a color_modem_amd.color.niir.NiirModem / HueCorrectingNiirModem (noise_level: the reference's numpy.random draws); the
    colour matrices restated term by term in the reference's operation order (niir.py:31-61) - on grey pixels (db, dr) are the rounding residues
    of exactly these sums and the pedestal's hue is THEIR angle."""

    @staticmethod
    def encode_components(r, g, b):                                                      # niir.py:31-40
        r, g, b = [numpy.asarray(c, dtype=numpy.float64) for c in (r, g, b)]
        luma = 0.299 * r + 0.587 * g + 0.114 * b
        db = 0.1472906403940887 * r + 0.2891625615763547 * g - 0.4364532019704434 * b
        dr = 0.6149122807017545 * r - 0.5149122807017544 * g - 0.1 * b
        return luma, db, dr

    @staticmethod
    def decode_components(luma, db, dr):                                                 # niir.py:52-61
        r = luma + 1.14 * dr
        g = luma + 0.3942419080068143 * db - 0.5806814310051107 * dr
        b = luma - 2.03 * db
        return r, g, b

    def __init__(self, modem):
        self.m = modem
        # the oracle's own scipy designs of niir.py:11-23, 90-93 (oracle/cm_oracle_design.py), not the product's filter objects
        self.f = design.niir_filters(modem.line_config.fs, modem.config)
        self.hue_correcting = bool(getattr(modem, 'hue_correcting', False))
        self.modulation_delay = 1 if self.hue_correcting else 0
        self._last_frame = -1
        self._last_line = -1
        self._last_phasemod_up = None
        self._last_modulated_frame = -1
        self._last_modulated_line = -1
        self._last_luma = self._last_db = self._last_dr = None

    # ---- encoder ---------------------------------------------------------------------------------
    def _add_offset(self, db, dr, noise_level=None):                                     # niir.py:42-49
        if noise_level is None:
            noise_level = float(getattr(self.m, '_noise_level', 0.0))
        saturation = numpy.sqrt(db * db + dr * dr) + 0.1
        if noise_level != 0.0:      # the draws of niir.py:45-46, in its order (numpy.random.seed pins them)
            db = db + (numpy.random.random_sample(len(db)) - 0.5) * noise_level
            dr = dr + (numpy.random.random_sample(len(dr)) - 0.5) * noise_level
        hue = numpy.arctan2(db, dr)
        return saturation * numpy.sin(hue), saturation * numpy.cos(hue)

    def _modulate_precorrected_chroma(self, frame, line, updb, updr):                    # niir.py:69-76
        m = self.m
        start = m.start_phase(frame, line)
        n = len(updb)
        phase = _phase_ramp(start, self.f['carrier_phase_step'], n)
        if not m.line_config.is_alternate_line(frame, line):
            return updb * numpy.sin(phase) + updr * numpy.cos(phase)
        return -numpy.sqrt(updb * updb + updr * updr) * numpy.sin(phase)

    def _modulate_offset_components(self, frame, line, luma, db, dr):                    # niir.py:85-90
        m = self.m
        db = apply_filter(self.f['pre'], db)
        dr = apply_filter(self.f['pre'], dr)
        return luma + self._modulate_precorrected_chroma(frame, line, db, dr)

    def modulate(self, frame, line, r, g, b):
        luma, db, dr = self.encode_components(r, g, b)
        if not self.hue_correcting:
            return self._modulate_offset_components(frame, line, luma, *self._add_offset(db, dr))   # niir.py:78-80 (with noise)
        return self.modulate_components(frame, line, luma, db, dr)

    def modulate_components(self, frame, line, luma, db, dr):
        luma, db, dr = [numpy.asarray(c, dtype=numpy.float64) for c in (luma, db, dr)]
        if not self.hue_correcting:
            return self._modulate_offset_components(frame, line, luma, *self._add_offset(db, dr, 0.0))   # niir.py:82-83 (no noise)
        # niir.py:181-202
        if frame != self._last_modulated_frame or line != self._last_modulated_line + 2 or self._last_db is None \
                or self._last_dr is None:
            self._last_luma, self._last_db, self._last_dr = luma, db, dr
        self._last_luma, luma = luma, self._last_luma
        last_saturation = numpy.sqrt(self._last_db * self._last_db + self._last_dr * self._last_dr)
        saturation = numpy.sqrt(db * db + dr * dr)
        divisor = last_saturation + saturation
        divisor = numpy.where(divisor == 0.0, 1.0, divisor)
        avgdb = (self._last_db * last_saturation + db * saturation) / divisor
        avgdr = (self._last_dr * last_saturation + dr * saturation) / divisor
        level = float(getattr(self.m, '_noise_level', 0.0))
        if level != 0.0:                                                                   # niir.py:192-194
            avgdb = avgdb + (numpy.random.random_sample(len(db)) - 0.5) * level
            avgdr = avgdr + (numpy.random.random_sample(len(dr)) - 0.5) * level
        ep = last_saturation + 0.1
        hue = numpy.arctan2(avgdb, avgdr)
        dbep, drep = ep * numpy.sin(hue), ep * numpy.cos(hue)
        self._last_db, self._last_dr = db, dr
        self._last_modulated_frame, self._last_modulated_line = frame, line
        return self._modulate_offset_components(frame, line - 2, luma, dbep, drep)

    # ---- decoder ---------------------------------------------------------------------------------
    @staticmethod
    def _remove_offset(db, dr):                                                          # niir.py:63-67
        saturation = numpy.maximum(numpy.sqrt(db * db + dr * dr) - 0.1, 0.0)
        hue = numpy.arctan2(db, dr)
        return saturation * numpy.sin(hue), saturation * numpy.cos(hue)

    def demodulate(self, frame, line, composite):
        return self.decode_components(*self.demodulate_components(frame, line, composite))

    def demodulate_components(self, frame, line, composite, strip_chroma=True):
        luma, db, dr = self._demodulate_offset_components(frame, line, composite, strip_chroma)
        return (luma,) + self._remove_offset(db, dr)

    def _demodulate_offset_components(self, frame, line, composite, strip_chroma=True):  # niir.py:106-164
        m = self.m
        composite = numpy.asarray(composite, dtype=numpy.float64)
        n = len(composite)
        f_up, f_base = self.f['bandpass_up'], self.f['baseband_up']
        if frame != self._last_frame or line != self._last_line + 2 or self._last_phasemod_up is None:
            last_modulated = self._modulate_precorrected_chroma(frame, line - 2, numpy.ones(n), numpy.zeros(n))
            self._last_phasemod_up = apply_filter(f_up, resample_poly(last_modulated, 3, 1))
        modulated_up = apply_filter(f_up, resample_poly(composite, 3, 1))
        demod_up = 0.5 * numpy.pi * numpy.abs(modulated_up)
        saturation_up = apply_filter(f_base, demod_up)
        with numpy.errstate(divide='ignore', invalid='ignore'):
            phasemod_up = modulated_up / saturation_up
        alt = m.line_config.is_alternate_line(frame, line)
        if not alt:
            carrier_up, huemod_up, shift = self._last_phasemod_up, phasemod_up, m.line_shift
        else:
            carrier_up, huemod_up, shift = phasemod_up, self._last_phasemod_up, -m.line_shift
        shifted = 0.5 * (carrier_up[0:-1] + carrier_up[1:])
        altcarrier_up = numpy.concatenate((numpy.zeros(1), numpy.diff(shifted), numpy.zeros(1))) * 3.0 / self.f['carrier_phase_step']
        sinphi = resample_poly(huemod_up * carrier_up, 1, 3)
        cosphi = resample_poly(huemod_up * altcarrier_up, 1, 3)
        with numpy.errstate(divide='ignore', invalid='ignore'):
            normalizer = numpy.sqrt(cosphi * cosphi + sinphi * sinphi)
            self.last_normalizer = normalizer      # (for tests: the hue is the angle of this decimated pair - ill-conditioned where it is short)
            cosphi = cosphi / normalizer
            sinphi = sinphi / normalizer
        sinphi, cosphi = -cosphi * numpy.sin(shift) - sinphi * numpy.cos(shift), \
            sinphi * numpy.sin(shift) - cosphi * numpy.cos(shift)
        saturation = resample_poly(saturation_up, 1, 3)
        db = saturation * sinphi
        dr = saturation * cosphi
        luma = composite
        if strip_chroma:
            sincarrier = resample_poly(carrier_up, 1, 3)
            coscarrier = resample_poly(altcarrier_up, 1, 3)
            if not alt:
                phase_shift, u_signal, v_signal = m.line_shift, db, dr
            else:
                phase_shift, u_signal, v_signal = 0.0, -numpy.sqrt(db * db + dr * dr), 0.0
            phase_shift += numpy.pi - f_up.phase_shift
            u_signal, v_signal = (u_signal * numpy.cos(phase_shift) - v_signal * numpy.sin(phase_shift)), \
                                 (u_signal * numpy.sin(phase_shift) + v_signal * numpy.cos(phase_shift))
            luma = luma - (u_signal * sincarrier + v_signal * coscarrier)
        self._last_phasemod_up = phasemod_up
        self._last_frame, self._last_line = frame, line
        return luma, db, dr


def make(modem):
    kind = modem._stack()['kind']
    return ProtoSecamOracle(modem) if kind == 'protosecam' else NiirOracle(modem)


# ---- frames: the row schedule of image.py:47-55, 75-83 ------------------------------------------------------------
def modulate_frames(modem, rgb, first_frame=0):
    rgb = numpy.asarray(rgb, dtype=numpy.float64)
    n, _, height, width = rgb.shape
    out = numpy.zeros((n, height, width))
    for i in range(n):
        orc = make(modem)
        delay = getattr(orc, 'modulation_delay', 0)
        frame = first_frame + i
        for field in range(2):
            for y in range(field, 2 * delay, 2):
                orc.modulate(frame, y, rgb[i, 0, y], rgb[i, 1, y], rgb[i, 2, y])
            for y in range(field, height, 2):
                iy = y + 2 * delay
                while iy >= height:
                    iy -= 2
                out[i, y] = orc.modulate(frame, y + 2 * delay, rgb[i, 0, iy], rgb[i, 1, iy], rgb[i, 2, iy])
    return out


def demodulate_frames(modem, comp, first_frame=0):
    comp = numpy.asarray(comp, dtype=numpy.float64)
    n, height, width = comp.shape
    out = numpy.zeros((n, 3, height, width))
    for i in range(n):
        orc = make(modem)
        for field in range(2):
            for y in range(field, height, 2):
                r, g, b = orc.demodulate(first_frame + i, y, comp[i, y])
                out[i, 0, y], out[i, 1, y], out[i, 2, y] = r, g, b
    return out
