# -*- coding: utf-8 -*-
"""NIIR / SECAM-IV: amplitude + phase-reference line-sequential colour (API mirror of
/root/reference/color_modem/color/niir.py:10-202).

Host side: constructor arguments, colour matrices and the three filter designs (the same scipy calls as the reference).
The per-line work runs in ``niir_demod_pair_kernel`` / ``niir_mod_kernel`` (csrc/cm_am_kernels.h; small batches: the scan
kernels of csrc/cm_am_scan_kernels.h) behind ``cm_am_*``; the decoder's hue path is float64 (csrc/cm_am_stages.h: NiirHue).
``noise_level`` other than 0 draws from numpy.random in the reference (niir.py:45-46, 189-191): the engine draws the same samples
in the reference's call order and hands them to the kernel as a plane, so ``numpy.random.seed`` reproduces the reference.
"""

import numpy

from color_modem_amd import utils
from color_modem_amd.color import pal
from color_modem_amd.rowapi import RowApi

# (luma, db, dr) = ENCODE . (r, g, b)   ref niir.py:31-40
ENCODE = numpy.array([[0.299, 0.587, 0.114],
                      [0.1472906403940887, 0.2891625615763547, -0.4364532019704434],
                      [0.6149122807017545, -0.5149122807017544, -0.1]])
# (r, g, b) = DECODE . (luma, db, dr)   ref niir.py:52-61
DECODE = numpy.array([[1.0, 0.0, 1.14],
                      [1.0, 0.3942419080068143, -0.5806814310051107],
                      [1.0, -2.03, 0.0]])

RESAMPLE_FACTOR = 3


class NiirModem(utils.ConstantFrequencyCarrier, RowApi):
    hue_correcting = False

    def __init__(self, line_config, config=pal.PalVariant.PAL, noise_level=0.0):
        RowApi.__init__(self)
        # niir.py:45-46 / 193-194: hue noise from numpy.random, drawn by the engine in the reference's call order (db, then dr,
        # per modulate() call) - numpy.random.seed() reproduces the reference's output
        self._noise_level = float(noise_level)
        self.line_config = line_config
        self.config = config
        fs = line_config.fs
        self._carrier_phase_step = 2.0 * numpy.pi * config.fsc / fs
        self._demodulate_resample_factor = RESAMPLE_FACTOR
        # ref niir.py:17-24, 95-98
        self._chroma_precorrect_lowpass = utils.iirdesign(2.0 * config.bandwidth3db / fs, 2.0 * config.bandwidth20db / fs, 3.0, 20.0)
        wc, wp, ws = 2.0 * config.fsc / fs, 2.0 * config.bandwidth3db / fs, 2.0 * config.bandwidth20db / fs
        self._demodulate_upsampled_baseband_filter = utils.iirdesign(wp / RESAMPLE_FACTOR, ws / RESAMPLE_FACTOR, 3.0, 20.0)
        self._demodulate_upsampled_filter = utils.iirdesign_wc(wc / RESAMPLE_FACTOR, wp / RESAMPLE_FACTOR, ws / RESAMPLE_FACTOR,
                                                               3.0, 20.0)

    @staticmethod
    def encode_components(r, g, b):
        assert len(r) == len(g) == len(b)
        # Term by term in the reference's own order (niir.py:34-36), not as a matrix product: on grey and nearly grey pixels (db, dr) are
        # rounding residues of these very sums, and the hue of the 0.1 pedestal (niir.py:42-49) is their angle.
        r, g, b = [numpy.asarray(c, dtype=numpy.float64) for c in (r, g, b)]
        luma = ENCODE[0, 0] * r + ENCODE[0, 1] * g + ENCODE[0, 2] * b
        db = ENCODE[1, 0] * r + ENCODE[1, 1] * g - (-ENCODE[1, 2]) * b
        dr = ENCODE[2, 0] * r - (-ENCODE[2, 1]) * g - (-ENCODE[2, 2]) * b
        return luma, db, dr

    @staticmethod
    def decode_components(luma, db, dr):
        assert len(luma) == len(db) == len(dr)
        r, g, b = DECODE @ numpy.stack([numpy.asarray(c, dtype=numpy.float64) for c in (luma, db, dr)])
        return r, g, b

    def _stack(self):
        return {'kind': 'niir', 'backend': self, 'hue_correcting': self.hue_correcting}


class HueCorrectingNiirModem(NiirModem):
    """Encoder-side hue averaging with the previous line (ref niir.py:167-202): modulation_delay 1."""
    hue_correcting = True

    def __init__(self, *args, **kwargs):
        super(HueCorrectingNiirModem, self).__init__(*args, **kwargs)
        self.modulation_delay = 1
