# -*- coding: utf-8 -*-
"""Line-standard geometry (host side).

Same public surface as the reference's ``color_modem/line.py`` (``LineStandard`` with its five
presets and ``detect``, ``LineConfig`` with ``fs``, ``analog_line`` and ``is_alternate_line``);
see /root/reference/color_modem/line.py:6-65.  Nothing here runs per pixel: the device plan
turns these into per-line phase/parity tables (color_modem_amd/plan.py).
"""

import collections

_FIELDS = ('frame_rate', 'total_lines',
           'odd_field_first_active_line', 'odd_field_last_active_line',
           'even_field_first_active_line', 'even_field_last_active_line',
           'total_width_factor')


class LineStandard(collections.namedtuple('LineStandard', _FIELDS)):
    """Timing of one analog scanning standard (ref line.py:6-39)."""
    __slots__ = ()

    def __new__(cls, *args, **kwargs):
        std = super(LineStandard, cls).__new__(cls, *args, **kwargs)
        odd = std.odd_field_last_active_line - std.odd_field_first_active_line
        even = std.even_field_last_active_line - std.even_field_first_active_line
        if odd < 0 or even < 0 or odd != even:
            raise AssertionError('fields must hold the same, non-negative number of lines')
        if std.active_lines > std.total_lines:
            raise AssertionError('more active lines than total lines')
        return std

    @property
    def active_lines(self):
        return (self.odd_field_last_active_line - self.odd_field_first_active_line
                + self.even_field_last_active_line - self.even_field_first_active_line + 2)

    @classmethod
    def presets(cls):
        return [v for v in vars(cls).values() if isinstance(v, cls)]

    @classmethod
    def detect(cls, active_lines):
        """Smallest preset that still holds `active_lines` lines (ref line.py:28-39)."""
        fitting = [std for std in cls.presets() if std.active_lines >= active_lines]
        if not fitting:
            raise IndexError('No supported line standard supports %d lines' % (active_lines,))
        smallest = min(std.active_lines for std in fitting)
        # among equally sized standards the reference ends up with the one defined last
        return [std for std in fitting if std.active_lines == smallest][-1]


LineStandard.BAIRD_405 = LineStandard(25.0, 405, 16, 203, 218, 405, 1.2)
LineStandard.NTSC_525 = LineStandard(30000.0 / 1001.0, 525, 21, 263, 283, 525, 858.0 / 720.0)
LineStandard.GERBER_625 = LineStandard(25.0, 625, 336, 623, 23, 310, 1.2)
LineStandard.FRENCH_819 = LineStandard(25.0, 819, 39, 407, 448, 816, 1.2)
LineStandard.BELGIAN_819 = LineStandard(25.0, 819, 437, 816, 27, 406, 1.2)


class LineConfig(object):
    """Image size bound to a line standard (ref line.py:49-65)."""

    def __init__(self, size, line_standard=None):
        if line_standard is None:
            line_standard = LineStandard.detect(size[1])
        self.size = (int(size[0]), int(size[1]))
        self.line_standard = line_standard
        self.fs = line_standard.frame_rate * line_standard.total_lines * size[0] * line_standard.total_width_factor
        self._line_shift = (line_standard.active_lines - size[1]) // 2

    def analog_line(self, digital_line):
        adjusted = digital_line + self._line_shift
        first = (self.line_standard.even_field_first_active_line if adjusted % 2 == 0
                 else self.line_standard.odd_field_first_active_line)
        return first + adjusted // 2

    def is_alternate_line(self, frame, line):
        return self.analog_line(line) % 2 == frame % 2
