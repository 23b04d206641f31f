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

# name, frames per second, lines per frame, active lines of the odd field (first, last), of the even field, ratio of the
# whole line to its active part (ref line.py:42-46)
_PRESETS = (
    ('BAIRD_405', 25.0, 405, (16, 203), (218, 405), 1.2),
    ('NTSC_525', 30000.0 / 1001.0, 525, (21, 263), (283, 525), 858.0 / 720.0),
    ('GERBER_625', 25.0, 625, (336, 623), (23, 310), 1.2),
    ('FRENCH_819', 25.0, 819, (39, 407), (448, 816), 1.2),
    ('BELGIAN_819', 25.0, 819, (437, 816), (27, 406), 1.2),
)


class LineStandard(collections.namedtuple('LineStandard', _FIELDS)):
    """Timing of one analog scanning standard (ref line.py:6-39)."""
    __slots__ = ()

    def __new__(cls, *args, **kwargs):
        std = super(LineStandard, cls).__new__(cls, *args, **kwargs)
        per_field = (std.odd_field_last_active_line - std.odd_field_first_active_line,
                     std.even_field_last_active_line - std.even_field_first_active_line)
        if min(per_field) < 0 or per_field[0] != per_field[1]:
            raise AssertionError('fields must hold the same, non-negative number of lines')
        if std.active_lines > std.total_lines:
            raise AssertionError('more active lines than total lines')
        return std

    @property
    def active_lines(self):
        # both fields hold the same number of lines (checked at construction)
        return 2 * (self.odd_field_last_active_line - self.odd_field_first_active_line + 1)

    @classmethod
    def presets(cls):
        return [getattr(cls, name) for name, *_ in _PRESETS]

    @classmethod
    def detect(cls, active_lines):
        """Smallest preset that still holds `active_lines` lines (ref line.py:28-39)."""
        fitting = [std for std in cls.presets() if std.active_lines >= active_lines]
        if not fitting:
            raise IndexError('No supported line standard supports %d lines' % (active_lines,))
        smallest = min(std.active_lines for std in fitting)
        # among equally sized standards the reference ends up with the one defined last
        return [std for std in fitting if std.active_lines == smallest][-1]


for _name, _rate, _total, _odd, _even, _factor in _PRESETS:
    setattr(LineStandard, _name, LineStandard(_rate, _total, _odd[0], _odd[1], _even[0], _even[1], _factor))


class LineConfig(object):
    """Image size bound to a line standard (ref line.py:49-65)."""

    def __init__(self, size, line_standard=None):
        width, height = int(size[0]), int(size[1])
        std = LineStandard.detect(height) if line_standard is None else line_standard
        self.size = (width, height)
        self.line_standard = std
        # sampling rate: `width` samples in the active part of every line
        self.fs = std.frame_rate * std.total_lines * size[0] * std.total_width_factor
        # image rows are centred in the active lines; rows alternate between the fields, even rows first
        self._line_shift = (std.active_lines - height) // 2
        self._field_start = (std.even_field_first_active_line, std.odd_field_first_active_line)

    def analog_line(self, digital_line):
        row = digital_line + self._line_shift
        return self._field_start[row & 1] + (row >> 1)

    def is_alternate_line(self, frame, line):
        return (self.analog_line(line) ^ frame) & 1 == 0
