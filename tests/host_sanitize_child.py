# -*- coding: utf-8 -*-
"""Child process of tests/test_host_sanitize.py: python tests/host_sanitize_child.py LIBRARY - builds the plans of every family against the
library given (the -DCM_HOST_DRY_RUN sanitizer build: no device needed), describes and destroys them, feeds malformed descriptors."""
import sys, os
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
LIB = os.path.abspath(sys.argv[1])

import ctypes, os, sys, warnings
warnings.filterwarnings('ignore')
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, 'tests'))
os.environ['CM_LIB'] = LIB
import numpy
from color_modem_amd import _native, comb, engine, line, plan
assert _native.LIB_PATH == LIB
L = _native.lib()

# plans without torch / a device: one handle per engine, created as engine._DevicePlans would under the current device
def dry_get(self, device=None):
    handle = self._plans.get(0)
    if handle is None:
        handle = ctypes.c_void_p()
        _native.check(self._create(ctypes.byref(handle)))
        self._plans[0] = handle
        if self.on_create is not None:
            self.on_create(handle)
    return handle
engine._DevicePlans.get = dry_get
engine._EngineBase._plan = property(lambda self: self._plans.get(None))

from color_modem_amd.color import mac, niir, ntsc, pal, protosecam, secam
import stacks
built = refused = 0
def build(make, *args, **kw):
    global built, refused
    try:
        eng = make(*args, **kw)
    except (NotImplementedError, ValueError, AttributeError, IndexError) as e:      # a refusal with a message is a result too
        refused += 1
        return None
    text = eng.describe()
    assert text
    for mode in ('rows', 'auto'):
        try:
            eng.set_small_batch(mode)
        except NotImplementedError:
            pass
    built += 1
    return eng

def variants(cls):
    return [v for k, v in sorted(vars(cls).items()) if isinstance(v, cls)]

WIDTHS = (480, 702, 720, 768, 1024, 1280, 1920)      # below / at / above 13.5 MHz: tuned, wide (cm_shapes_wide.h) and run-time shapes, one not a multiple of 4
QAM = {
    'pal': (pal.PalVariant, 576, [lambda lc, v: pal.PalSModem(lc, v), lambda lc, v: pal.PalDModem(lc, v), lambda lc, v: pal.Pal3DModem(lc, v),
                                  lambda lc, v: pal.PalDModem(lc, v, notch=4.0), lambda lc, v: pal.Pal3DModem(lc, v, avg=comb.minavg),
                                  lambda lc, v: comb.SimpleCombModem(pal.PalSModem(lc, v)), lambda lc, v: comb.ColorAveragingModem(pal.PalSModem(lc, v)),
                                  lambda lc, v: comb.Simple3DCombModem(pal.PalDModem(lc, v)), lambda lc, v: comb.SimpleCombModem(pal.Pal3DModem(lc, v), notch=3.0, avg=comb.minavg),
                                  lambda lc, v: comb.Simple3DCombModem(pal.PalDModem(lc, v), avg=stacks.weighted_avg), lambda lc, v: pal.Pal3DModem(lc, v, avg=stacks.damped_avg),
                                  lambda lc, v: pal.PalDModem(lc, v, notch=1.0)]),
    'ntsc': (ntsc.NtscVariant, 480, [lambda lc, v: ntsc.NtscModem(lc, v), lambda lc, v: ntsc.NtscCombModem(lc, v),
                                     lambda lc, v: comb.Simple3DCombModem(ntsc.NtscCombModem(lc, v)), lambda lc, v: comb.SimpleCombModem(ntsc.NtscModem(lc, v), avg=comb.minavg),
                                     lambda lc, v: comb.ColorAveragingModem(ntsc.NtscModem(lc, v)), lambda lc, v: comb.Simple3DCombModem(ntsc.NtscCombModem(lc, v, notch=2.5), notch=8.0)]),
    'secam': (secam.SecamVariant, 576, [lambda lc, v: secam.SecamModem(lc, v), lambda lc, v: comb.ColorAveragingModem(secam.SecamModem(lc, v))]),
}
for system, (cls, full, makers) in QAM.items():
    for iv, v in enumerate(variants(cls)):
        for w in WIDTHS:
            # every stack on the system's first variant, the three basic ones on the others; a few rows of the full-height standard (the
            # per-line tables are Python loops - this is a sweep of the NATIVE constructors), the full height once per system
            for h in ((full, 9) if (iv == 0 and w == 720) else (9,)):
                for make in (makers if iv == 0 else makers[:3]):
                    def one(components=False, strip=True):
                        lc = line.LineConfig((w, h), line.LineStandard.detect(full))
                        return engine.make_engine(make(lc, v), components=components, strip_chroma=strip, min_lines=0 if h == full else 24)
                    build(one)
                    if w in (720, 1024) and h == 9 and iv == 0:
                        build(one, True, False)
                        build(one, True, True)
# Proto-SECAM / NIIR
for std in ('FRENCH_819', 'BELGIAN_819', 'GERBER_625', 'NTSC_525'):
    for w in (400, 720, 1000, 1280):
        lc = line.LineConfig((w, 12), getattr(line.LineStandard, std))
        for make in (lambda: protosecam.ProtoSecamModem(lc), lambda: comb.ColorAveragingModem(protosecam.ProtoSecamModem(lc)),
                     lambda: protosecam.ProtoSecamModem(lc, premod_luma_filter=False), lambda: niir.NiirModem(lc), lambda: niir.HueCorrectingNiirModem(lc),
                     lambda: niir.NiirModem(lc, noise_level=0.05)):
            for comps in (False, True):
                build(lambda: engine.make_engine(make(), components=comps, strip_chroma=not comps))
# D2-MAC: the tuned shape, resampled rows and lines
for w, cw in ((720, 1080), (720, 720), (768, 1080), (640, 900), (1920, 4096), (300, 401)):
    lc = line.LineConfig((w, 11), line.LineStandard.GERBER_625)
    for make in (lambda: mac.MacModem(lc, cw), lambda: comb.ColorAveragingModem(mac.MacModem(lc, cw))):
        build(lambda: engine.make_engine(make()))
# the nested stacks (generic.py) hold engines of every kind
for name in stacks.NESTED:
    for size in ((720, 10), (702, 9)):
        build(lambda: engine.make_engine(stacks.make_nested(name, size)))
print('plans built: %d engines, %d refusals with a message' % (built, refused))
assert built > 400

# ---- malformed descriptors: an error code and a message, never a crash or a wild read -----------------------------------------
def fresh():
    return plan.build_plan(stacks.make('pal_d', (720, 576)))
handle = ctypes.c_void_p()
checked = 0
def expect_error(mutate, what):
    global checked
    bp = fresh()
    keep = mutate(bp.desc)
    rc = L.cm_plan_create(ctypes.byref(bp.desc), ctypes.byref(handle))
    assert rc != 0 and L.cm_last_error(), what
    assert not handle.value
    checked += 1
def setter(path, value):
    def f(d):
        obj = d
        for name in path[:-1]:
            obj = getattr(obj, name)
        setattr(obj, path[-1], value)
    return f
for path, value in ((('abi_version',), 99), (('width',), 3), (('width',), -720), (('height',), 0), (('pipeline',), 7), (('depth',), 5), (('depth',), -1),
                    (('skip_calls',), 1), (('demod_main', 'wrap_mode'), 3), (('demod_main', 'n_lines'), 0), (('demod_main', 'frame_cycle'), 0),
                    (('demod_first', 'n_lines'), 5), (('extract2x', 'n_sections'), 9), (('extract2x', 'n_sections'), -2),
                    (('precorrect', 'shift'), 4000), (('extract2x', 'shift'), -5000), (('remove2x', 'n_sections'), 40)):
    expect_error(setter(path, value), path)
def null_table(d):
    d.demod_main.table = None
expect_error(null_table, 'null table')
def bad_section(d):
    d.extract2x.sos[0][0] = 0.0
def not_bandpass(d):
    d.extract2x.sos[0][1] = 0.5
def nan_pole(d):
    d.demod_lp.sos[0][4] = float('nan')
def no_front_lowpass(d):
    d.pald_lp.n_sections = 0         # a shorter cascade is legal (the run-time shape pads it): a PAL-D plan without its low-pass builds
# a plan is usable in one direction when only the other one lacks a kernel instance (cm_plan_create): what only the decoder cannot take
# leaves an encode-only plan
for odd in (nan_pole, no_front_lowpass, bad_section, not_bandpass, setter(('extract2x', 'shift'), -3), setter(('notch', 'n_sections'), 3), setter(('demod_lp', 'shift'), 77)):
    bp = fresh(); odd(bp.desc)
    rc = L.cm_plan_create(ctypes.byref(bp.desc), ctypes.byref(handle))      # accepted or refused - but nothing may be read out of bounds on the way
    if rc == 0:
        L.cm_plan_destroy(handle); handle = ctypes.c_void_p()
    checked += 1
assert L.cm_plan_create(None, ctypes.byref(handle)) != 0
# SECAM and the AM / MAC descriptors
from color_modem_amd import plan_am
sd = plan.build_plan(stacks.make('secam', (720, 576)))
for path, value in ((('secam', 'present'), 0), (('secam', 'chroma_bp', 'n_sections'), 7), (('secam', 'bell', 'shift'), 2), (('secam', 'preroll'), -5)):
    d = plan.build_plan(stacks.make('secam', (720, 576)))
    setter(path, value)(d.desc)
    rc = L.cm_plan_create(ctypes.byref(d.desc), ctypes.byref(handle))
    if rc == 0:
        L.cm_plan_destroy(handle); handle = ctypes.c_void_p()
    checked += 1
ad = plan_am.build_am_desc(niir.NiirModem(line.LineConfig((720, 576))))
for name, value in (('abi_version', 1), ('kind', 9), ('width', 2), ('height', 0), ('frame_cycle', 0)):
    d = plan_am.build_am_desc(niir.NiirModem(line.LineConfig((720, 576))))
    setattr(d, name, value)
    rc = L.cm_am_plan_create(ctypes.byref(d), ctypes.byref(handle))
    if rc == 0:
        L.cm_am_plan_destroy(handle); handle = ctypes.c_void_p()
    else:
        assert L.cm_last_error()
    checked += 1
print('malformed descriptors: %d' % checked)
print('sanitized host run complete')
