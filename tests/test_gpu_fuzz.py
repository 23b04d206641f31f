# -*- coding: utf-8 -*-
"""A bounded, fixed-seed slice of the randomised parity sweep (tests/fuzz_parity.py) in the driver-run GPU suite:
random (stack, variant, width 480 .. 1920, height, frame count, first frame) of every family - PAL, NTSC, SECAM, the comb wrappers incl.
avg= callables, Proto-SECAM / NIIR, D2-MAC, the nested stacks - both directions and the fused byte boundaries, against the float64
oracles, hard cap 1e-5 on every case.  The builder's long campaigns (thousands of cases, profiles/r0*_fuzz_summary.txt) use the same
function with other seeds."""
import pytest

pytestmark = pytest.mark.gpu


@pytest.mark.parametrize('only,cases,seed', [('pal', 40, 601), ('ntsc', 36, 602), ('secam', 36, 603), ('am', 20, 604), ('nested', 20, 605), (None, 40, 606)])
def test_fixed_seed_fuzz_slice(only, cases, seed):
    import fuzz_parity
    lines = []
    done, worst, bad = fuzz_parity.run(cases, seed, only, out=lines.append, max_h=96, full_share=0.08)
    assert done == cases
    assert not bad and worst < 1e-5, '\n'.join([ln for ln in lines if 'FAIL' in ln] or lines[-3:])
