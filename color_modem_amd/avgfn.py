# -*- coding: utf-8 -*-
"""``avg=`` functions (ref comb.py:9-15, 72, 81-84; pal.py:144-148) applied to planes that live on the device.

``comb.avg`` and ``comb.minavg`` have torch forms of their own here (the same arithmetic in float32).  A function of the caller's is first
tried on the float32 device tensors - a whole batch at a time where the reference hands it one float64 numpy row per call; elementwise
functions of two arrays, the only kind that makes sense there, behave the same.  A function written against numpy (``numpy.where``,
``numpy.signbit`` ... - what the reference's own functions use) cannot take device tensors and raises TypeError on them: it is called again
with float64 numpy arrays through host memory.  Only that TypeError is taken as "cannot take tensors": anything else the function or the
device raises (a HIP out-of-memory RuntimeError, an assertion of the function itself) propagates.
"""

import numpy


def apply(fn, last, curr):
    """fn(last, curr) for two float32 tensors of one shape on one device -> float32 tensor of that shape on that device"""
    import torch
    from color_modem_amd import comb
    if fn is comb.avg:
        return 0.5 * (last + curr)
    if fn is comb.minavg:       # comb.py:13-15: sign * min(|a|, |b|), sign = (1 - signbit(a)) - signbit(b)
        sign = (1.0 - torch.signbit(last).to(last.dtype)) - torch.signbit(curr).to(last.dtype)
        return sign * torch.minimum(last.abs(), curr.abs())
    try:
        res = fn(last, curr)
    except TypeError:
        try:
            res = fn(last.detach().cpu().double().numpy(), curr.detach().cpu().double().numpy())
        except TypeError as e:
            raise TypeError('avg=%r must be an elementwise function of two arrays (it is tried on float32 torch tensors on the device, a whole '
                            'batch at a time, then on float64 numpy arrays): %s' % (fn, e))
    if not torch.is_tensor(res):
        res = torch.as_tensor(numpy.asarray(res), dtype=torch.float32, device=last.device)
    if tuple(res.shape) != tuple(curr.shape):
        raise ValueError('avg=%r returned shape %s for inputs of shape %s' % (fn, tuple(res.shape), tuple(curr.shape)))
    return res.to(device=last.device, dtype=torch.float32)
