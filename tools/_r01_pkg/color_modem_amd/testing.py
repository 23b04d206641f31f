# -*- coding: utf-8 -*-
"""Deterministic synthetic inputs shared by the golden generator, the tests and bench.py.

Everything here is integer arithmetic followed by exact power-of-two scaling, so the
same arrays come out on every machine and numpy version (no dependence on a
floating-point RNG stream or on libm).
"""

import numpy

_MASK = numpy.uint64(0xFFFFFFFFFFFFFFFF)


def _splitmix64(z):
    # z: uint64 array; wrap-around arithmetic is what we want here
    with numpy.errstate(over='ignore'):
        z = (z + numpy.uint64(0x9E3779B97F4A7C15)) & _MASK
        z = ((z ^ (z >> numpy.uint64(30))) * numpy.uint64(0xBF58476D1CE4E5B9)) & _MASK
        z = ((z ^ (z >> numpy.uint64(27))) * numpy.uint64(0x94D049BB133111EB)) & _MASK
        return z ^ (z >> numpy.uint64(31))


def hash_uniform(shape, seed):
    """float64 array of `shape`, values k / 2**24 in [0, 1), k from a 64-bit integer hash."""
    n = int(numpy.prod(shape))
    idx = numpy.arange(n, dtype=numpy.uint64)
    with numpy.errstate(over='ignore'):
        z = _splitmix64(idx ^ _splitmix64(numpy.array([seed], dtype=numpy.uint64)))
    k = (z >> numpy.uint64(40)).astype(numpy.float64)
    return (k / float(1 << 24)).reshape(shape)


def smooth_uniform(shape, seed, taps=4):
    """hash_uniform box-smoothed along the last axis (`taps` wide); still exact in float64."""
    shape = tuple(shape)
    wide = hash_uniform(shape[:-1] + (shape[-1] + taps - 1,), seed)
    acc = numpy.zeros(shape, dtype=numpy.float64)
    for t in range(taps):
        acc += wide[..., t:t + shape[-1]]
    return acc / float(taps)


def synthetic_rgb(n_frames, height, width, seed=1234):
    """rgb[F, 3, H, W] float32 in [0, 1): frame f uses seed + f."""
    out = numpy.empty((n_frames, 3, height, width), dtype=numpy.float32)
    for f in range(n_frames):
        out[f] = smooth_uniform((3, height, width), seed + f).astype(numpy.float32)
    return out


def synthetic_composite(n_frames, height, width, seed=1234):
    """composite[F, H, W] float32 in [-0.05, 1.03): smoothed noise, not a valid colour signal.

    The demodulators have no data-dependent control flow, so this exercises the same
    arithmetic as a real signal; parity on it is the stricter test for the linear paths.
    """
    out = numpy.empty((n_frames, height, width), dtype=numpy.float32)
    for f in range(n_frames):
        out[f] = (1.08 * smooth_uniform((height, width), seed + f) - 0.05).astype(numpy.float32)
    return out
