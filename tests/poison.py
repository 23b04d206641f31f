"""Register / LDS poison in front of a launch (tests/sim/cm_poison.hip -> libcm_poison.so, built by __graft_entry__.build()): TEST TOOL."""
import ctypes
import os

_LIB = None


def lib():
    global _LIB
    if _LIB is None:
        path = os.path.join(os.path.dirname(os.path.abspath(__file__)), 'sim', 'libcm_poison.so')
        L = ctypes.CDLL(path)
        L.cm_poison_device.argtypes = [ctypes.c_void_p, ctypes.c_uint32]
        L.cm_poison_probe.argtypes = [ctypes.c_void_p, ctypes.c_void_p, ctypes.c_int]
        _LIB = L
    return _LIB


def poison(pattern=0x7fc0babe):
    """Leave `pattern` (a quiet NaN) in every VGPR / AGPR and every LDS byte of the current device, on torch's current stream."""
    import torch
    rc = lib().cm_poison_device(ctypes.c_void_p(torch.cuda.current_stream().cuda_stream), pattern)
    assert rc == 0, rc


def probe(n_blocks=1024):
    """What a kernel that writes neither finds in a vector register, an accumulator register and an LDS word: uint32 [n_blocks, 3, 64]."""
    import torch
    out = torch.zeros((n_blocks, 3, 64), dtype=torch.int32, device='cuda')
    rc = lib().cm_poison_probe(ctypes.c_void_p(torch.cuda.current_stream().cuda_stream), ctypes.c_void_p(out.data_ptr()), n_blocks)
    assert rc == 0, rc
    torch.cuda.synchronize()
    return out.cpu().numpy().view('uint32')
