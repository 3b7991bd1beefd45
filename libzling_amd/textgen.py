"""The benchmark's workload generator (libzling_amd/host/textgen.c -> host/libzlng_textgen.so), bound for bench.py and the
scripts: synthetic enwik-shaped text and the de Bruijn block.  Host-side only, no GPU and no checker involved: the product's
bench makes its own input (the CPU checker under oracle/ is needed only for the parity / cpu_baseline legs)."""
import ctypes as C
import os

import numpy as np

_SO = os.path.join(os.path.dirname(os.path.abspath(__file__)), "host", "libzlng_textgen.so")
_u8p = C.POINTER(C.c_uint8)
_lib = None


def _tg():
    global _lib
    if _lib is None:
        if not os.path.exists(_SO):
            raise RuntimeError("libzlng_textgen.so is missing: run `python __graft_entry__.py` (libzling_amd.build.build_textgen)")
        _lib = C.CDLL(_SO)
        _lib.zt_generate.argtypes = [_u8p, C.c_size_t, C.c_uint64]
        _lib.zt_debruijn3.argtypes = [_u8p]
        _lib.zt_debruijn3.restype = C.c_size_t
    return _lib


def textgen(n, first_chunk=0):
    """`n` bytes of synthetic enwik-shaped text starting at generator chunk `first_chunk` (chunks are independent: rank r of a
    weak-scaling run takes the chunks behind rank r - 1's)."""
    out = np.empty(n, dtype=np.uint8)
    if n:
        _tg().zt_generate(out.ctypes.data_as(_u8p), n, first_chunk)
    return out


def debruijn3():
    """de Bruijn sequence B(256, 3): 16,777,216 bytes, every 3-byte window exactly once (one token per byte: the largest token pool)."""
    out = np.empty(1 << 24, dtype=np.uint8)
    n = _tg().zt_debruijn3(out.ctypes.data_as(_u8p))
    assert n == 1 << 24
    return out
