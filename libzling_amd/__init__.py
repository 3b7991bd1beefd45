"""libzling_amd -- MI355X-native ROLZ+Huffman block codec, drop-in for libzling's Encode/Decode path.

This Python module is plumbing for tests and benchmarks: a ctypes view of the C-ABI in
include/zlng.h (libzlng_hip.so, hand-written gfx950 kernels).  The product is the native
library; there is no Python or CPU implementation of the codec behind these calls, and
loading fails loudly when the HIP library has not been built.
"""
import ctypes as C
import os

import numpy as np

PKG = os.path.dirname(os.path.abspath(__file__))
HIP_SO = os.path.join(PKG, "libzlng_hip.so")
BLOCK = 16777216
MTF_STATE = 65536

_u8p = C.POINTER(C.c_uint8)
_szp = C.POINTER(C.c_size_t)
_lib = None


class ZlngError(RuntimeError):
    def __init__(self, code, what=""):
        self.code = code
        super().__init__("zlng error %d (%s) %s" % (code, strerror(code), what))


def _preload_torch_hip_runtime():
    """A process must run ONE HIP runtime.  PyTorch wheels bundle their own libamdhip64.so.7 (same
    soname as /opt/rocm's, which libzlng_hip.so links): whichever is loaded first serves both.
    torch cannot initialise on the system copy, so when torch is installed make its copy the
    process-wide one before libzlng_hip.so is opened (a no-op if torch is already imported)."""
    import importlib.util
    import sys
    if "torch" in sys.modules or os.environ.get("ZLNG_SYSTEM_HIP") == "1":     # ZLNG_SYSTEM_HIP=1: torch-free tools on /opt/rocm's runtime
        return
    spec = importlib.util.find_spec("torch")
    if spec is None or not spec.origin:
        return
    tl = os.path.join(os.path.dirname(spec.origin), "lib")
    for name in ("libhsa-runtime64.so", "libamdhip64.so"):
        path = os.path.join(tl, name)
        if os.path.exists(path):
            try:
                C.CDLL(path, mode=C.RTLD_GLOBAL)
            except OSError:
                return


def lib():
    """Load libzlng_hip.so (built in-tree by libzling_amd.build / __graft_entry__.build)."""
    global _lib
    if _lib is None:
        if not os.path.exists(HIP_SO):
            raise ImportError("libzlng_hip.so is not built: run `python -m libzling_amd.build` "
                              "(there is no fallback implementation)")
        _preload_torch_hip_runtime()
        # ZLNG_HIP_SO: another build of the same sources (scripts/sanitize.sh device: kernels under -fsanitize=address)
        L = C.CDLL(os.environ.get("ZLNG_HIP_SO") or HIP_SO)
        L.zlng_device_count.restype = C.c_int
        L.zlng_create.restype = C.c_void_p
        L.zlng_create.argtypes = [C.c_int, C.c_int, C.c_int, C.c_int, C.POINTER(C.c_int)]
        L.zlng_destroy.argtypes = [C.c_void_p]
        L.zlng_encode_bound.restype = C.c_size_t
        L.zlng_encode_bound.argtypes = [C.c_size_t]
        L.zlng_encode_blocks.argtypes = [C.c_void_p, _u8p, C.c_size_t, _u8p, C.c_size_t, _szp, _szp]
        L.zlng_encode_blocks_device.argtypes = [C.c_void_p, C.c_void_p, C.c_size_t, C.c_void_p, C.c_size_t, _szp, _szp]
        L.zlng_encode_parse_device.argtypes = [C.c_void_p, C.c_void_p, C.c_size_t]
        L.zlng_encode_finish_device.argtypes = [C.c_void_p, C.c_void_p, C.c_size_t, _szp, _szp]
        L.zlng_encode_parse_after.argtypes = [C.c_void_p, C.c_void_p]
        L.zlng_get_state.argtypes = [C.c_void_p, _u8p, C.POINTER(C.c_int)]
        L.zlng_set_state.argtypes = [C.c_void_p, _u8p, C.c_int]
        L.zlng_get_state_device.argtypes = [C.c_void_p, C.c_void_p, C.POINTER(C.c_int)]
        L.zlng_set_state_device.argtypes = [C.c_void_p, C.c_void_p, C.c_int]
        L.zlng_decode_blocks.argtypes = [C.c_void_p, _u8p, C.c_size_t, _szp, _u8p, C.c_size_t, _szp, _szp]
        L.zlng_decode_blocks_device.argtypes = [C.c_void_p, C.c_void_p, C.c_size_t, _szp, C.c_void_p, C.c_size_t, _szp, _szp]
        L.zlng_last_timings.argtypes = [C.c_void_p, C.POINTER(C.c_char_p), C.POINTER(C.c_float), C.c_int]
        L.zlng_stream.restype = C.c_void_p
        L.zlng_stream.argtypes = [C.c_void_p]
        L.zlng_strerror.restype = C.c_char_p
        L.zlng_strerror.argtypes = [C.c_int]
        L.zlng_group_create.restype = C.c_void_p
        L.zlng_group_create.argtypes = [C.POINTER(C.c_int), C.c_int, C.c_int, C.c_int, C.POINTER(C.c_int)]
        L.zlng_group_destroy.argtypes = [C.c_void_p]
        L.zlng_group_members.argtypes = [C.c_void_p]
        L.zlng_group_capacity.restype = C.c_size_t
        L.zlng_group_capacity.argtypes = [C.c_void_p]
        L.zlng_group_encode_blocks.argtypes = [C.c_void_p, _u8p, C.c_size_t, _u8p, C.c_size_t, _szp, _szp]
        L.zlng_group_encode_parse.argtypes = [C.c_void_p, _u8p, C.c_size_t]
        L.zlng_group_encode_finish.argtypes = [C.c_void_p, _u8p, C.c_size_t, _szp, _szp]
        L.zlng_group_get_state.argtypes = [C.c_void_p, _u8p, C.POINTER(C.c_int)]
        L.zlng_group_set_state.argtypes = [C.c_void_p, _u8p, C.c_int]
        L.zlng_set_host_rank_contexts.argtypes = [C.c_void_p, C.c_int]
        L.zlng_group_set_host_rank_contexts.argtypes = [C.c_void_p, C.c_int]
        _lib = L
    return _lib


def strerror(code):
    return lib().zlng_strerror(code).decode()


def encode_bound(n):
    return lib().zlng_encode_bound(n)


def _ptr(a):
    return a.ctypes.data_as(_u8p)


class Stream:
    """One zlng_ctx: an encode (or decode) stream on one GPU."""

    def __init__(self, device=0, level=0, encode=True, max_blocks=8):
        err = C.c_int(0)
        self._h = lib().zlng_create(device, level, 1 if encode else 0, max_blocks, C.byref(err))
        if not self._h:
            raise ZlngError(err.value, "zlng_create")
        self.max_blocks = max_blocks
        self.level = level

    def close(self):
        h, self._h = getattr(self, "_h", None), None
        if h and _lib is not None:
            _lib.zlng_destroy(h)

    __del__ = close

    def __enter__(self):
        return self

    def __exit__(self, *a):
        self.close()

    # ---- host buffers -------------------------------------------------------------
    def encode(self, data):
        """Encode a whole number of blocks (or the stream tail) held in host memory."""
        a = np.ascontiguousarray(np.frombuffer(bytes(data), np.uint8) if not isinstance(data, np.ndarray) else data)
        cap = encode_bound(a.size)
        out = np.empty(cap, np.uint8)
        n = C.c_size_t(0)
        nb = (a.size + BLOCK - 1) // BLOCK
        ends = (C.c_size_t * max(nb, 1))()
        rc = lib().zlng_encode_blocks(self._h, _ptr(a) if a.size else None, a.size, _ptr(out), cap, C.byref(n), ends)
        if rc != 0:
            raise ZlngError(rc, "zlng_encode_blocks")
        self.block_ends = list(ends)[:nb]
        return out[: n.value].copy()

    def encode_into(self, a, out):
        """Host entry point with caller-owned buffers (no allocation inside the call): bytes written."""
        n = C.c_size_t(0)
        nb = (a.size + BLOCK - 1) // BLOCK
        ends = (C.c_size_t * max(nb, 1))()
        rc = lib().zlng_encode_blocks(self._h, _ptr(a) if a.size else None, a.size, _ptr(out), out.size, C.byref(n), ends)
        if rc != 0:
            raise ZlngError(rc, "zlng_encode_blocks")
        self.block_ends = list(ends)[:nb]
        return n.value

    def decode(self, z, cap):
        a = np.ascontiguousarray(np.frombuffer(bytes(z), np.uint8) if not isinstance(z, np.ndarray) else z)
        out = np.empty(max(cap, 1), np.uint8)
        n = C.c_size_t(0)
        used = C.c_size_t(0)
        rc = lib().zlng_decode_blocks(self._h, _ptr(a) if a.size else None, a.size, C.byref(used), _ptr(out), cap,
                                      C.byref(n), None)
        if rc != 0:
            raise ZlngError(rc, "zlng_decode_blocks")
        return out[: n.value].copy()

    # ---- device buffers (raw pointers, e.g. torch tensor .data_ptr()) ---------------
    def encode_device(self, d_in, in_len, d_out, out_cap):
        n = C.c_size_t(0)
        nb = (in_len + BLOCK - 1) // BLOCK
        ends = (C.c_size_t * max(nb, 1))()
        rc = lib().zlng_encode_blocks_device(self._h, d_in, in_len, d_out, out_cap, C.byref(n), ends)
        if rc != 0:
            raise ZlngError(rc, "zlng_encode_blocks_device")
        self.block_ends = list(ends)[:nb]
        return n.value

    def decode_device(self, d_in, in_len, d_out, out_cap):
        """(compressed bytes consumed, bytes produced); self.block_ends = end offset of every decoded block."""
        n = C.c_size_t(0)
        used = C.c_size_t(0)
        ends = (C.c_size_t * max(self.max_blocks, 1))()
        rc = lib().zlng_decode_blocks_device(self._h, d_in, in_len, C.byref(used), d_out, out_cap, C.byref(n), ends)
        if rc != 0:
            raise ZlngError(rc, "zlng_decode_blocks_device")
        self.block_ends = list(ends)
        return used.value, n.value

    def parse_device(self, d_in, in_len):
        rc = lib().zlng_encode_parse_device(self._h, d_in, in_len)
        if rc != 0:
            raise ZlngError(rc, "zlng_encode_parse_device")

    def parse_after(self, first):
        """Whatever is queued on this context from now on starts after the parse last queued on `first` (same device) has finished."""
        rc = lib().zlng_encode_parse_after(self._h, first._h)
        if rc != 0:
            raise ZlngError(rc, "zlng_encode_parse_after")

    def finish_device(self, d_out, out_cap):
        n = C.c_size_t(0)
        rc = lib().zlng_encode_finish_device(self._h, d_out, out_cap, C.byref(n), None)
        if rc != 0:
            raise ZlngError(rc, "zlng_encode_finish_device")
        return n.value

    def get_state(self):
        buf = np.empty(MTF_STATE, np.uint8)
        lv = C.c_int(0)
        rc = lib().zlng_get_state(self._h, _ptr(buf), C.byref(lv))
        if rc != 0:
            raise ZlngError(rc, "zlng_get_state")
        return buf, lv.value

    def set_state(self, mtf, level):
        mtf = np.ascontiguousarray(mtf, np.uint8)
        rc = lib().zlng_set_state(self._h, _ptr(mtf), level)
        if rc != 0:
            raise ZlngError(rc, "zlng_set_state")

    def get_state_device(self, d_ptr):
        lv = C.c_int(0)
        rc = lib().zlng_get_state_device(self._h, d_ptr, C.byref(lv))
        if rc != 0:
            raise ZlngError(rc, "zlng_get_state_device")
        return lv.value

    def set_state_device(self, d_ptr, level):
        rc = lib().zlng_set_state_device(self._h, d_ptr, level)
        if rc != 0:
            raise ZlngError(rc, "zlng_set_state_device")

    def set_host_rank_contexts(self, k):
        """Measured alternative (zlng.h): the k longest rank chains of the following calls on host threads; 0 = all-device."""
        rc = lib().zlng_set_host_rank_contexts(self._h, k)
        if rc != 0:
            raise ZlngError(rc, "zlng_set_host_rank_contexts")

    def debug_fetch(self, what, blk, dtype, count):
        """Test hook (zlng_debug_fetch): internal buffer `what` of block `blk` after the last encode."""
        out = np.empty(count, dtype=dtype)
        f = lib().zlng_debug_fetch
        f.argtypes = [C.c_void_p, C.c_int, C.c_int, C.c_void_p, C.c_size_t]
        rc = f(self._h, what, blk, out.ctypes.data, out.nbytes)
        if rc != 0:
            raise ZlngError(rc, "zlng_debug_fetch")
        return out

    def debug_lengths(self, freq):
        """Test hook (zlng_debug_lengths): K4 on caller-supplied rows of 546 counts -> (lens u8, codes u16)."""
        freq = np.ascontiguousarray(freq, np.uint32)
        assert freq.ndim == 2 and freq.shape[1] == 546
        lens = np.empty(freq.shape, np.uint8)
        codes = np.empty(freq.shape, np.uint16)
        f = lib().zlng_debug_lengths
        f.argtypes = [C.c_void_p, C.c_void_p, C.c_int, C.c_void_p, C.c_void_p]
        rc = f(self._h, freq.ctypes.data, freq.shape[0], lens.ctypes.data, codes.ctypes.data)
        if rc != 0:
            raise ZlngError(rc, "zlng_debug_lengths")
        return lens, codes

    def passes(self):
        """Test hook: parse passes the last encode needed (1 = no level-schedule repair, no pool growth)."""
        f = lib().zlng_debug_passes
        f.argtypes = [C.c_void_p]
        return f(self._h)

    def block_tokens(self, blk):
        ntok, nsub = self.debug_fetch(5, blk, np.uint32, 2)
        tok = self.debug_fetch(0, blk, np.uint32, int(ntok))
        cuts = self.debug_fetch(1, blk, np.uint32, 4 * int(nsub)).reshape(-1, 4)
        return tok, cuts

    def timings(self):
        names = (C.c_char_p * 16)()
        ms = (C.c_float * 16)()
        n = lib().zlng_last_timings(self._h, names, ms, 16)
        return [(names[i].decode(), float(ms[i])) for i in range(n)]

    def hip_stream(self):
        return lib().zlng_stream(self._h)


class Group:
    """One zlng_group: ONE stream over several contexts / devices (contiguous block ranges, state handed member to member)."""

    def __init__(self, devices, level=0, blocks_per_member=8):
        err = C.c_int(0)
        arr = (C.c_int * len(devices))(*devices)
        self._h = lib().zlng_group_create(arr, len(devices), level, blocks_per_member, C.byref(err))
        if not self._h:
            raise ZlngError(err.value, "zlng_group_create")
        self.level = level

    def close(self):
        h, self._h = getattr(self, "_h", None), None
        if h and _lib is not None:
            _lib.zlng_group_destroy(h)

    __del__ = close

    def __enter__(self):
        return self

    def __exit__(self, *a):
        self.close()

    def encode(self, data, split=False):
        a = np.ascontiguousarray(np.frombuffer(bytes(data), np.uint8) if not isinstance(data, np.ndarray) else data)
        cap = encode_bound(a.size)
        out = np.empty(cap, np.uint8)
        n = C.c_size_t(0)
        nb = (a.size + BLOCK - 1) // BLOCK
        ends = (C.c_size_t * max(nb, 1))()
        if split:
            rc = lib().zlng_group_encode_parse(self._h, _ptr(a), a.size)
            if rc == 0:
                rc = lib().zlng_group_encode_finish(self._h, _ptr(out), cap, C.byref(n), ends)
        else:
            rc = lib().zlng_group_encode_blocks(self._h, _ptr(a) if a.size else None, a.size, _ptr(out), cap, C.byref(n), ends)
        if rc != 0:
            raise ZlngError(rc, "zlng_group_encode")
        self.block_ends = list(ends)[:nb]
        return out[: n.value].copy()

    def set_host_rank_contexts(self, k):
        rc = lib().zlng_group_set_host_rank_contexts(self._h, k)
        if rc != 0:
            raise ZlngError(rc, "zlng_group_set_host_rank_contexts")

    def get_state(self):
        buf = np.empty(MTF_STATE, np.uint8)
        lv = C.c_int(0)
        rc = lib().zlng_group_get_state(self._h, _ptr(buf), C.byref(lv))
        if rc != 0:
            raise ZlngError(rc, "zlng_group_get_state")
        return buf, lv.value

    def set_state(self, mtf, level):
        mtf = np.ascontiguousarray(mtf, np.uint8)
        rc = lib().zlng_group_set_state(self._h, _ptr(mtf), level)
        if rc != 0:
            raise ZlngError(rc, "zlng_group_set_state")


def encode(data, level=0, device=0):
    """Whole-stream convenience: encode `data` (any size) as one .zlng stream on one GPU."""
    a = np.ascontiguousarray(np.frombuffer(bytes(data), np.uint8) if not isinstance(data, np.ndarray) else data)
    nb = max(1, (a.size + BLOCK - 1) // BLOCK)
    with Stream(device, level, True, nb) as s:
        return s.encode(a)
