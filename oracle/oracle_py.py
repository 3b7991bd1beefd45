"""ctypes bindings onto oracle/liboracle.so and (when built) oracle/_ref/libzling_ref.so.

TEST INFRASTRUCTURE ONLY: imported by tests/, __graft_entry__.smoke() and bench.py's
cpu_baseline leg.  Nothing under libzling_amd/ imports this module.
"""
import ctypes as C
import os
import subprocess

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
_u8p = C.POINTER(C.c_uint8)


def build(force=False):
    if os.environ.get("ZLNG_ORACLE_SO"):             # scripts/sanitize.sh: the ASan/UBSan build of the same source
        return os.environ["ZLNG_ORACLE_SO"]
    so = os.path.join(HERE, "liboracle.so")
    src = os.path.join(HERE, "zlng_oracle.c")
    stale = (not os.path.exists(so)) or os.path.getmtime(so) < os.path.getmtime(src)
    if force or stale or (os.path.isdir("/root/reference/src") and not os.path.exists(os.path.join(HERE, "_ref", "libzling_ref.so"))):
        import fcntl
        with open(os.path.join(HERE, ".build.lock"), "w") as lock:      # parallel test workers: one make at a time
            fcntl.flock(lock, fcntl.LOCK_EX)
            subprocess.check_call(["make", "-s", "-C", HERE])
    return so


def _ptr(a):
    return a.ctypes.data_as(_u8p)


class Oracle:
    """The CPU restatement (zlng_oracle.c)."""

    def __init__(self):
        self.lib = C.CDLL(build())
        L = self.lib
        L.zo_encode.argtypes = [_u8p, C.c_size_t, C.c_int, _u8p, C.c_size_t, C.POINTER(C.c_size_t)]
        L.zo_decode.argtypes = [_u8p, C.c_size_t, _u8p, C.c_size_t, C.POINTER(C.c_size_t)]
        L.zo_decode_ex.argtypes = [_u8p, C.c_size_t, _u8p, C.c_size_t, C.POINTER(C.c_size_t), C.POINTER(C.c_uint32)]
        L.zo_encode_bound.argtypes = [C.c_size_t]
        L.zo_encode_bound.restype = C.c_size_t
        L.zo_stream_new.restype = C.c_void_p
        L.zo_stream_new.argtypes = [C.c_int]
        L.zo_stream_free.argtypes = [C.c_void_p]
        L.zo_reset_buckets.argtypes = [C.c_void_p]
        L.zo_stream_get_mtf.argtypes = [C.c_void_p, _u8p]
        L.zo_stream_set_mtf.argtypes = [C.c_void_p, _u8p]
        L.zo_stream_get_level.argtypes = [C.c_void_p]
        L.zo_stream_set_level.argtypes = [C.c_void_p, C.c_int]
        L.zo_encode_blocks.argtypes = [C.c_void_p, _u8p, C.c_size_t, _u8p, C.c_size_t, C.POINTER(C.c_size_t)]
        L.zo_parse_subblock.argtypes = [C.c_void_p, C.c_int, _u8p, C.c_int, C.POINTER(C.c_int), C.c_void_p,
                                        C.POINTER(C.c_int), C.c_int]
        L.zo_mtf_rank.argtypes = [C.c_void_p, C.c_void_p, C.c_size_t]
        L.zo_histogram.argtypes = [C.c_void_p, C.c_size_t, C.c_void_p, C.c_void_p]
        L.zo_make_length_table.argtypes = [C.c_void_p, C.c_void_p, C.c_int, C.c_int]
        L.zo_make_encode_table.argtypes = [C.c_void_p, C.c_void_p, C.c_int, C.c_int]
        L.zo_pack_subblock.argtypes = [C.c_void_p, C.c_size_t, C.c_void_p, C.c_void_p, _u8p]
        L.zo_pack_subblock.restype = C.c_size_t
        for name, n, dt in (("zo_mtfnext", 256, C.c_uint8), ("zo_matchidx_code", 4096, C.c_uint8),
                            ("zo_matchidx_base", 32, C.c_uint16), ("zo_matchidx_blen", 32, C.c_uint8)):
            getattr(L, name).restype = C.POINTER(dt)

    def table(self, name, n):
        p = getattr(self.lib, name)()
        return np.array([p[i] for i in range(n)], dtype=np.int64)

    def mtfinit(self):
        return np.array((C.c_uint8 * 256).in_dll(self.lib, "zo_mtfinit"), dtype=np.int64)

    def encode(self, data, level=0):
        a = np.frombuffer(bytes(data), dtype=np.uint8) if not isinstance(data, np.ndarray) else data
        a = np.ascontiguousarray(a)
        cap = self.lib.zo_encode_bound(a.size)
        out = np.empty(cap, dtype=np.uint8)
        n = C.c_size_t(0)
        rc = self.lib.zo_encode(_ptr(a) if a.size else None, a.size, level, _ptr(out), cap, C.byref(n))
        if rc != 0:
            raise RuntimeError("zo_encode rc=%d" % rc)
        return out[: n.value].copy()

    def decode(self, z, cap):
        a = np.ascontiguousarray(np.frombuffer(bytes(z), dtype=np.uint8) if not isinstance(z, np.ndarray) else z)
        out = np.empty(max(cap, 1), dtype=np.uint8)
        n = C.c_size_t(0)
        rc = self.lib.zo_decode(_ptr(a) if a.size else None, a.size, _ptr(out), cap, C.byref(n))
        return rc, out[: n.value].copy()

    def decode_ex(self, z, cap):
        """(rc, bytes of the complete blocks in front of the first error, flags): flags != 0 means one of the decoder's own rules
        for hostile streams decided or coloured the verdict (ZO_DEV_*, zlng_oracle.h) -- the reference itself would read or write
        memory it does not own there."""
        a = np.ascontiguousarray(np.frombuffer(bytes(z), dtype=np.uint8) if not isinstance(z, np.ndarray) else z)
        out = np.empty(max(cap, 1), dtype=np.uint8)
        n = C.c_size_t(0)
        fl = C.c_uint32(0)
        rc = self.lib.zo_decode_ex(_ptr(a) if a.size else None, a.size, _ptr(out), cap, C.byref(n), C.byref(fl))
        return rc, out[: n.value].copy(), int(fl.value)

    def decode_blockwise(self, z, cap, state=None):
        """The block-at-a-time decoder (zo_dstream_*, what tests/cxx/zlng_stub.c is built on): (rc, bytes of the good blocks in front
        of the first error, blocks decoded, final tables).  `state`: 65,536 table bytes to start from (default: a fresh decoder)."""
        L = self.lib
        L.zo_dstream_new.restype = C.c_void_p
        L.zo_dstream_free.argtypes = [C.c_void_p]
        L.zo_dstream_get_mtf.argtypes = [C.c_void_p, _u8p]
        L.zo_dstream_set_mtf.argtypes = [C.c_void_p, _u8p]
        L.zo_dstream_decode_block.argtypes = [C.c_void_p, _u8p, C.c_size_t, C.POINTER(C.c_size_t), _u8p, C.c_size_t, C.POINTER(C.c_size_t), C.POINTER(C.c_uint32)]
        a = np.ascontiguousarray(z)
        d = L.zo_dstream_new()
        if state is not None:
            L.zo_dstream_set_mtf(d, _ptr(np.ascontiguousarray(state)))
        out = np.empty(max(cap, 1), dtype=np.uint8)
        ip, op, nblk, rc = C.c_size_t(0), 0, 0, 0
        while ip.value < a.size:
            n = C.c_size_t(0)
            rc = L.zo_dstream_decode_block(d, _ptr(a), a.size, C.byref(ip), _ptr(out[op:]) if op < out.size else _ptr(out), cap - op, C.byref(n), None)
            if rc != 0:
                break
            op += n.value
            nblk += 1
        mtf = np.empty(65536, np.uint8)
        L.zo_dstream_get_mtf(d, _ptr(mtf))
        L.zo_dstream_free(d)
        return rc, out[:op].copy(), nblk, mtf

    def decode_stats(self, z, cap, lag=False):
        """(rc, dict): what the decode reached (zo_dstats, zlng_oracle.h); lag=True also fills the replay-split model's fields."""
        class St(C.Structure):
            _fields_ = [("max_inserts_one_context", C.c_uint32), ("max_distance", C.c_uint32), ("matches", C.c_uint64),
                        ("matches_in_wrapped_ring", C.c_uint64), ("beyond_window", C.c_uint64), ("far_matches", C.c_uint64),
                        ("dst_straddles_64k", C.c_uint64), ("src_straddles_64k", C.c_uint64), ("scratch", C.c_uint32 * 256),
                        ("writer", C.c_void_p), ("lag_hist", C.c_uint64 * 24), ("tokens", C.c_uint64), ("literals", C.c_uint64),
                        ("words", C.c_uint64), ("match_bytes", C.c_uint64)]
        a = np.ascontiguousarray(z)
        out = np.empty(max(cap, 1), dtype=np.uint8)
        n = C.c_size_t(0)
        st = St()
        wr = np.zeros((1 << 24) + 1024, np.uint32) if lag else None
        st.writer = wr.ctypes.data if lag else None
        self.lib.zo_decode_stats.argtypes = [_u8p, C.c_size_t, _u8p, C.c_size_t, C.POINTER(C.c_size_t), C.POINTER(St)]
        rc = self.lib.zo_decode_stats(_ptr(a), a.size, _ptr(out), cap, C.byref(n), C.byref(st))
        d = {k: int(getattr(st, k)) for k, _ in St._fields_[:8]}
        if lag:
            d.update(lag_hist=[int(v) for v in st.lag_hist], tokens=int(st.tokens), literals=int(st.literals), words=int(st.words),
                     match_bytes=int(st.match_bytes))
        return rc, d

    # ---- stage API -------------------------------------------------------------------
    def parse_block(self, block, level=0, apply_mtf=False, stream=None):
        """Parse one <=16 MiB block.  Returns (tokens u32, cuts[(tok_end, encpos, rlen)])."""
        L = self.lib
        own = stream is None
        s = L.zo_stream_new(level) if own else stream
        L.zo_reset_buckets(s)
        ib = np.zeros(block.size + 275, dtype=np.uint8)
        ib[: block.size] = block
        toks, cuts = [], []
        encpos = C.c_int(0)
        rlen = C.c_int(0)
        buf = np.empty(262144, dtype=np.uint32)
        total = 0
        while encpos.value < block.size:
            nt = L.zo_parse_subblock(s, level, _ptr(ib), block.size, C.byref(encpos), buf.ctypes.data, C.byref(rlen),
                                     1 if apply_mtf else 0)
            toks.append(buf[:nt].copy())
            total += nt
            cuts.append((total, encpos.value, rlen.value))
        if own:
            L.zo_stream_free(s)
        return (np.concatenate(toks) if toks else np.empty(0, np.uint32)), cuts

    def histogram(self, tok):
        f1 = np.zeros(514, np.uint32)
        f2 = np.zeros(32, np.uint32)
        tok = np.ascontiguousarray(tok, dtype=np.uint32)
        self.lib.zo_histogram(tok.ctypes.data, tok.size, f1.ctypes.data, f2.ctypes.data)
        return f1, f2

    def length_table(self, freq, limit):
        freq = np.ascontiguousarray(freq, dtype=np.uint32)
        out = np.zeros(freq.size, np.uint32)
        self.lib.zo_make_length_table(freq.ctypes.data, out.ctypes.data, freq.size, limit)
        return out

    def encode_table(self, lens, limit):
        lens = np.ascontiguousarray(lens, dtype=np.uint32)
        out = np.zeros(lens.size, np.uint16)
        self.lib.zo_make_encode_table(lens.ctypes.data, out.ctypes.data, lens.size, limit)
        return out

    def pack(self, tok, l1, l2):
        tok = np.ascontiguousarray(tok, dtype=np.uint32)
        out = np.zeros(273 + tok.size * 4 + 16, np.uint8)
        n = self.lib.zo_pack_subblock(tok.ctypes.data, tok.size, np.ascontiguousarray(l1, np.uint32).ctypes.data,
                                      np.ascontiguousarray(l2, np.uint32).ctypes.data, _ptr(out))
        return out[:n].copy()


class Reference:
    """The real reference compiled by oracle/Makefile (absent if oracle/_ref was never built)."""

    @staticmethod
    def available():
        if os.environ.get("ZLNG_NO_REF") == "1":       # scripts/sanitize.sh: the uninstrumented reference stays out of an ASan process
            return False
        build()
        return os.path.exists(os.path.join(HERE, "_ref", "libzling_ref.so"))

    def __init__(self):
        build()
        self.lib = C.CDLL(os.path.join(HERE, "_ref", "libzling_ref.so"))
        L = self.lib
        L.ref_encode.argtypes = [_u8p, C.c_size_t, C.c_int, _u8p, C.c_size_t, C.POINTER(C.c_size_t)]
        L.ref_decode.argtypes = [_u8p, C.c_size_t, _u8p, C.c_size_t, C.POINTER(C.c_size_t), C.c_char_p, C.c_size_t]
        L.ref_make_length_table.argtypes = [C.c_void_p, C.c_void_p, C.c_int, C.c_int]
        L.ref_make_encode_table.argtypes = [C.c_void_p, C.c_void_p, C.c_int, C.c_int]
        L.ref_rolz_new.restype = C.c_void_p
        L.ref_rolz_free.argtypes = [C.c_void_p]
        L.ref_rolz_block.argtypes = [C.c_void_p, C.c_int, _u8p, C.c_int, C.c_void_p, C.c_size_t, C.c_void_p, C.c_int]
        L.ref_mtf_chain.argtypes = [_u8p, C.c_size_t, _u8p]
        L.ref_mtf_chain.restype = C.c_uint64

    def encode(self, data, level=0):
        a = np.ascontiguousarray(np.frombuffer(bytes(data), dtype=np.uint8) if not isinstance(data, np.ndarray) else data)
        cap = a.size * 2 + a.size // 100000 * 400 + 4096
        out = np.empty(cap, dtype=np.uint8)
        n = C.c_size_t(0)
        rc = self.lib.ref_encode(_ptr(a) if a.size else None, a.size, level, _ptr(out), cap, C.byref(n))
        if rc != 0:
            raise RuntimeError("ref_encode rc=%d" % rc)
        return out[: n.value].copy()

    def decode(self, z, cap):
        a = np.ascontiguousarray(np.frombuffer(bytes(z), dtype=np.uint8) if not isinstance(z, np.ndarray) else z)
        out = np.empty(max(cap, 1), dtype=np.uint8)
        n = C.c_size_t(0)
        msg = C.create_string_buffer(256)
        rc = self.lib.ref_decode(_ptr(a) if a.size else None, a.size, _ptr(out), cap, C.byref(n), msg, 256)
        return rc, out[: n.value].copy(), msg.value.decode()

    def length_table(self, freq, limit):
        freq = np.ascontiguousarray(freq, dtype=np.uint32)
        out = np.zeros(freq.size + (freq.size & 1), np.uint32)
        self.lib.ref_make_length_table(freq.ctypes.data, out.ctypes.data, freq.size, limit)
        return out[: freq.size]

    def mtf_chain(self, lits):
        """Ranks of one context's literal bytes from the initial table (the reference's ZlingMTFEncoder)."""
        a = np.ascontiguousarray(lits, dtype=np.uint8)
        out = np.empty(a.size, np.uint8)
        self.lib.ref_mtf_chain(_ptr(a), a.size, _ptr(out))
        return out

    def rolz_block(self, block, level=0):
        """Reference u16 token stream of one block with a FRESH encoder: (tok16, [(encpos, rlen)])."""
        e = self.lib.ref_rolz_new()
        ib = np.zeros(block.size + 275, dtype=np.uint8)
        ib[: block.size] = block
        cap = 2 * block.size + 600000
        t = np.zeros(cap, np.uint16)
        cuts = np.zeros(2 * 200, np.int32)
        ns = self.lib.ref_rolz_block(e, level, _ptr(ib), block.size, t.ctypes.data, cap, cuts.ctypes.data, 200)
        self.lib.ref_rolz_free(e)
        assert ns >= 0
        cl = [(int(cuts[2 * i]), int(cuts[2 * i + 1])) for i in range(ns)]
        return t[: sum(c[1] for c in cl)].copy(), cl


_tg = None


def textgen(n, first_chunk=0):
    """Synthetic enwik-shaped text (libzling_amd/host/textgen.c)."""
    global _tg
    if _tg is None:
        so = os.path.join(HERE, "..", "libzling_amd", "host", "libzlng_textgen.so")
        _tg = C.CDLL(so)
        _tg.zt_generate.argtypes = [_u8p, C.c_size_t, C.c_uint64]
    out = np.empty(n, dtype=np.uint8)
    if n:
        _tg.zt_generate(_ptr(out), n, first_chunk)
    return out


def debruijn3():
    """de Bruijn sequence B(256, 3): 16,777,216 bytes, every 3-byte window once (libzling_amd/host/textgen.c)."""
    textgen(0)
    _tg.zt_debruijn3.argtypes = [_u8p]
    _tg.zt_debruijn3.restype = C.c_size_t
    out = np.empty(1 << 24, dtype=np.uint8)
    n = _tg.zt_debruijn3(_ptr(out))
    assert n == 1 << 24
    return out
