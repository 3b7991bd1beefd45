"""Hostile-stream differential of the HIP decoder (K7 frame walk, K8 Huffman decode, K9 replay) against the oracle, through the
C-ABI: thousands of damaged and crafted .zlng streams (tests/hostile.py), and for each one

  * the same verdict class -- ZLNG_E_* <-> the reference's exception message (src/libzling.cpp:316, 327, 382, 392, 399, 407),
    ZLNG_E_TRUNC for an input that ends inside a block -- reached in the same place: the bytes of the complete blocks in front of
    the error are reported first, exactly those;
  * when both accept: the same bytes;
  * the process survives every case, and the context decodes a good stream afterwards.

The oracle's own verdicts are held against the REAL reference on the CPU side (tests/test_oracle_hostile.py)."""
import ctypes as C

import numpy as np
import pytest

import corpus
import hostile

pytestmark = pytest.mark.gpu
CODE = {0: 0, -2: -10, -3: -11, -4: -12, -5: -13, -6: -14, -7: -15, -8: -16}      # ZO_E_* -> ZLNG_E_*


@pytest.fixture(scope="module")
def zl():
    import libzling_amd as zl
    assert zl.lib().zlng_device_count() >= 1, "no gfx950 device visible"
    return zl


def gpu_verdict(zl, s, init, m, cap):
    """(code, bytes reported before it): zlng_decode_blocks called the way a streaming caller does -- again on the rest of the
    input while it reports blocks (good blocks in front of a bad one come out first, the error at the head of the next call)."""
    L = zl.lib()
    p8 = lambda a, o=0: C.cast(a.ctypes.data + o, C.POINTER(C.c_uint8))
    s.set_state(init, 0)
    m = np.ascontiguousarray(m)
    out = np.empty(max(cap, 1), np.uint8)
    pos = produced = 0
    while pos < m.size:
        used, n = C.c_size_t(0), C.c_size_t(0)
        rc = L.zlng_decode_blocks(s._h, p8(m, pos), m.size - pos, C.byref(used), p8(out, produced), cap - produced, C.byref(n), None)
        if rc != 0:
            return rc, out[:produced]
        assert used.value > 0
        pos += used.value
        produced += n.value
    return 0, out[:produced]


def run_differential(zl, oracle, seed, count, classes=hostile.CLASSES):
    good_x = corpus.get("text_64k")
    good_z = oracle.encode(good_x, 0)
    bad, seen = [], set()
    with zl.Stream(0, 0, False, 4) as s:
        init, _ = s.get_state()
        for i, (name, m, cap) in enumerate(hostile.mutants(oracle, seed, count, classes)):
            rc, y, _flags = oracle.decode_ex(m, cap)
            code, got = gpu_verdict(zl, s, init, m, cap)
            seen.add(code)
            if code != CODE[rc] or got.size != y.size or not np.array_equal(got, y):
                bad.append((i, name, "oracle %d / %d B" % (rc, y.size), "gpu %d / %d B" % (code, got.size)))
            if i % 250 == 249:                                         # the context is still sound
                code, got = gpu_verdict(zl, s, init, good_z, good_x.size)
                assert code == 0 and np.array_equal(got, good_x), "context broken after mutant %d (%s)" % (i, name)
    return bad, seen


def test_hostile_streams_same_verdict_same_bytes_as_the_oracle(zl, oracle):
    bad, seen = run_differential(zl, oracle, 20260929, 3200)
    assert not bad, (len(bad), bad[:12])
    assert seen >= {0, -10, -11, -12, -13, -15, -16}                   # every reachable class was reached


def test_hostile_length_tables(zl, oracle):
    """Over- and under-subscribed length sets only: the decode tables of an over-subscribed set are defined by the reference's
    fill order and its 10-bit-first lookup (k_huff_decode computes every entry then)."""
    bad, _ = run_differential(zl, oracle, 77, 900, classes=("table",))
    assert not bad, (len(bad), bad[:12])


def test_hostile_crafted_token_streams(zl, oracle):
    bad, _ = run_differential(zl, oracle, 5, 600, classes=("crafted",))
    assert not bad, (len(bad), bad[:12])
