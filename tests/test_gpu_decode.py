"""GPU parity of the decode path (K7 frame walk, K8 Huffman decode, K9 ROLZ+MTF replay) through the C-ABI."""
import os

import numpy as np
import pytest

import corpus

pytestmark = pytest.mark.gpu
G = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")


@pytest.fixture(scope="module")
def zl():
    import libzling_amd as zl
    assert zl.lib().zlng_device_count() >= 1, "no gfx950 device visible"
    return zl


def gpu_decode(zl, z, cap, nb=2):
    with zl.Stream(0, 0, False, nb) as s:
        return s.decode(z, cap)


@pytest.mark.parametrize("name", sorted(corpus.SMALL))
def test_small_golden_streams_decode(zl, name):
    x = np.fromfile(os.path.join(G, name + ".bin"), dtype=np.uint8)
    for lv in (0, 2, 4):
        z = np.fromfile(os.path.join(G, "%s.e%d.zlng" % (name, lv)), dtype=np.uint8)
        back = gpu_decode(zl, z, max(x.size, 1))
        assert np.array_equal(back, x), (name, lv)


@pytest.mark.parametrize("name", ["text_700k", "rand_1m", "zeros_1m", "abc_1m", "skew_400k", "mixed_e4"])
def test_large_reference_streams_decode(zl, oracle, name):
    x = corpus.get(name)
    for lv in (0, 4):
        z = oracle.encode(x, lv)                 # pinned to the reference's bytes by test_oracle_golden.py
        assert np.array_equal(gpu_decode(zl, z, x.size), x), (name, lv)


def test_two_block_stream_carries_mtf(zl, oracle):
    x = corpus.get("carry_2blk")
    z = oracle.encode(x, 0)
    assert np.array_equal(gpu_decode(zl, z, x.size, nb=2), x)


def test_gpu_roundtrip_of_gpu_stream(zl):
    from oracle_py import textgen
    x = np.concatenate([textgen(900_000, 81), np.zeros(70_000, np.uint8), textgen(200_000, 82)])
    z = zl.encode(x, 3)
    assert np.array_equal(gpu_decode(zl, z, x.size), x)


def test_corrupt_streams_map_to_reference_errors(zl, oracle):
    x = corpus.get("text_64k")
    z = oracle.encode(x, 0)

    def code_of(bad):
        with pytest.raises(zl.ZlngError) as e:
            gpu_decode(zl, bad, x.size)
        return e.value.code

    bad = z.copy(); bad[0] = 7
    assert code_of(bad) == -10                                   # "invalid encflag."
    bad = z.copy(); bad[5:9] = [0, 0x10, 0, 0]
    assert code_of(bad) == -11                                   # "invalid block size."
    bad = z.copy(); bad[1:5] = [0, 0, 0, 9]
    assert code_of(bad) == -15                                   # "lzdecode failed."
    assert code_of(z[:-1]) == -16                                # block not closed: truncated
    assert "invalid encflag" in zl.strerror(-10) and "lzdecode failed" in zl.strerror(-15)
    # corrupt bits inside the payload either still decode to a stream of the right length that
    # differs, or raise one of the stream errors -- never crash, never hang
    rng = np.random.Generator(np.random.PCG64(3))
    for _ in range(20):
        bad = z.copy()
        bad[int(rng.integers(300, z.size - 2))] ^= 1 << int(rng.integers(0, 8))
        try:
            back = gpu_decode(zl, bad, x.size)
            assert back.size <= x.size
        except zl.ZlngError as e:
            assert e.code in (-12, -13, -14, -15, -1)


def test_each_huffman_validity_check_deterministically(zl, oracle):
    x, cases = corpus.corrupt_cases(oracle)
    want_gpu = {"code1": -12, "code2": -13, "lz": -15}
    for name, bad, ocode in cases:
        assert oracle.decode(bad, x.size)[0] == ocode, name
        with pytest.raises(zl.ZlngError) as e:
            gpu_decode(zl, bad, x.size)
        assert e.value.code == want_gpu[name], (name, e.value.code)


def test_full_size_decode_round_trip(zl):
    """BASELINE config 5: the e0 .zlng of the 10^9-byte stream decodes back to the input (SHA-256), 60 blocks in one call."""
    import hashlib
    from oracle_py import textgen
    n = 1_000_000_000
    x = textgen(n, 0)
    nb = (n + zl.BLOCK - 1) // zl.BLOCK
    with zl.Stream(0, 0, True, nb) as s:
        z = s.encode(x)
    with zl.Stream(0, 0, False, nb) as d:
        back = d.decode(z, n)
    assert back.size == n
    assert hashlib.sha256(back.tobytes()).digest() == hashlib.sha256(x.tobytes()).digest()


def test_good_blocks_are_reported_before_a_bad_one(zl, oracle):
    """A corrupt sub-block in the second block of a prefix: the first block is decoded and reported (the reference writes every
    block before it throws, src/libzling.cpp:306-420); the next call meets the bad block alone and returns its error; the
    tables the context keeps are those at the start of the bad block, so a repaired stream continues correctly."""
    import ctypes as C
    from oracle_py import textgen
    x = textgen(2 * zl.BLOCK + 100_000, 23)
    z = oracle.encode(x, 0)
    # end of block 0 = the only 0x00 flag byte that follows a whole number of sub-blocks; find it by walking the frame
    p, ends = 0, []
    while p < z.size:
        if z[p] == 0:
            ends.append(p + 1); p += 1; continue
        p += 13 + int.from_bytes(z[p + 9:p + 13].tobytes(), "big")
    bad = z.copy()
    bad[ends[0] + 13: ends[0] + 13 + 257] = 0                       # block 1, first sub-block: no code for any symbol -> bad code1
    L = zl.lib()
    out = np.empty(3 * zl.BLOCK, np.uint8)
    used, n = C.c_size_t(0), C.c_size_t(0)
    bends = (C.c_size_t * 4)()
    p8 = lambda a: a.ctypes.data_as(C.POINTER(C.c_uint8))
    with zl.Stream(0, 0, False, 4) as d:
        rc = L.zlng_decode_blocks(d._h, p8(bad), bad.size, C.byref(used), p8(out), out.size, C.byref(n), bends)
        assert rc == 0 and used.value == ends[0] and n.value == zl.BLOCK and bends[0] == zl.BLOCK
        assert np.array_equal(out[: zl.BLOCK], x[: zl.BLOCK])
        rest = np.ascontiguousarray(bad[ends[0]:])
        rc = L.zlng_decode_blocks(d._h, p8(rest), rest.size, C.byref(used), p8(out), out.size, C.byref(n), bends)
        assert rc == -12 and n.value == 0                              # "invalid huffman stream. (bad code1)"
        good = np.ascontiguousarray(z[ends[0]:])                       # the state survived the failed call
        rc = L.zlng_decode_blocks(d._h, p8(good), good.size, C.byref(used), p8(out), out.size, C.byref(n), bends)
        assert rc == 0 and np.array_equal(out[: n.value], x[zl.BLOCK:])


def test_cli_decodes_an_unterminated_final_block(tmp_path, oracle):
    """The reference's inner loop also ends at end of input (src/libzling.cpp:312-313): a last block that lacks its 0x00
    terminator is still written.  Through the C++ API (tools/zling_demo)."""
    import subprocess
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    demo = os.environ.get("ZLNG_DEMO") or os.path.join(root, "tools", "zling_demo")
    x = corpus.get("text_64k")
    z = oracle.encode(x, 0)
    assert z[-1] == 0
    q = subprocess.run([demo, "d"], input=z[:-1].tobytes(), stdout=subprocess.PIPE, stderr=subprocess.PIPE)
    assert q.returncode == 0 and q.stdout == x.tobytes(), q.stderr[-300:]
    q = subprocess.run([demo, "d"], input=z[:-40].tobytes(), stdout=subprocess.PIPE, stderr=subprocess.PIPE)
    assert q.returncode != 0                                           # cut inside a payload: still an error


def test_replay_loop_on_runs_overlaps_and_far_sources(zl, oracle):
    """The generated, software-pipelined token loop (k_rolz_replay; csrc/replay_loop.h) on what its special cases are for: text,
    long runs (matches longer than one wavefront, overlapping copies with periods of 1..63), word-MRU tokens, incompressible bytes
    and a match source beyond the 64 KiB LDS window."""
    from oracle_py import textgen
    rng = np.random.Generator(np.random.PCG64(33))
    far = textgen(300_000, 7)
    parts = [textgen(500_000, 5), np.zeros(70_000, np.uint8), np.tile(np.frombuffer(b"abcabcabd", np.uint8), 9_000),
             rng.integers(0, 256, 120_000, dtype=np.uint8), far, textgen(200_000, 6), far[:150_000],     # a copy from 350 KB back
             np.tile(np.frombuffer(b"the quick brown fox ", np.uint8), 4_000)]
    parts += [np.tile(np.arange(p, dtype=np.uint8) + 65, 3000 // p + 40) for p in range(1, 64)]          # every copy period 1..63
    x = np.concatenate(parts)
    for lv in (0, 4):
        z = oracle.encode(x, lv)
        assert np.array_equal(gpu_decode(zl, z, x.size), x), lv
