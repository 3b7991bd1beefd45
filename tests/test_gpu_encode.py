"""GPU parity: the HIP encode path (through the C-ABI) against golden vectors and the oracle."""
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


@pytest.mark.parametrize("name", sorted(corpus.SMALL))
def test_small_golden_exact(zl, name):
    x = np.fromfile(os.path.join(G, name + ".bin"), dtype=np.uint8)
    for lv in range(5):
        want = np.fromfile(os.path.join(G, "%s.e%d.zlng" % (name, lv)), dtype=np.uint8)
        got = zl.encode(x, lv)
        assert np.array_equal(got, want), (name, lv, got.size, want.size)


@pytest.mark.parametrize("name", ["text_700k", "rand_1m", "zeros_1m", "abc_1m", "skew_400k", "skew2_600k", "mixed_e4"])
def test_large_golden_sha(zl, manifest, oracle, name):
    x = corpus.get(name)
    for key, meta in sorted(manifest["streams"].items()):
        if not key.startswith(name + ".e"):
            continue
        lv = int(key[-1])
        z = zl.encode(x, lv)
        if not (z.size == meta["size"] and corpus.sha(z) == meta["sha256"]):
            ref = oracle.encode(x, lv)
            bad = int(np.argmax(z[: min(z.size, ref.size)] != ref[: min(z.size, ref.size)]))
            pytest.fail("%s: size %d vs %d, first difference at byte %d" % (key, z.size, meta["size"], bad))


@pytest.mark.parametrize("name", ["carry_2blk", "text_33m"])
def test_multiblock_golden_sha(zl, manifest, name):
    """Cross-block MTF carry against the reference's streams."""
    x = corpus.get(name)
    for lv in (0, 4):
        meta = manifest["streams"]["%s.e%d" % (name, lv)]
        z = zl.encode(x, lv)
        assert z.size == meta["size"] and corpus.sha(z) == meta["sha256"], (name, lv)


def test_many_blocks_in_flight_vs_oracle(zl, oracle):
    """7 blocks parsed concurrently, > 256 sub-block workgroups in the Huffman kernels."""
    from oracle_py import textgen
    x = textgen(7 * zl.BLOCK - 12345, 40)
    with zl.Stream(0, 0, True, 7) as s:
        z = s.encode(x)
        ends = s.block_ends
    ref = oracle.encode(x, 0)
    assert z.size == ref.size
    assert np.array_equal(z, ref), "first difference at byte %d" % int(np.argmax(z != ref))
    assert ends[-1] == z.size and all(z[e - 1] == 0 for e in ends)      # every block ends with its 0x00
    for it in range(2):                                                   # and the result is reproducible
        assert np.array_equal(zl.encode(x, 0), z)


def test_stage_parity_per_subblock(zl, oracle):
    """Every kernel against the oracle's stage API: K1 tokens + cuts, K2 ranks, K3 freq, K4 lengths/codes, K6 payload."""
    from oracle_py import textgen
    nb = 4
    x = np.concatenate([textgen(3 * zl.BLOCK, 60), np.random.Generator(np.random.PCG64(5)).integers(0, 256, 700_000, dtype=np.uint8)])
    st = oracle.lib.zo_stream_new(0)
    with zl.Stream(0, 0, True, nb) as s:
        z = s.encode(x)
        for b in range(nb):
            blk = x[b * zl.BLOCK:(b + 1) * zl.BLOCK]
            otok, ocuts = oracle.parse_block(blk, 0, apply_mtf=True, stream=st)
            tok, cuts = s.block_tokens(b)
            assert np.array_equal(tok, otok), "tokens of block %d" % b
            assert [tuple(int(v) for v in c[1:]) for c in cuts] == [c for c in ocuts]
            freq = s.debug_fetch(2, b, np.uint32, 80 * 546).reshape(80, 546)
            lens = s.debug_fetch(3, b, np.uint8, 80 * 546).reshape(80, 546)
            codes = s.debug_fetch(6, b, np.uint16, 80 * 546).reshape(80, 546)
            olen = s.debug_fetch(4, b, np.uint32, 80)
            off = s.debug_fetch(7, b, np.uint64, 80)
            for k in range(cuts.shape[0]):
                t = tok[cuts[k, 0]:cuts[k, 1]]
                f1, f2 = oracle.histogram(t)
                l1, l2 = oracle.length_table(f1, 15), oracle.length_table(f2, 8)
                assert np.array_equal(freq[k, :514], f1) and np.array_equal(freq[k, 514:], f2), (b, k)
                assert np.array_equal(lens[k, :514], l1) and np.array_equal(lens[k, 514:], l2), (b, k)
                assert np.array_equal(codes[k, :514], oracle.encode_table(l1, 15)), (b, k)
                assert np.array_equal(codes[k, 514:], oracle.encode_table(l2, 8)), (b, k)
                pay = oracle.pack(t, l1, l2)
                assert pay.size == olen[k], (b, k)
                assert np.array_equal(z[int(off[k]) + 13: int(off[k]) + 13 + pay.size], pay), (b, k)
    oracle.lib.zo_stream_free(st)


def test_split_parse_finish_and_state_handoff(zl, oracle):
    """Block-range sharding: two contexts encode halves of one stream, the MTF state is handed over."""
    import torch
    from oracle_py import textgen
    x = textgen(2 * zl.BLOCK + 300_000, 70)
    ref = oracle.encode(x, 0)
    a, b = x[: zl.BLOCK], x[zl.BLOCK:]
    outs = []
    with zl.Stream(0, 0, True, 1) as s0, zl.Stream(0, 0, True, 2) as s1:
        da = torch.cat([torch.from_numpy(a).cuda(), torch.zeros(512, dtype=torch.uint8, device="cuda")])
        db = torch.cat([torch.from_numpy(b).cuda(), torch.zeros(512, dtype=torch.uint8, device="cuda")])
        oa = torch.empty(zl.encode_bound(a.size), dtype=torch.uint8, device="cuda")
        ob = torch.empty(zl.encode_bound(b.size), dtype=torch.uint8, device="cuda")
        s1.parse_device(db.data_ptr(), b.size)            # rank 1 parses before it has the state
        na = s0.encode_device(da.data_ptr(), a.size, oa.data_ptr(), oa.numel())
        st = torch.empty(zl.MTF_STATE, dtype=torch.uint8, device="cuda")
        lv = s0.get_state_device(st.data_ptr())
        s1.set_state_device(st.data_ptr(), lv)
        nb_ = s1.finish_device(ob.data_ptr(), ob.numel())
        outs = [oa[:na].cpu().numpy(), ob[:nb_].cpu().numpy()]
    assert np.array_equal(np.concatenate(outs), ref)


def test_rejects_bad_arguments(zl):
    with pytest.raises(zl.ZlngError):
        zl.Stream(0, 7, True, 1)                            # level outside 0..4 (reference silently corrupts)
    with zl.Stream(0, 0, True, 1) as s:
        with pytest.raises(zl.ZlngError):
            s.encode(np.zeros(2 * zl.BLOCK, np.uint8))      # more blocks than the context was sized for
        assert s.encode(np.zeros(0, np.uint8)).size == 0    # empty input -> empty stream


def test_split_host_api_rejects_misuse(zl):
    """zlng_encode_finish without a pending parse, a parse on a decode context and oversized ranges fail with
    ZLNG_E_ARG instead of touching memory."""
    import ctypes as C
    L = zl.lib()
    L.zlng_encode_parse.argtypes = [C.c_void_p, C.c_void_p, C.c_size_t]
    L.zlng_encode_finish.argtypes = [C.c_void_p, C.c_void_p, C.c_size_t, C.POINTER(C.c_size_t), C.POINTER(C.c_size_t)]
    x = np.zeros(1000, np.uint8)
    out = np.empty(zl.encode_bound(x.size), np.uint8)
    n = C.c_size_t(0)
    with zl.Stream(0, 0, True, 1) as s:
        assert L.zlng_encode_finish(s._h, out.ctypes.data, out.size, C.byref(n), None) == -1          # nothing parsed
        assert L.zlng_encode_parse(s._h, x.ctypes.data, 0) == -1
        big = np.zeros(2 * zl.BLOCK, np.uint8)
        assert L.zlng_encode_parse(s._h, big.ctypes.data, big.size) == -1                             # > max_blocks
        assert L.zlng_encode_parse(s._h, x.ctypes.data, x.size) == 0
        assert L.zlng_encode_finish(s._h, out.ctypes.data, 4, C.byref(n), None) == -3                 # ZLNG_E_CAP
    with zl.Stream(0, 0, False, 1) as d:
        assert L.zlng_encode_parse(d._h, x.ctypes.data, x.size) == -1                                 # decode context
