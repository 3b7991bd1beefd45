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


# ------------------------------------------------------------------------------ round 2: sharding, groups, repair, pools
def _mixed(nbytes, chunk, stretches, seed=21):
    """Text with incompressible stretches [(offset, length)] -- at e1-e4 every stretch flips current_level to 0 and back."""
    from oracle_py import textgen
    x = textgen(nbytes, chunk)
    rng = np.random.Generator(np.random.PCG64(seed))
    for off, ln in stretches:
        x[off:off + ln] = rng.integers(0, 256, ln, dtype=np.uint8)
    return x


def test_sharded_ranges_e4_three_ranks_with_batches(zl, oracle):
    """BASELINE config 4 in miniature, on one GPU: three "ranks" (block ranges), each fed through 2-3 contexts of one
    block (per-rank batching), e4, all parses queued before any state arrives, tables + current_level handed from context
    to context and rank to rank.  An incompressible stretch makes rank 0's range END at level 0 and another one crosses a
    context boundary inside rank 1 (src/libzling.cpp:185, 261-266).  Bytes must equal the single-stream oracle encoding."""
    import torch
    from libzling_amd import sharding
    B = zl.BLOCK
    total = 7 * B + 200_000
    x = _mixed(total, 33, [(2 * B - 600_000, 900_000), (3 * B - 400_000, 700_000), (5 * B + 1_000_000, 500_000)])
    ref = oracle.encode(x, 4)
    plan = [(0, 2 * B), (2 * B, 2 * B), (4 * B, total - 4 * B)]          # 2 + 2 + 4 blocks
    d_in = torch.cat([torch.from_numpy(x).cuda(), torch.zeros(512, dtype=torch.uint8, device="cuda")])
    encs, outs = [], []
    for off, n in plan:
        nb = (n + B - 1) // B
        e = sharding.RangeEncoder(lambda blocks: zl.Stream(0, 4, True, blocks), nb, 1 if nb <= 2 else 2)
        assert len(e.streams) >= 2
        encs.append(e)
        outs.append(torch.empty(zl.encode_bound(n) + 64, dtype=torch.uint8, device="cuda"))
    for e, (off, n) in zip(encs, plan):                                   # every rank parses before any state exists
        e.parse(d_in.data_ptr() + off, n)
    st = torch.zeros(sharding.STATE_BUF, dtype=torch.uint8, device="cuda")
    init, lv = encs[0].streams[0].get_state()
    st[: zl.MTF_STATE].copy_(torch.from_numpy(init)); torch.cuda.synchronize()
    parts, levels = [], []
    for e, o in zip(encs, outs):
        segs, lv = e.finish(o.data_ptr(), o.numel(), st.data_ptr(), lv)
        levels.append(lv)
        parts += [o[a:a + k].cpu().numpy() for a, k in segs]
    z = np.concatenate(parts)
    assert levels[0] == 0, "the first range ends inside the incompressible stretch: its exit level must be 0"
    assert z.size == ref.size and np.array_equal(z, ref), "first difference at byte %d" % int(np.argmax(z[: min(z.size, ref.size)] != ref[: min(z.size, ref.size)]))
    for e in encs:
        e.close()


@pytest.mark.parametrize("schedule", ["two_in_flight", "one_in_flight", "staggered", "auto"])
def test_range_through_five_contexts_under_every_parse_schedule(zl, oracle, schedule):
    """One range through five one-block contexts at e4 with the parses ordered by zlng_encode_parse_after (two / one in flight) and
    staggered in time by RangeEncoder's helper thread: the schedule changes WHEN a parse runs, never what it produces -- the bytes
    are the single-stream oracle encoding, twice in a row (a second step re-arms the events and the helper thread)."""
    import torch
    from libzling_amd import sharding
    B = zl.BLOCK
    total = 4 * B + 900_000
    x = _mixed(total, 71, [(B - 300_000, 500_000), (3 * B + 100_000, 400_000)])
    ref = oracle.encode(x, 4)
    d_in = torch.cat([torch.from_numpy(x).cuda(), torch.zeros(512, dtype=torch.uint8, device="cuda")])
    kw = {"two_in_flight": dict(parses_in_flight=2), "one_in_flight": dict(parses_in_flight=1),
          "staggered": dict(stagger=(2, 0.05)), "auto": dict(stagger="auto")}[schedule]
    enc = sharding.RangeEncoder(lambda blocks: zl.Stream(0, 4, True, blocks), 5, 1, **kw)
    assert len(enc.streams) == 5
    out = torch.empty(zl.encode_bound(total) + 64, dtype=torch.uint8, device="cuda")
    st = torch.zeros(sharding.STATE_BUF, dtype=torch.uint8, device="cuda")
    init, lv0 = enc.streams[0].get_state()
    for _ in range(2):
        st[: zl.MTF_STATE].copy_(torch.from_numpy(init)); torch.cuda.synchronize()
        enc.parse(d_in.data_ptr(), total)
        segs, lv = enc.finish(out.data_ptr(), out.numel(), st.data_ptr(), lv0)
        z = np.concatenate([out[a:a + k].cpu().numpy() for a, k in segs])
        assert z.size == ref.size and np.array_equal(z, ref)
    assert len(enc.stage_times()) == 5 and all(p > 0 and r > 0 for p, r, h in enc.stage_times())
    with zl.Stream(0, 0, False, 1) as dec:                               # a decode context is not a parse to wait for
        with pytest.raises(zl.ZlngError):
            enc.streams[1].parse_after(dec)
    enc.close()


@pytest.mark.parametrize("level", [0, 4])
def test_group_two_members_on_one_device(zl, oracle, level):
    """zlng_group: ONE stream over two contexts standing in for two devices (ZLNG_DEVICES=0,0 in the shim): bytes equal the
    oracle's at e0 and e4, in one call, in the split parse/finish form, and across two calls (the state lives in the group)."""
    B = zl.BLOCK
    x = _mixed(5 * B + 77_777, 44, [(2 * B + 500_000, 800_000), (3 * B - 300_000, 600_000)])
    ref = oracle.encode(x, level)
    with zl.Group([0, 0], level, 3) as g:
        z = g.encode(x)
        assert np.array_equal(z, ref)
        assert g.block_ends[-1] == z.size and all(z[e - 1] == 0 for e in g.block_ends)
    with zl.Group([0, 0], level, 2) as g:                                 # two calls: 4 blocks, then the tail
        a = g.encode(x[: 4 * B], split=True)
        b = g.encode(x[4 * B:])
        assert np.array_equal(np.concatenate([a, b]), ref)
        with pytest.raises(zl.ZlngError):
            g.encode(x)                                                   # 6 blocks > 2 members x 2


def test_level_adaptation_many_flips_is_repaired_in_few_passes(zl, oracle):
    """>= 10 level flips across >= 8 blocks at e4 (src/libzling.cpp:261-266).  Round 1 re-ran the whole range once per
    flip; the schedule repair now re-speculates every later sub-block from the measured ratios and re-parses only from the
    offending rank group on, so the mixed stream must cost a small multiple of a flip-free one."""
    import time
    B = zl.BLOCK
    n = 9 * B + 123_456
    stretches = [(int((k + 0.45) * 0.75 * B), 600_000) for k in range(11)]     # 11 stretches -> 22 flips, spread over 8+ blocks
    assert stretches[-1][0] + 600_000 < n and stretches[-1][0] // B >= 7
    x = _mixed(n, 55, stretches)
    from oracle_py import textgen
    plain = textgen(n, 56)
    with zl.Stream(0, 4, True, 10) as s:
        s.encode(plain[: 2 * B])                                          # warm-up (module load, pools)
    with zl.Stream(0, 4, True, 10) as s:
        t = time.perf_counter(); zp = s.encode(plain); t_plain = time.perf_counter() - t
    with zl.Stream(0, 4, True, 10) as s:
        t = time.perf_counter(); z = s.encode(x); t_mixed = time.perf_counter() - t
        passes = s.passes()
    ref = oracle.encode(x, 4)
    assert np.array_equal(z, ref), "first difference at byte %d" % int(np.argmax(z[: min(z.size, ref.size)] != ref[: min(z.size, ref.size)]))
    assert np.array_equal(zp, oracle.encode(plain, 4))
    print("e4, 22 flips over 8 blocks: %d parse passes, %.2f s vs %.2f s for flip-free text" % (passes, t_mixed, t_plain))
    assert passes <= 6, passes                                            # round 1: one full pass per flip
    assert t_mixed < 6.0 * t_plain + 2.0, (t_mixed, t_plain)


def test_huffman_lengths_kernel_on_tie_heavy_golden_tables(zl):
    """K4 fed directly with the tie-heavy frequency tables whose lengths the REFERENCE produced (tests/golden/huff_tables.npz,
    SURVEY H4: ties decide 40 of 67 real tables)."""
    d = np.load(os.path.join(G, "huff_tables.npz"))
    f1 = [(f, l) for f, l, (n, lim) in zip(d["freq"], d["len"], d["meta"]) if n == 514]
    f2 = [(f, l) for f, l, (n, lim) in zip(d["freq"], d["len"], d["meta"]) if n == 32]
    assert len(f1) == len(f2) == 150
    rows = np.zeros((150, 546), np.uint32)
    want = np.zeros((150, 546), np.uint8)
    for r in range(150):
        rows[r, :514] = f1[r][0][:514]; want[r, :514] = f1[r][1][:514]
        rows[r, 514:] = f2[r][0][:32]; want[r, 514:] = f2[r][1][:32]
    with zl.Stream(0, 0, True, 2) as s:
        lens, codes = s.debug_lengths(rows)
    bad = np.argwhere(lens != want)
    assert bad.size == 0, "row %d symbol %d: %d vs %d" % (bad[0][0], bad[0][1], lens[tuple(bad[0])], want[tuple(bad[0])])


def test_token_pool_overflow_grows_and_repeats(zl, oracle):
    """A block of incompressible data needs one token per byte, more than the default pool (0.44 per byte): the parser
    reports the overflow, the context grows its pools once and repeats the call; bytes still equal the oracle's."""
    rng = np.random.Generator(np.random.PCG64(77))
    from oracle_py import textgen
    x = np.concatenate([rng.integers(0, 256, zl.BLOCK, dtype=np.uint8), textgen(1_000_000, 9)])
    for lv in (0, 3):
        with zl.Stream(0, lv, True, 2) as s:
            z = s.encode(x)
            assert np.array_equal(z, oracle.encode(x, lv)), lv
            z2 = s.encode(x[zl.BLOCK:])                                   # the grown context keeps working (stream continues)
        st = oracle.lib.zo_stream_new(lv)
        import ctypes as C
        cap = oracle.lib.zo_encode_bound(x.size)
        o1 = np.empty(cap, np.uint8); o2 = np.empty(cap, np.uint8); n1 = C.c_size_t(0); n2 = C.c_size_t(0)
        p = lambda a: a.ctypes.data_as(C.POINTER(C.c_uint8))
        assert oracle.lib.zo_encode_blocks(st, p(x), x.size, p(o1), cap, C.byref(n1)) == 0
        tail = np.ascontiguousarray(x[zl.BLOCK:])
        assert oracle.lib.zo_encode_blocks(st, p(tail), tail.size, p(o2), cap, C.byref(n2)) == 0
        oracle.lib.zo_stream_free(st)
        assert np.array_equal(z2, o2[: n2.value]), lv


def test_failed_call_leaves_the_stream_state_untouched(zl, oracle):
    """ZLNG_E_CAP from the host entry points must not advance the MTF tables / current_level: the same call with enough
    room then produces the reference's bytes (the reference carries this state in its encoder object, src/libzling.cpp:180-197)."""
    import ctypes as C
    from oracle_py import textgen
    L = zl.lib()
    L.zlng_encode_parse.argtypes = [C.c_void_p, C.c_void_p, C.c_size_t]
    L.zlng_encode_finish.argtypes = [C.c_void_p, C.c_void_p, C.c_size_t, C.POINTER(C.c_size_t), C.POINTER(C.c_size_t)]
    x = textgen(zl.BLOCK + 500_000, 12)
    ref = oracle.encode(x, 0)
    a, b = x[: zl.BLOCK], np.ascontiguousarray(x[zl.BLOCK:])
    out = np.empty(zl.encode_bound(x.size), np.uint8)
    n = C.c_size_t(0)
    with zl.Stream(0, 0, True, 1) as s:
        za = s.encode(a)
        before, lv = s.get_state()
        assert L.zlng_encode_blocks(s._h, b.ctypes.data_as(C.POINTER(C.c_uint8)), b.size, out.ctypes.data_as(C.POINTER(C.c_uint8)), 100, C.byref(n), None) == -3
        assert np.array_equal(s.get_state()[0], before)
        assert L.zlng_encode_parse(s._h, b.ctypes.data, b.size) == 0
        assert L.zlng_encode_finish(s._h, out.ctypes.data, 100, C.byref(n), None) == -3
        assert np.array_equal(s.get_state()[0], before)
        assert L.zlng_encode_finish(s._h, out.ctypes.data, out.size, C.byref(n), None) == 0      # the range stayed pending
        assert np.array_equal(np.concatenate([za, out[: n.value]]), ref)
        with pytest.raises(zl.ZlngError):
            s.set_state(before, 3)                                        # current_level is 0 or the context's level


@pytest.mark.parametrize("parser", ["serial"])
def test_alternative_parsers_are_bit_exact(parser):
    """The one-lane serial form of the parser (k_rolz_parse_serial, ZLNG_PARSER=serial: the reference's loop token by token) is the
    on-device cross-check of the production parser: same bytes as the oracle at e0 and e4 on text with an incompressible stretch
    and a sub-block cut inside a window.  (Own process: the parser is chosen when the context is created.  The one-wavefront and
    the pipelined parsers of rounds 1-3 are retired: git history, scripts/experiments/retired/ up to commit 0899800.)"""
    import subprocess
    import sys
    code = r'''
import sys, numpy as np
sys.path[:0] = [%r, %r]
import libzling_amd as zl
from oracle_py import Oracle, textgen
n = 700_000 if %r == "serial" else 0
x = textgen(n, 17)
x[n // 2: n // 2 + 300_000] = np.random.Generator(np.random.PCG64(3)).integers(0, 256, 300_000, dtype=np.uint8)
o = Oracle()
for lv in (0, 4):
    assert np.array_equal(zl.encode(x, lv), o.encode(x, lv)), lv
print("ok")
''' % (os.path.dirname(G.rstrip("/")).rsplit("/tests", 1)[0], os.path.join(os.path.dirname(G.rstrip("/")).rsplit("/tests", 1)[0], "oracle"), parser)
    env = dict(os.environ, ZLNG_PARSER=parser)
    r = subprocess.run([sys.executable, "-c", code], env=env, stdout=subprocess.PIPE, stderr=subprocess.PIPE, timeout=900)
    assert r.returncode == 0 and b"ok" in r.stdout, r.stderr.decode()[-2000:]


def test_host_rank_chain_alternative_is_bit_exact():
    """ZLNG_HOST_RANK_CONTEXTS (opt-in, measured alternative: the longest rank chains on host threads, the rest on the device):
    same bytes as the oracle over several blocks and two calls of one stream, at e0 and e4.  Own process: the mode is chosen
    when the context is created."""
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    code = r'''
import sys, numpy as np
sys.path[:0] = [%r, %r]
import libzling_amd as zl
from oracle_py import Oracle, textgen
x = textgen(3 * zl.BLOCK + 500_000, 29)
o = Oracle()
for lv in (0, 4):
    want = o.encode(x, lv)
    with zl.Stream(0, lv, True, 2) as s:
        got = np.concatenate([s.encode(x[: 2 * zl.BLOCK]), s.encode(x[2 * zl.BLOCK:])])
    assert np.array_equal(got, want), lv
print("ok")
''' % (root, os.path.join(root, "oracle"))
    env = dict(os.environ, ZLNG_HOST_RANK_CONTEXTS="3")
    r = subprocess.run([sys.executable, "-c", code], env=env, stdout=subprocess.PIPE, stderr=subprocess.PIPE, timeout=900)
    assert r.returncode == 0 and b"ok" in r.stdout, r.stderr.decode()[-2000:]


@pytest.mark.parametrize("level", [0, 4])
def test_long_matches_across_a_block_end(zl, oracle, level):
    """Maximum-length matches up to the very end of a block and in a short block behind it: every window lane is "open"
    (its 16-byte compare ran through), the lanes near the block end have no room for a match (no sentinel), and the chain of
    settled lanes crosses into them.  A 1,000-byte period with a few edits, 16 MiB + 5,000 bytes; and all zeros, 17 MiB."""
    rng = np.random.Generator(np.random.PCG64(99))
    unit = rng.integers(97, 123, 1000, dtype=np.uint8)
    x = np.tile(unit, (zl.BLOCK + 5000) // 1000 + 1)[: zl.BLOCK + 5000].copy()
    for o in rng.integers(0, x.size, 200):
        x[o] ^= 1
    x[zl.BLOCK - 300: zl.BLOCK - 280] = 33            # a run that ends inside the last 275 bytes of block 0
    for data in (x, np.zeros(17 << 20, np.uint8)):
        z = zl.encode(data, level)
        ref = oracle.encode(data, level)
        assert z.size == ref.size and np.array_equal(z, ref)


def test_a_block_of_one_byte_tokens_fills_the_largest_token_pool_exactly(zl, oracle):
    """de Bruijn B(256, 3) XOR 0x55: a 16 MiB block the reference parses into 16,777,216 tokens, one per input byte -- the worst
    case of the token pools (kTokCapMax words per block).  The parser's overflow check must not fire on the pool that cannot
    overflow; the default pool overflows once and the call repeats on the grown pools (2 passes)."""
    from oracle_py import debruijn3
    x = debruijn3() ^ np.uint8(0x55)
    tok, cuts = oracle.parse_block(x, 0)
    assert tok.size == x.size == 1 << 24
    with zl.Stream(0, 0, True, 1) as s:
        z = s.encode(x)
        assert s.passes() == 2
        t, _c = s.block_tokens(0)
        assert t.size == 1 << 24
    ref = oracle.encode(x, 0)
    assert z.size == ref.size and np.array_equal(z, ref)


@pytest.mark.parametrize("level", [0, 4])
def test_hybrid_host_rank_chains_through_group_and_range_encoder(zl, oracle, level):
    """SURVEY 8(e) Option C wired through the multi-context drivers: zlng_group_set_host_rank_contexts on a group of three
    members and RangeEncoder.set_host_rank_contexts on a range fed through three contexts -- the k longest rank chains of every
    member's finish on host threads, the other chains on the device.  Bytes equal the oracle's; switching back to all-device
    on the same objects gives the same bytes again."""
    import torch
    from libzling_amd import sharding
    B = zl.BLOCK
    x = _mixed(5 * B + 300_000, 55, [(2 * B - 300_000, 700_000)])
    ref = oracle.encode(x, level)
    with zl.Group([0, 0, 0], level, 2) as g:
        g.set_host_rank_contexts(3)
        st, lv = g.get_state()
        assert np.array_equal(g.encode(x), ref)
        g.set_state(st, lv)
        g.set_host_rank_contexts(0)
        assert np.array_equal(g.encode(x), ref)
    n = x.size
    nb = (n + B - 1) // B
    d_in = torch.cat([torch.from_numpy(x).cuda(), torch.zeros(512, dtype=torch.uint8, device="cuda")])
    out = torch.empty(zl.encode_bound(n) + 64, dtype=torch.uint8, device="cuda")
    enc = sharding.RangeEncoder(lambda blocks: zl.Stream(0, level, True, blocks), nb, 2)
    assert len(enc.streams) == 3
    enc.set_host_rank_contexts(4)
    stt = torch.zeros(sharding.STATE_BUF, dtype=torch.uint8, device="cuda")
    init, lv0 = enc.streams[0].get_state()
    stt[: zl.MTF_STATE].copy_(torch.from_numpy(init)); torch.cuda.synchronize()
    enc.parse(d_in.data_ptr(), n)
    segs, _lv = enc.finish(out.data_ptr(), out.numel(), stt.data_ptr(), lv0)
    z = np.concatenate([out[a:a + k].cpu().numpy() for a, k in segs])
    assert z.size == ref.size and np.array_equal(z, ref)
    enc.close()


def test_rank_chain_on_a_context_whose_ranks_stay_above_64(zl, oracle):
    """The blank's context followed by 200 different bytes in rotation: every literal of that context has a rank >= 64, so every
    tile of its chain leaves the state-only statement at its first literal and is finished by slow_step + the re-entrant
    recording tile (mtf_rank.hip ZLNG_MTF_TILE_RE), entered at every step index over the run; mixed with text so that ranks
    21..63 and the neighbour-swap path occur in the same tiles."""
    from oracle_py import textgen
    rng = np.random.Generator(np.random.PCG64(44))
    n = 1_500_000
    x = textgen(n, 12).copy()
    pos = np.flatnonzero(x[:-1] == 32)                     # the byte behind every blank
    sym = (np.arange(pos.size) * 7 % 200 + 33).astype(np.uint8)
    keep = rng.random(pos.size) < 0.7                      # 70 % rotated (rank >= 64 mostly), the rest stay text
    x[pos[keep] + 1] = sym[keep]
    for lv in (0, 4):
        z = zl.encode(x, lv)
        assert np.array_equal(z, oracle.encode(x, lv)), lv
