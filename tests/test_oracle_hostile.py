"""The oracle's verdicts on hostile streams against the REAL reference (oracle/_ref), on the CPU.

The GPU differential (tests/test_gpu_hostile.py) compares the HIP decoder with the oracle; this test is what entitles the oracle
to that role: for every mutant on which none of the oracle's own rules fired (flags == 0, zlng_oracle.h ZO_DEV_*) the reference --
run in a forked child, because on hostile input it may read or write memory it does not own -- must reach the same verdict (the
same exception message class, src/libzling.cpp:316, 327, 382, 392, 399, 407, or success) and must have written the same bytes
(on success all of them; on an error the complete blocks in front of it).  Where a rule did fire the reference's behaviour is
only counted: those are the documented deviations."""
import collections
import hashlib
import os
import pickle
import signal

import numpy as np
import pytest

import hostile
from oracle_py import Reference

MSG = {"invalid encflag.": -2, "invalid block size.": -3, "(bad code1)": -4, "(bad code2)": -5, "(bad ex-bits)": -6, "lzdecode failed.": -7}


def ref_verdict(ref, m, cap, timeout=3):
    """(class, bytes written, sha256 of them) from the reference in a forked child; ("crash", signal) if it dies."""
    r, w = os.pipe()
    pid = os.fork()
    if pid == 0:
        try:
            import faulthandler
            faulthandler.disable()                                      # a reference that dies here is a result, not a report
            os.close(r)
            signal.alarm(timeout)
            rc, y, msg = ref.decode(m, cap)
            cls = 0 if rc == 0 else (-1 if rc == -1 else next((v for k, v in MSG.items() if msg.endswith(k)), -99))
            os.write(w, pickle.dumps((cls, int(y.size), hashlib.sha256(y.tobytes()).hexdigest())))
        finally:
            os._exit(0)
    os.close(w)
    data = b""
    while True:
        chunk = os.read(r, 65536)
        if not chunk:
            break
        data += chunk
    os.close(r)
    _, status = os.waitpid(pid, 0)
    if not data:
        return ("crash", os.WTERMSIG(status) if os.WIFSIGNALED(status) else -1, None)
    return pickle.loads(data)


@pytest.mark.skipif(not Reference.available(), reason="oracle/_ref not built (no /root/reference here)")
def test_oracle_verdicts_are_the_references_wherever_the_reference_is_defined(oracle):
    ref = Reference()
    stats = collections.Counter()
    deviations = collections.Counter()
    bad = []
    for name, m, cap in hostile.mutants(oracle, 20260929, 1600):
        rc, y, flags = oracle.decode_ex(m, cap)
        stats[(name.split(":")[0], rc)] += 1
        if flags & 32:                                                  # ZO_DEV_SELF: the reference's copy loop never ends on a source
            deviations[(flags, "reference: hangs")] += 1                 # that is its destination (src/libzling_lz.cpp:91-95)
            continue
        got = ref_verdict(ref, m, cap)
        if flags:
            deviations[(flags, "same verdict" if got[0] == rc else "reference: %s" % (got[0],))] += 1
            continue
        mine = (rc, int(y.size), hashlib.sha256(y.tobytes()).hexdigest())
        if tuple(got) != mine:
            bad.append((name, mine[:2], tuple(got)[:2]))
    assert not bad, bad[:10]
    # every verdict class of the decoder was reached without a deviation rule, i.e. was held against the reference
    assert {rc for (_, rc) in stats} >= {0, -2, -3, -4, -5, -7, -8}      # (-6, the ex-bits test, cannot fire: the codes name at most index 4095)
    print(sorted(deviations.items()))


@pytest.mark.skipif(not Reference.available(), reason="oracle/_ref not built (no /root/reference here)")
def test_over_subscribed_tables_decode_like_the_reference_two_level_lookup(oracle):
    """An over-subscribed length set makes table entries collide: the reference fills in symbol order (the last symbol wins,
    src/libzling_huffman.cpp:140-153) and asks its 10-bit fast table first (src/libzling.cpp:361, 376-379), so a short code beats a
    longer one of a later symbol.  Mutants that ONLY touch table bytes, many of them, success or not."""
    ref = Reference()
    n_ok = 0
    for name, m, cap in hostile.mutants(oracle, 77, 900, classes=("table",)):
        rc, y, flags = oracle.decode_ex(m, cap)
        if flags:
            continue
        got = ref_verdict(ref, m, cap)
        assert tuple(got) == (rc, int(y.size), hashlib.sha256(y.tobytes()).hexdigest()), name
        n_ok += rc == 0
    assert n_ok >= 20                                                   # damaged tables that still decode: the interesting half


def test_oracle_decoder_is_memory_safe_on_hostile_streams(oracle):
    """No reference needed: the restatement's decoder over 4,000 mutants -- every verdict one of its own codes, no more bytes reported
    than the capacity, and (under scripts/sanitize.sh cpu, which runs this suite on the ASan + UBSan build of the oracle) no read or
    write outside its buffers, which is the reason its deviation rules exist."""
    seen = collections.Counter()
    for name, m, cap in hostile.mutants(oracle, 424242, 4000):
        rc, y, flags = oracle.decode_ex(m, cap)
        assert rc in (0, -1, -2, -3, -4, -5, -6, -7, -8) and y.size <= cap and 0 <= flags < 256, (name, rc, flags)
        seen[rc] += 1
    assert seen[0] > 100 and seen[-7] > 100 and seen[-8] > 100


def _brev(v, bits):
    return int(format(v, "0%db" % bits)[::-1], 2) if bits else 0


def _lut_entry_exact(i, lens, limit, fast):
    """csrc/decode.hip lut_entry_exact, restated: the decode-table entry of index i computed -- not filled -- from the symbols of every
    length in symbol order: of a length's symbols whose code matches i, the largest; any length <= `fast` in front of the longer ones."""
    best_fast = best_slow = -1
    code = 0
    for L in range(1, limit + 1):
        syms = [c for c, l in enumerate(lens) if l == L]
        start = code
        code = (code + len(syms)) * 2
        if not syms:
            continue
        m = (1 << L) - 1
        q = _brev(i & m, L)
        k0 = (q - start) & m
        if k0 >= len(syms):
            continue
        c = syms[k0 + (((len(syms) - 1 - k0) >> L) << L)]
        if L <= fast:
            best_fast = max(best_fast, c)
        else:
            best_slow = max(best_slow, c)
    return best_fast if best_fast >= 0 else (best_slow if best_slow >= 0 else 0xFFFF)


def test_computed_decode_table_entries_equal_the_reference_fill_order(oracle):
    """The HIP decoder cannot FILL the decode table of an over-subscribed length set the way the reference does (symbols in order, a
    later one overwriting an earlier one, src/libzling_huffman.cpp:140-153: lanes would race), so k_huff_decode computes each entry
    (lut_entry_exact).  The same formula in Python against tables filled in the reference's order from the oracle's own codes
    (ZlingMakeEncodeTable restated), with the 10-bit fast table in front for alphabet 1 (src/libzling.cpp:361, 376-379): random
    over-, under- and exactly subscribed sets of both alphabets, every entry."""
    rng = np.random.Generator(np.random.PCG64(31))

    def filled(lens, limit, bits):
        codes = oracle.encode_table(np.asarray(lens, np.uint32), limit)
        lut = [0xFFFF] * (1 << bits)
        for c, l in enumerate(lens):
            if 0 < l <= bits:
                for i in range(int(codes[c]), 1 << bits, 1 << l):
                    lut[i] = c
        return lut
    for trial in range(40):
        n, limit, fast = (514, 15, 10) if trial % 2 == 0 else (32, 8, 8)
        kind = trial % 5
        if kind == 0: lens = rng.integers(0, 16, n)
        elif kind == 1: lens = np.where(rng.random(n) < 0.1, rng.integers(1, 6, n), 0)
        elif kind == 2: lens = np.where(rng.random(n) < 0.5, rng.integers(8, 16, n), 0)
        elif kind == 3: lens = np.full(n, int(rng.integers(1, 16)))
        else: lens = np.where(rng.random(n) < 0.03, rng.integers(1, 16, n), 0)
        lens = [int(v) for v in lens]
        full = filled(lens, limit, limit)
        if n == 514:
            fast_t = filled(lens, limit, fast)
            want = [fast_t[i & ((1 << fast) - 1)] if fast_t[i & ((1 << fast) - 1)] != 0xFFFF else full[i] for i in range(1 << limit)]
            idx = rng.integers(0, 1 << limit, 600)
        else:
            want = full
            idx = range(1 << limit)
        for i in idx:
            assert _lut_entry_exact(int(i), lens, limit, fast) == want[int(i)], (trial, int(i))


def test_big_hostile_streams_reach_the_big_structures(oracle):
    """What tests/hostile.py's big set claims, measured by the checker's decoder (zo_decode_stats): a wrapped ring with matches taken
    from it, copies from further back than the replay kernel's 64 KiB LDS window and than 128 KiB, copies whose destination and
    source straddle a 64 KiB boundary, a full 16 MiB block, several sub-blocks at a generic level."""
    rng = np.random.Generator(np.random.PCG64(1))
    name, z, size = hostile.big_crafted(oracle, rng, 0)
    rc, st = oracle.decode_stats(z, size + 4096)
    assert rc == 0 and st["max_inserts_one_context"] > 4096 and st["matches_in_wrapped_ring"] > 50, (name, st)
    name, z, size = hostile.big_crafted(oracle, rng, 1)
    rc, st = oracle.decode_stats(z, size + 4096)
    assert rc == 0 and st["far_matches"] >= 10 and st["dst_straddles_64k"] >= 2 and st["src_straddles_64k"] >= 2 and st["max_distance"] > 131072, (name, st)
    name, z, size = hostile.big_crafted(oracle, rng, 2)
    rc, st = oracle.decode_stats(z, size + 4096)
    assert rc == 0 and st["matches_in_wrapped_ring"] > 1000 and st["far_matches"] > 1000 and len(hostile.walk(z)[0]) == 3, (name, st)
    for name, z, cap in hostile.big_bases(oracle):
        rc, st = oracle.decode_stats(z, cap)
        assert rc == 0 and st["max_inserts_one_context"] > 100 * 4096 // (8 if "3m" in name else 1) and st["far_matches"] > 10000, (name, st)
        subs, ends = hostile.walk(z)
        if name.startswith("text_16m"):
            assert len(ends) == 2 and max(e for _, _, e, _, _ in subs) == 16777216          # a full block
        if name.endswith(".e4"):
            assert len(subs) >= 3


@pytest.mark.skipif(not Reference.available(), reason="oracle/_ref not built (no /root/reference here)")
def test_oracle_equals_the_reference_on_big_hostile_streams(oracle):
    """The same differential as above on streams that reach the big structures (no mutant of the small set reaches a
    wrapped ring, a full block or the LDS window's wrap): every mutant without a deviation rule must get the reference's verdict and
    bytes; the crafted ones that are left undamaged must DECODE in both.  The share decided by a ZO_DEV_* rule stays under 35 %."""
    ref = Reference()
    stats = collections.Counter()
    bad, n, dev = [], 0, 0
    for name, m, cap in hostile.big_mutants(oracle, 20260930, 200):
        rc, y, flags = oracle.decode_ex(m, cap)
        n += 1
        stats[rc] += 1
        if name.startswith("big:") and "too-many" not in name:
            assert rc == 0, name                                        # a valid crafted stream
        if flags:
            dev += 1
            continue
        got = ref_verdict(ref, m, cap, timeout=10)
        mine = (rc, int(y.size), hashlib.sha256(y.tobytes()).hexdigest())
        if tuple(got) != mine:
            bad.append((name, mine[:2], tuple(got)[:2]))
    assert not bad, bad[:10]
    assert dev / n < 0.35, (dev, n)
    assert stats[0] >= 20 and stats[-7] >= 40 and stats[-4] + stats[-5] >= 5
    print("big hostile: %d mutants, %d by a deviation rule (%.0f %%), verdicts %s" % (n, dev, 100.0 * dev / n, dict(stats)))


def test_blockwise_decoder_equals_the_whole_stream_decoder_and_rolls_back(oracle):
    """zo_dstream_decode_block (the block-at-a-time form with carried tables that the C-ABI stand-in of the CPU suite is built on,
    tests/cxx/zlng_stub.c) against zo_decode_ex: same verdict, same bytes in front of the error, on valid multi-block streams and on
    hostile mutants; and an error leaves the tables as the last good block left them -- the same good block decodes the same way after
    a failed one as before it."""
    n_err = 0
    for name, m, cap in hostile.mutants(oracle, 606, 700):
        rc, y, _flags = oracle.decode_ex(m, cap)
        rc2, y2, nblk, mtf = oracle.decode_blockwise(m, cap)
        assert rc2 == rc and np.array_equal(y2, y), name
        n_err += rc != 0
    assert n_err > 200
    import corpus
    x = corpus.get("carry_2blk")
    z = oracle.encode(x, 0)
    subs, ends = hostile.walk(z)
    b1 = z[: ends[0] + 1]                                              # the first block alone
    rc, y1, nblk, t1 = oracle.decode_blockwise(b1, x.size)
    assert rc == 0 and nblk == 1
    bad = z[ends[0] + 1:].copy(); bad[13: 13 + 257] = 0                 # the second block with an empty length table
    rc, _y, nblk, t_after_bad = oracle.decode_blockwise(bad, x.size, state=t1)
    assert rc == -4 and nblk == 0 and np.array_equal(t_after_bad, t1)   # bad code1; tables untouched
    rc, y2, nblk, _t = oracle.decode_blockwise(z[ends[0] + 1:], x.size, state=t_after_bad)
    assert rc == 0 and np.array_equal(np.concatenate([y1, y2]), x)      # ... and the real second block still decodes on them
