"""The CPU oracle (oracle/zlng_oracle.c) against the committed golden vectors.

The vectors were produced by the real reference (tests/golden/make_golden.py); these
tests pin the oracle without needing /root/reference at run time.
"""
import os

import numpy as np
import pytest

import corpus

G = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")


def tok_to_u16(tok):
    """Expand one-word-per-token form into the reference's u16 stream."""
    sym = (tok & 0xFFFF).astype(np.uint16)
    aux = (tok >> 16).astype(np.uint16)
    is_m = sym >= 258
    out = np.empty(tok.size + int(is_m.sum()), np.uint16)
    pos = np.arange(tok.size) + np.concatenate([[0], np.cumsum(is_m)[:-1]]) if tok.size else np.empty(0, np.int64)
    out[pos] = sym
    out[pos[is_m] + 1] = aux[is_m]
    return out


def test_inputs_are_stable(manifest):
    for name, meta in manifest["inputs"].items():
        x = corpus.get(name)
        assert x.size == meta["size"] and corpus.sha(x) == meta["sha256"], name


@pytest.mark.parametrize("name", sorted(corpus.SMALL))
def test_small_streams_exact(oracle, name):
    x = np.fromfile(os.path.join(G, name + ".bin"), dtype=np.uint8)
    for lv in range(5):
        want = np.fromfile(os.path.join(G, "%s.e%d.zlng" % (name, lv)), dtype=np.uint8)
        got = oracle.encode(x, lv)
        assert np.array_equal(got, want), (name, lv)
        rc, back = oracle.decode(want, x.size)
        assert rc == 0 and np.array_equal(back, x), (name, lv)


@pytest.mark.parametrize("name", sorted(corpus.LARGE))
def test_large_streams_sha(oracle, manifest, name):
    x = corpus.get(name)
    for key, meta in manifest["streams"].items():
        if not key.startswith(name + ".e"):
            continue
        lv = int(key[-1])
        z = oracle.encode(x, lv)
        assert z.size == meta["size"] and corpus.sha(z) == meta["sha256"], key
        if lv == 0:
            rc, back = oracle.decode(z, x.size)
            assert rc == 0 and np.array_equal(back, x), key


def test_edge_sizes(manifest):
    assert manifest["streams"]["text_0.e0"]["size"] == 0       # empty in -> empty out
    assert manifest["streams"]["text_1.e0"]["size"] == 288     # 1 + 12 + 273 + 1 + 1


def test_carry_distinguishes_reset(oracle):
    """The 2-block fixture must differ from independently encoded blocks (MTF persists)."""
    x = corpus.get("carry_2blk")
    whole = oracle.encode(x, 0)
    b1 = oracle.encode(x[: corpus.BLOCK], 0)
    b2 = oracle.encode(x[corpus.BLOCK:], 0)
    assert np.array_equal(whole[: b1.size], b1)
    assert not np.array_equal(whole[b1.size:], b2)


@pytest.mark.parametrize("key", ["text_700k.e0", "text_700k.e4", "rand_1m.e0", "rand_1m.e4", "abc_1m.e0", "abc_1m.e4",
                                 "skew_400k.e0", "skew_400k.e4"])
def test_rolz_stage_golden(oracle, manifest, key):
    name, lv = key.rsplit(".e", 1)
    x = corpus.get(name)
    tok, cuts = oracle.parse_block(x, int(lv), apply_mtf=True)
    want = manifest["rolz"][key]
    assert [[c[1], c[2]] for c in cuts] == [list(c) for c in want["cuts"]]
    assert corpus.sha(tok_to_u16(tok)) == want["sha256_u16"]


def test_parse_is_mtf_independent(oracle):
    """the parse (everything but literal ranks) does not depend on MTF state."""
    x = corpus.get("text_700k")
    raw, c1 = oracle.parse_block(x, 0, apply_mtf=False)
    rk, c2 = oracle.parse_block(x, 0, apply_mtf=True)
    assert c1 == c2
    lit = ((raw & 0xFFFF) < 256) & ((raw >> 16) != 0xFFFF)
    assert np.array_equal(raw[~lit], rk[~lit])
    assert np.array_equal(raw >> 16, rk >> 16)


def test_huffman_length_tables(oracle):
    d = np.load(os.path.join(G, "huff_tables.npz"))
    for f, l, (n, limit) in zip(d["freq"], d["len"], d["meta"]):
        got = oracle.length_table(f[:n], int(limit))
        assert np.array_equal(got, l[:n])
        if got.any():
            assert got.max() <= limit


def test_tables_follow_generating_rule(oracle):
    nxt = oracle.table("zo_mtfnext", 256)
    assert all(nxt[i] == (int(i * 0.95) if i < 128 else int(i * 0.55)) for i in range(256))
    code = oracle.table("zo_matchidx_code", 4096)
    base = oracle.table("zo_matchidx_base", 32)
    blen = oracle.table("zo_matchidx_blen", 32)
    assert list(blen) == [0, 0, 0, 0, 1, 1, 2, 2, 3, 3, 4, 4, 5, 5, 6, 6, 7, 7] + [8] * 14
    for i in range(4096):
        c = code[i]
        assert base[c] <= i < base[c] + (1 << blen[c])
    assert sorted(oracle.mtfinit()) == list(range(256))


def test_decoder_rejects_corruption(oracle):
    x = corpus.get("text_64k")
    z = oracle.encode(x, 0)
    bad = z.copy(); bad[0] = 7
    assert oracle.decode(bad, x.size)[0] == -2          # invalid encflag
    bad = z.copy(); bad[5:9] = [0, 0x10, 0, 0]            # rlen 1048576 > 262144
    assert oracle.decode(bad, x.size)[0] == -3          # invalid block size
    bad = z.copy(); bad[1:5] = [0, 0, 0, 9]               # encpos mismatch
    assert oracle.decode(bad, x.size)[0] == -7          # lzdecode failed
