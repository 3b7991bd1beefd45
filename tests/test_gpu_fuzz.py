"""Randomised differential test of the whole encode path (GPU vs oracle) and round trip through the GPU decoder.

The reference's own fuzz test feeds random bytes only (test/fuzzy/libzling_fuzzy.py:20-42), which never exercises
matches, lazy parsing, the word MRU or conflicts between token starts that share a hash slot; these inputs do:
text with edits, tiny alphabets, periodic data with drifting periods, runs, sparse noise, mixtures, all levels.
"""
import numpy as np
import pytest

pytestmark = pytest.mark.gpu


def make_input(rng, kind, n, text):
    if kind == 0:                                   # text slice with random edits
        o = int(rng.integers(0, text.size - n))
        x = text[o:o + n].copy()
        idx = rng.integers(0, max(n, 1), n // 40)
        x[idx] = rng.integers(0, 256, idx.size, dtype=np.uint8)
        return x
    if kind == 1:                                   # tiny alphabet: dense hash-slot conflicts, long chains
        return rng.integers(0, int(rng.integers(1, 5)), n, dtype=np.uint8)
    if kind == 2:                                   # periodic with drifting period and occasional breaks
        out, p, tot = [], int(rng.integers(1, 40)), 0
        while tot < n:
            unit = rng.integers(97, 123, p, dtype=np.uint8)
            out.append(np.tile(unit, int(rng.integers(2, 60)))); tot += out[-1].size
            if rng.random() < 0.3:
                p = int(rng.integers(1, 40))
            if rng.random() < 0.2:
                out.append(rng.integers(0, 256, int(rng.integers(1, 9)), dtype=np.uint8)); tot += out[-1].size
        return np.concatenate(out)[:n]
    if kind == 3:                                   # runs of few symbols, varying lengths (word-MRU and self-hits)
        syms = rng.integers(0, 256, 6, dtype=np.uint8)
        lens = rng.integers(1, 400, n // 100 + 8)
        which = syms[rng.integers(0, 6, lens.size)]
        return np.repeat(which, lens)[:n] if lens.sum() >= n else np.resize(np.repeat(which, lens), n)
    if kind == 4:                                   # repeated short words separated by one of two separators
        words = [rng.integers(97, 123, int(rng.integers(1, 6)), dtype=np.uint8) for _ in range(int(rng.integers(2, 30)))]
        pick = rng.integers(0, len(words), n // 2 + 4)
        sep = np.where(rng.random(pick.size) < 0.8, 32, 10).astype(np.uint8)
        parts = [np.concatenate([words[int(w)], sep[i:i + 1]]) for i, w in enumerate(pick[: n // 2 + 4])]
        return np.concatenate(parts)[:n]
    if kind == 5:                                   # incompressible / compressible alternation (level adaptation at e1-e4)
        out, tot = [], 0
        while tot < n:
            m = int(rng.integers(1000, 400_000))
            out.append(rng.integers(0, 256, m, dtype=np.uint8) if rng.random() < 0.5 else text[:m]); tot += m
        return np.concatenate(out)[:n]
    if kind == 7:                                   # copies of earlier stretches, 17..600 bytes long, some edited: long and
        x = text[:n].copy()                         # mid-length matches, lazy probes at long lengths (source-code-like)
        at = 64
        while at + 700 < n:
            m = int(rng.integers(17, 600))
            src = int(rng.integers(0, at - 16)) if rng.random() < 0.7 else max(0, at - int(rng.integers(1, 40)))
            m = min(m, at - src) if rng.random() < 0.5 else m
            for k in range(m):                      # (byte by byte: a source may overlap its copy)
                x[at + k] = x[src + k]
            if rng.random() < 0.3:
                x[at + int(rng.integers(0, m))] ^= 1
            at += m + int(rng.integers(0, 30))
        return x
    return rng.integers(0, 256, n, dtype=np.uint8)  # plain noise


@pytest.mark.parametrize("seed", range(6))
def test_differential_vs_oracle_and_roundtrip(oracle, seed):
    import libzling_amd as zl
    from oracle_py import textgen
    rng = np.random.Generator(np.random.PCG64(1000 + seed))
    text = textgen(3_000_000, 200 + seed)
    for it in range(8):
        kind = (seed + it) % 8
        n = int(rng.integers(1, 900_000)) if it else int(rng.integers(1, 600))
        x = np.ascontiguousarray(make_input(rng, kind, n, text))
        lv = int(rng.integers(0, 5))
        z = zl.encode(x, lv)
        ref = oracle.encode(x, lv)
        assert z.size == ref.size and np.array_equal(z, ref), (seed, it, kind, lv, x.size)
        with zl.Stream(0, 0, False, 1) as d:
            assert np.array_equal(d.decode(z, x.size), x), (seed, it, kind, lv)
