"""Deterministic named inputs shared by the golden generator and the parity tests.

Every generator is a pure function of its name; tests/golden/manifest.json pins the
SHA-256 of each input so a drifting generator is detected rather than silently re-baselined.
"""
import hashlib
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "oracle"))

from oracle_py import textgen  # noqa: E402  (synthetic text generator, libzling_amd/host/textgen.c)

BLOCK = 16777216


def sha(a):
    return hashlib.sha256(np.ascontiguousarray(a).tobytes()).hexdigest()


def _rng(seed):
    return np.random.Generator(np.random.PCG64(seed))


def _text(n, chunk=0):
    return textgen(n, chunk)


def _skew(n, nsym, seed):
    p = np.array([2.0 ** -i for i in range(nsym)])
    p /= p.sum()
    return _rng(seed).choice(nsym, n, p=p).astype(np.uint8)


def _carry():
    # block 1 = text + "abc" filler up to exactly 16 MiB, block 2 = text.
    head = _text(200_000, 7)
    fill = np.frombuffer((b"abc" * ((BLOCK - head.size) // 3 + 1))[: BLOCK - head.size], dtype=np.uint8)
    tail = _text(60_000, 8)
    return np.concatenate([head, fill, tail])


SMALL = {  # committed as input + .zlng at every level
    **{"text_%d" % n: (lambda n=n: _text(n)) for n in (0, 1, 2, 3, 4, 5, 274, 275, 276, 277, 278, 279, 280, 1000)},
    "text_64k": lambda: _text(65536, 3),
    "rand_4k": lambda: _rng(11).integers(0, 256, 4096, dtype=np.uint8),
    "zeros_20k": lambda: np.zeros(20000, np.uint8),
    "abc_30k": lambda: np.frombuffer(b"abc" * 10000, dtype=np.uint8),
    "runs_ab": lambda: np.frombuffer((b"a" * 700 + b"b" * 900 + b"ab" * 400) * 8, dtype=np.uint8),
    "bytes_ff": lambda: np.full(5000, 255, np.uint8),
    "skew_24k": lambda: _skew(24000, 40, 5),
}

LARGE = {  # pinned by SHA-256 of the reference's .zlng (and sub-block cut lists)
    "text_700k": lambda: _text(700_000, 1),
    "rand_1m": lambda: _rng(12).integers(0, 256, 1 << 20, dtype=np.uint8),
    "zeros_1m": lambda: np.zeros(1 << 20, np.uint8),
    "abc_1m": lambda: np.frombuffer((b"abc" * 350000)[: 1 << 20], dtype=np.uint8),
    "skew_400k": lambda: _skew(400_000, 40, 6),
    "skew2_600k": lambda: np.concatenate([_skew(300_000, 3, 8), _text(300_000, 2)]),
    "mixed_e4": lambda: np.concatenate([_rng(13).integers(0, 256, 300_000, dtype=np.uint8), _text(500_000, 4)]),
    "carry_2blk": _carry,
    "text_33m": lambda: _text(2 * BLOCK + 1_000_000, 20),
}

ALL = {**SMALL, **LARGE}


def get(name):
    return np.ascontiguousarray(ALL[name](), dtype=np.uint8)


def corrupt_cases(oracle):
    """Deterministic streams that hit each Huffman validity check of the decoder exactly (src/libzling.cpp:381, 391, 398):
    [(name, bytes, oracle error code)].  Built from the one-sub-block stream of text_64k:
      * code1  -- all 257 nibble bytes of length table 1 zeroed: no code exists, the first 15-bit lookup misses;
      * code2  -- the 16 nibble bytes of length table 2 zeroed: the first match's index code misses;
      * lz     -- rlen cut so that the last counted u16 entry is a match symbol: the reference keeps the index entry behind it (it is
                  stored at tbuf[rlen], src/libzling.cpp:398, and the replay reads it there), so the sub-block decodes short of its
                  encpos: "lzdecode failed".  (The third check, "bad ex-bits", :399, cannot fire at all: the 32 index codes name at
                  most 3840 + 255.  tests/test_oracle_hostile.py holds this against the real reference.)"""
    x = get("text_64k")
    z = oracle.encode(x, 0)
    pay = 13                                           # flag + encpos + rlen + olen
    c1 = z.copy(); c1[pay:pay + 257] = 0
    c2 = z.copy(); c2[pay + 257:pay + 273] = 0
    tok, _ = oracle.parse_block(x, 0)
    first_match = int(np.argmax((tok & 0xFFFF) >= 258))
    assert (tok[first_match] & 0xFFFF) >= 258 and not ((tok[:first_match] & 0xFFFF) >= 258).any()
    ex = z.copy()
    ex[5:9] = list(int(first_match + 1).to_bytes(4, "big"))       # u16 index of the first match == its token index
    return x, [("code1", c1, -4), ("code2", c2, -5), ("lz", ex, -7)]
