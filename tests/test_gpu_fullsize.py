"""Full-size checks at BASELINE.json's configuration (10^9-byte stream, 60 blocks in flight)."""
import hashlib

import numpy as np
import pytest

pytestmark = pytest.mark.gpu


def test_enwik9_shape_e0_matches_oracle_blockwise(oracle):
    """The whole 10^9-byte e0 stream equals the CPU oracle's, compared block by block (a checksum per block,
    so a mismatch names the block) -- the same stream bench.py times."""
    import libzling_amd as zl
    from oracle_py import textgen
    n = 1_000_000_000
    x = textgen(n, 0)
    nb = (n + zl.BLOCK - 1) // zl.BLOCK
    with zl.Stream(0, 0, True, nb) as s:
        z = s.encode(x)
        ends = s.block_ends
    ref = oracle.encode(x, 0)
    assert z.size == ref.size
    prev = 0
    for b, e in enumerate(ends):
        assert hashlib.sha256(z[prev:e].tobytes()).digest() == hashlib.sha256(ref[prev:e].tobytes()).digest(), "block %d" % b
        prev = e
    assert prev == z.size
    # structural property, independent of the oracle: the frame walks exactly to the end, every block
    # closes with 0x00 and the sub-block sizes add up
    p, blocks, covered = 0, 0, 0
    while p < z.size:
        if z[p] == 0:
            blocks += 1; p += 1; covered += last; continue
        enc, rl, ol = (int.from_bytes(z[p + 1 + 4 * k: p + 5 + 4 * k].tobytes(), "big") for k in range(3))
        assert rl <= 262144 and 273 <= ol <= 393216
        last = enc
        p += 13 + ol
    assert p == z.size and blocks == nb and covered == n
