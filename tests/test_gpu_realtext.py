"""Real text: source and documentation files that ship with this image (the same on the GPU box), concatenated in sorted
path order -- long matches, long runs of blanks, many distinct symbols per context: what the generator's text does not
have.  GPU vs oracle, byte for byte, at e0 and at e4, and a round trip through the GPU decoder."""
import os
import sys

import numpy as np
import pytest

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "scripts"))


@pytest.fixture(scope="module")
def real_text():
    from real_text_soak import gather
    try:
        x, nfiles = gather(48 << 20)
    except ValueError:                              # np.concatenate of nothing
        pytest.skip("no text files found in this image")
    if x.size < (20 << 20):
        pytest.skip("only %d bytes of text files in this image" % x.size)
    return x


@pytest.mark.parametrize("level", [0, 4])
def test_real_text_matches_oracle(oracle, real_text, level):
    import libzling_amd as zl
    nb = (real_text.size + zl.BLOCK - 1) // zl.BLOCK
    with zl.Stream(0, level, True, nb) as s:
        z = s.encode(real_text)
    ref = oracle.encode(real_text, level)
    assert z.size == ref.size and np.array_equal(z, ref), "first difference at %d" % int(np.argmax(z[:min(z.size, ref.size)] != ref[:min(z.size, ref.size)]))


def test_real_text_round_trip_on_the_gpu(real_text):
    import libzling_amd as zl
    x = real_text[: 20 << 20]
    z = zl.encode(x, 0)
    with zl.Stream(0, 0, False, 2) as d:
        assert np.array_equal(d.decode(z, x.size), x)


def test_long_matches_are_common_in_this_text(oracle, real_text):
    """What makes this input a different test from the generator's: most of its bytes sit in matches longer than the 16 bytes
    phase 1 compares (those lanes are left open and settled by the parser only when they are token starts)."""
    tok, _ = oracle.parse_block(real_text[: 1 << 24], 0)
    sym = (tok & 0xFFFF).astype(np.int64)
    ln = sym[sym >= 258] - 258 + 4
    assert ln[ln > 16].sum() > (1 << 24) // 4
