"""The HIP decoder against the oracle on hostile streams that reach its BIG structures (tests/hostile.py, big set): rings that have
wrapped, a full 16 MiB block, match sources beyond the replay kernel's 64 KiB LDS window and copies that straddle its wrap, several
sub-blocks at a generic level, crafted bodies of thousands of tokens, a block with more u16 entries than a block can hold
(k_frame_walk's `too_many`, the oracle's ZO_DEV_ENTRIES).  Same comparison as tests/test_gpu_hostile.py: verdict class, the bytes
reported in front of the error, the bytes on success, the context still sound afterwards.  The oracle's verdicts on the very same
mutants are held to the REAL reference on the CPU (tests/test_oracle_hostile.py::test_oracle_equals_the_reference_on_big_hostile_streams).

Written in round 6 while the GPU pool was closed to this repository: sorted behind the rest of the suite so that a first-run
surprise here cannot hide it under `-x`."""
import numpy as np
import pytest

import corpus
import hostile
from test_gpu_hostile import CODE, gpu_verdict

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def zl():
    import libzling_amd as zl
    assert zl.lib().zlng_device_count() >= 1, "no gfx950 device visible"
    return zl


def test_big_hostile_streams_same_verdict_same_bytes_as_the_oracle(zl, oracle):
    good_x = corpus.get("text_64k")
    good_z = oracle.encode(good_x, 0)
    bad, seen, n = [], set(), 0
    with zl.Stream(0, 0, False, 4) as s:
        init, _ = s.get_state()
        for i, (name, m, cap) in enumerate(hostile.big_mutants(oracle, 20260930, 60)):     # the first 60 of the CPU differential's 200
            rc, y, _flags = oracle.decode_ex(m, cap)
            code, got = gpu_verdict(zl, s, init, m, cap)
            seen.add(code)
            n += 1
            if code != CODE[rc] or got.size != y.size or not np.array_equal(got, y):
                bad.append((i, name, "oracle %d / %d B" % (rc, y.size), "gpu %d / %d B" % (code, got.size)))
        code, got = gpu_verdict(zl, s, init, good_z, good_x.size)
        assert code == 0 and np.array_equal(got, good_x), "context broken after the big mutants"
    assert not bad, (len(bad), bad[:12])
    assert n == 60 and seen >= {0, -15, -16}
