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
