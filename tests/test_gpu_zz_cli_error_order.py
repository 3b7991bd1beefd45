"""The shim's two Decode paths meet errors in the reference's order (src/libzling.cpp:312-404), through tools/zling_demo.

This file sorts behind the other GPU tests on purpose: it was written after the GPU pool closed to this repository in round 5 and
has not run on a GPU yet (its expectations are the oracle's verdicts, tests/test_oracle_hostile.py; the C-ABI behaviour it relies on
is held to the oracle by tests/test_gpu_hostile.py, which has).  A first-run surprise here must not hide the suite under `-x`."""
import os
import subprocess

import pytest

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
DEMO = os.environ.get("ZLNG_DEMO") or os.path.join(ROOT, "tools", "zling_demo")


@pytest.fixture(scope="module", autouse=True)
def built():
    from libzling_amd import build
    build.build_all()
    assert os.path.exists(DEMO)


@pytest.mark.parametrize("readahead", ["0", "1"])
def test_cli_meets_decode_errors_in_stream_order(tmp_path, oracle, readahead):
    """The reference's Decode is one loop (src/libzling.cpp:312-404): the Huffman stream of sub-block k is decoded before the flag
    of sub-block k + 1 is read, so a damaged table in front of a damaged flag is "bad code1", and the flag's own message only
    appears when nothing in front of it fails.  Both shim paths: block by block (a handler is installed: zling_demo's progress
    handler) and the batched read-ahead path."""
    import corpus
    import hostile
    x = corpus.get("text_700k")                                      # one block, two sub-blocks
    z = oracle.encode(x, 0)
    subs, _ends = hostile.walk(z)
    assert len(subs) == 2
    f1 = subs[1][0]

    def run(m):
        bad = tmp_path / "bad.zlng"
        bad.write_bytes(m.tobytes())
        r = subprocess.run([DEMO, "d", str(bad), str(tmp_path / "x")], stderr=subprocess.PIPE, env=dict(os.environ, ZLNG_DECODE_READAHEAD=readahead))
        assert r.returncode != 0
        return r.stderr
    m = z.copy(); m[f1] = 7
    assert b"invalid encflag" in run(m)
    m[subs[0][1]: subs[0][1] + 257] = 0                              # ... and no code for any symbol in the sub-block in front of it
    assert b"bad code1" in run(m)
    m = z.copy(); m[f1 + 5: f1 + 9] = [0, 0x10, 0, 0]                 # rlen of the second sub-block over the limit
    assert b"invalid block size" in run(m)
    m[subs[0][0] + 1: subs[0][0] + 5] = [0, 0, 0, 9]                  # ... behind a first sub-block whose lengths miss its encpos
    assert b"lzdecode failed" in run(m)
