"""The callback protocol of the drop-in C++ API, driven by a caller-written Inputter / Outputter / ActionHandler
(tests/cxx/protocol_test.cpp): short reads and writes, callback order, error returns, and a handler that writes to / reads from
the streams inside OnProcess -- what src/libzling.cpp:174-291 and :293-427 make observable."""
import json
import os
import struct
import subprocess

import numpy as np
import pytest

import corpus

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
BIN = os.environ.get("ZLNG_PROTOCOL_TEST") or os.path.join(ROOT, "tests", "cxx", "protocol_test")      # ZLNG_PROTOCOL_TEST: scripts/sanitize.sh host (the ASan build)


def block_ends(z, trailer=0):
    """End offset of every block of a .zlng stream (Appendix A of SURVEY.md: 01 encpos rlen olen payload ... 00), with `trailer`
    extra bytes behind each block's terminator (the Adler32 variant writes four)."""
    ends, pos, n = [], 0, len(z)
    while pos < n:
        while z[pos] == 1:
            olen = struct.unpack(">I", bytes(z[pos + 9: pos + 13]))[0]
            pos += 13 + olen
        assert z[pos] == 0
        pos += 1 + trailer
        ends.append(pos)
    return ends


@pytest.fixture(scope="module")
def run(tmp_path_factory, oracle):
    from libzling_amd import build
    build.build_all()
    d = tmp_path_factory.mktemp("protocol")
    x = corpus.get("text_33m")                       # two full 16 MiB blocks + 1,000,000 bytes
    src, prefix = str(d / "in.bin"), str(d / "out")
    x.tofile(src)
    p = subprocess.run([BIN, "0", src, prefix], stdout=subprocess.PIPE, check=True, timeout=900)
    log = json.loads(p.stdout.decode())
    return x, oracle.encode(x, 0), prefix, log


def kinds(ev):
    return [e[0] for e in ev]


def test_encode_short_io_is_bit_exact_and_ordered(run):
    x, want, prefix, log = run
    z = np.fromfile(prefix + ".zlng", dtype=np.uint8)
    assert np.array_equal(z, want)
    e = log["encode"]
    assert e["rc"] == 0
    ev = e["events"]
    assert kinds(ev) == [0, 1, 1, 1, 2]                          # OnInit, one OnProcess per block, OnDone
    assert all(v[5] == 1 for v in ev)                            # every callback on the caller's thread
    assert ev[0][2] == 0 and ev[0][3] == 0                       # OnInit before any I/O (src/libzling.cpp:175-178)
    ends = block_ends(z)
    sizes = [corpus.BLOCK, corpus.BLOCK, 1_000_000]
    off = 0
    for k, v in enumerate(ev[1:4]):
        assert v[1] == sizes[k]                                  # OnProcess(raw block, its size) ...
        assert v[4] == _adler(x[off: off + sizes[k]])            # ... with that block's bytes
        assert v[3] == ends[k]                                   # all of block k's bytes pushed, none of block k + 1 (:269-283)
        off += sizes[k]
    assert ev[4][3] == z.size


def test_decode_short_io_reads_nothing_ahead_of_a_handler(run):
    x, want, prefix, log = run
    y = np.fromfile(prefix + ".dec", dtype=np.uint8)
    assert np.array_equal(y, x)
    d = log["decode"]
    assert d["rc"] == 0 and d["threw"] == ""
    ev = d["events"]
    assert kinds(ev) == [0, 1, 1, 1, 2] and all(v[5] == 1 for v in ev)
    ends = block_ends(want)
    out = 0
    for k, v in enumerate(ev[1:4]):
        out += v[1]
        assert v[2] == ends[k]          # the inputter stands exactly behind block k's terminator (src/libzling.cpp:306-336)
        assert v[3] == out              # block k's bytes are out before OnProcess(k) (:412-419)
    assert out == x.size


def test_adler32_handler_pair_round_trips(run):
    """demo/zling.cpp:124-132 with ENABLE_ADLER32_CHECKSUM: the writing side appends 4 bytes behind every block, the reading side
    pulls them from the inputter inside OnProcess -- a stream this Encode wrote must decode with the matching handler."""
    x, want, prefix, log = run
    za = np.fromfile(prefix + ".adler.zlng", dtype=np.uint8)
    ends, ends_a = block_ends(want), block_ends(za, trailer=4)
    assert [e + 4 * (k + 1) for k, e in enumerate(ends)] == ends_a
    prev = prev_a = 0
    for k in range(3):                                            # same block bytes, checksum behind each
        assert np.array_equal(za[prev_a: ends_a[k] - 4], want[prev: ends[k]])
        prev, prev_a = ends[k], ends_a[k]
    ev = log["encode_adler"]["events"]
    assert [struct.unpack(">I", bytes(za[e - 4: e]))[0] for e in ends_a] == [v[4] for v in ev[1:4]]
    d = log["decode_adler"]
    assert d["rc"] == 0 and d["threw"] == ""
    assert np.array_equal(np.fromfile(prefix + ".adler.dec", dtype=np.uint8), x)
    bad = log["decode_adler_damaged"]
    assert "adler32 checksum not match" in bad["threw"]
    assert kinds(bad["events"]) == [0, 1, 1, 1]                   # the handler's exception leaves Decode (no OnDone, like the reference)


@pytest.mark.parametrize("name", ["encode_output_error", "encode_input_error", "decode_output_error", "decode_input_error"])
def test_stream_error_in_the_middle_returns_minus_one_and_fires_ondone(run, name):
    r = run[3][name]
    assert r["rc"] == -1 and r["threw"] == ""
    k = kinds(r["events"])
    assert k[0] == 0 and k[-1] == 2 and k.count(2) == 1 and set(k[1:-1]) <= {1}


def test_decode_without_handler_batched_path(run):
    r = run[3]["decode_no_handler"]
    assert r["rc"] == 0 and r["same_as_input"] is True


def test_per_call_traits_of_the_handler(run):
    """libzling.h's two extension tags: EncodePlacement (this call over two contexts of device 0, the four longest rank chains on
    host threads: same bytes) and DecodeReadAhead (this call may read ahead: the batched path, same output, and at the first
    OnProcess the inputter already stands behind more than block 0 -- the whole 3-block stream fits the first 8 MiB chunk)."""
    x, want, prefix, log = run
    e = log["encode_placement"]
    assert e["rc"] == 0 and e["threw"] == "" and e["same_bytes"] is True
    d = log["decode_read_ahead_trait"]
    assert d["rc"] == 0 and d["threw"] == "" and d["same_as_input"] is True and d["calls"] == 3
    assert d["inputter_pos_at_first_onprocess"] > block_ends(want)[0]


def _adler(a):
    import zlib
    return zlib.adler32(a.tobytes()) & 0xFFFFFFFF
