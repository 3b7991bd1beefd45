"""The drop-in C++ API (libzling_amd.so over the C-ABI) exercised through tools/zling_demo, the
counterpart of the reference's demo/zling.cpp used by its own fuzz test (test/fuzzy/libzling_fuzzy.py:20-42)."""
import os
import subprocess

import numpy as np
import pytest

import corpus

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
DEMO = os.environ.get("ZLNG_DEMO") or os.path.join(ROOT, "tools", "zling_demo")      # ZLNG_DEMO: scripts/sanitize.sh host (the ASan build)
G = os.path.join(ROOT, "tests", "golden")


@pytest.fixture(scope="module", autouse=True)
def built():
    from libzling_amd import build
    build.build_all()
    assert os.path.exists(DEMO)


@pytest.mark.parametrize("name,level", [("text_64k", 0), ("text_64k", 4), ("runs_ab", 2), ("text_1", 0), ("text_0", 0)])
def test_cli_encode_matches_golden_and_roundtrips(tmp_path, name, level):
    src = os.path.join(G, name + ".bin")
    want = np.fromfile(os.path.join(G, "%s.e%d.zlng" % (name, level)), dtype=np.uint8)
    enc, dec = str(tmp_path / "o.zlng"), str(tmp_path / "o.bin")
    subprocess.check_call([DEMO, "e%d" % level, src, enc])
    assert np.array_equal(np.fromfile(enc, dtype=np.uint8), want)
    subprocess.check_call([DEMO, "d", enc, dec])
    assert np.array_equal(np.fromfile(dec, dtype=np.uint8), np.fromfile(src, dtype=np.uint8))


def test_cli_streams_through_pipes_multiblock(tmp_path, oracle):
    x = corpus.get("carry_2blk")
    p = subprocess.run([DEMO, "e0"], input=x.tobytes(), stdout=subprocess.PIPE, check=True)
    z = np.frombuffer(p.stdout, dtype=np.uint8)
    assert np.array_equal(z, oracle.encode(x, 0))
    q = subprocess.run([DEMO, "d"], input=p.stdout, stdout=subprocess.PIPE, check=True)
    assert q.stdout == x.tobytes()


def test_cli_rejects_corrupt_stream(tmp_path):
    bad = tmp_path / "bad.zlng"
    bad.write_bytes(bytes([9, 1, 2, 3]))
    r = subprocess.run([DEMO, "d", str(bad), str(tmp_path / "x")], stderr=subprocess.PIPE)
    assert r.returncode != 0 and b"invalid encflag" in r.stderr


def test_stream_state_is_carried_across_calls(oracle):
    """The shim feeds a long stream to the GPU in several calls (ZLNG_BATCH_BLOCKS); MTF tables and level state must
    persist in the context between them (src/libzling.cpp:185,197: the reference reuses one encoder object)."""
    x = corpus.get("carry_2blk")
    want = oracle.encode(x, 0)
    env = dict(os.environ, ZLNG_BATCH_BLOCKS="1")
    p = subprocess.run([DEMO, "e0"], input=x.tobytes(), stdout=subprocess.PIPE, check=True, env=env)
    assert np.array_equal(np.frombuffer(p.stdout, dtype=np.uint8), want)
    q = subprocess.run([DEMO, "d"], input=p.stdout, stdout=subprocess.PIPE, check=True, env=env)
    assert q.stdout == x.tobytes()


def test_python_stream_called_block_by_block(oracle):
    import libzling_amd as zl
    from oracle_py import textgen
    x = textgen(2 * zl.BLOCK + 500_000, 91)
    want = oracle.encode(x, 4)
    parts = []
    with zl.Stream(0, 4, True, 1) as s:
        for off in range(0, x.size, zl.BLOCK):
            parts.append(s.encode(x[off:off + zl.BLOCK]))
    assert np.array_equal(np.concatenate(parts), want)


@pytest.mark.parametrize("level,batch,pipeline", [(0, 1, "1"), (0, 2, "1"), (4, 1, "1"), (2, 2, "1"), (4, 1, "0")])
def test_shim_pipeline_over_two_contexts(oracle, level, batch, pipeline):
    """batches alternate between two contexts, the parse of batch k+1 is queued before batch k is
    finished and the MTF tables + current_level travel between the contexts through host memory.  Five blocks with a
    random (incompressible) stretch in the middle, so at e2/e4 the level adaptation (src/libzling.cpp:261-266) flips
    across a batch boundary: the finishing context must notice that its parse was speculated at the wrong entry level."""
    import libzling_amd as zl
    from oracle_py import textgen
    rng = np.random.default_rng(77)
    x = np.concatenate([textgen(zl.BLOCK + 300_000, 5), rng.integers(0, 256, zl.BLOCK, dtype=np.uint8),
                        textgen(2 * zl.BLOCK + 123_457, 6)])
    want = oracle.encode(x, level)
    env = dict(os.environ, ZLNG_BATCH_BLOCKS=str(batch), ZLNG_PIPELINE=pipeline)
    p = subprocess.run([DEMO, "e%d" % level], input=x.tobytes(), stdout=subprocess.PIPE, check=True, env=env)
    assert np.array_equal(np.frombuffer(p.stdout, dtype=np.uint8), want)


def test_split_host_api_matches_one_call(oracle):
    """zlng_encode_parse / zlng_encode_finish (host buffers) on two contexts by hand == one zlng_encode_blocks call."""
    import ctypes as C
    import libzling_amd as zl
    from oracle_py import textgen
    L = zl.lib()
    L.zlng_encode_parse.argtypes = [C.c_void_p, C.c_void_p, C.c_size_t]
    L.zlng_encode_finish.argtypes = [C.c_void_p, C.c_void_p, C.c_size_t, C.POINTER(C.c_size_t), C.POINTER(C.c_size_t)]
    x = textgen(3 * zl.BLOCK + 1000, 12)
    want = oracle.encode(x, 0)
    cuts = [0, 2 * zl.BLOCK, x.size]
    with zl.Stream(0, 0, True, 2) as a, zl.Stream(0, 0, True, 2) as b:
        parts = []
        ctxs = [a, b]
        for i in range(2):                       # both parses are queued before anything is finished
            seg = np.ascontiguousarray(x[cuts[i]:cuts[i + 1]])
            assert L.zlng_encode_parse(ctxs[i]._h, seg.ctypes.data, seg.size) == 0
            ctxs[i]._seg = seg
        st = None
        for i in range(2):
            if st is not None:
                ctxs[i].set_state(*st)
            out = np.empty(zl.encode_bound(ctxs[i]._seg.size), dtype=np.uint8)
            n = C.c_size_t(0)
            ends = (C.c_size_t * 2)()
            assert L.zlng_encode_finish(ctxs[i]._h, out.ctypes.data, out.size, C.byref(n), ends) == 0
            parts.append(out[:n.value].copy())
            st = ctxs[i].get_state()
    assert np.array_equal(np.concatenate(parts), want)


@pytest.mark.parametrize("level,devices,batch", [(0, "0,0", 2), (4, "0,0", 1), (4, "0,0,0", 2)])
def test_shim_spreads_one_stream_over_devices(oracle, level, devices, batch):
    """ZLNG_DEVICES: baidu::zling::Encode hands every batch to a zlng_group -- one context per listed device, contiguous
    block ranges, MTF tables + current_level handed member to member (src/libzling.cpp:185, 197, 261-266).  Two or three
    contexts on device 0 stand in for as many GPUs; the incompressible stretch crosses member and batch boundaries."""
    import libzling_amd as zl
    from oracle_py import textgen
    rng = np.random.default_rng(78)
    x = np.concatenate([textgen(2 * zl.BLOCK - 400_000, 7), rng.integers(0, 256, 900_000, dtype=np.uint8),
                        textgen(3 * zl.BLOCK + 55_555, 8)])
    want = oracle.encode(x, level)
    env = dict(os.environ, ZLNG_DEVICES=devices, ZLNG_BATCH_BLOCKS=str(batch))
    p = subprocess.run([DEMO, "e%d" % level], input=x.tobytes(), stdout=subprocess.PIPE, check=True, env=env)
    assert np.array_equal(np.frombuffer(p.stdout, dtype=np.uint8), want)


def test_batched_decode_moves_to_the_full_size_context_and_keeps_the_tables(tmp_path, oracle):
    """Decode() without handler-side reads (no handler, a DecodeReadAhead handler, or ZLNG_DECODE_READAHEAD=1) starts on a 4-block
    context and rebuilds a larger one once the stream has produced 4 blocks (four times the blocks, but no more than are left once the
    input has ended), carrying the literal tables over (zlng_get_state / zlng_set_state).  Ten blocks of compressible data -- every
    one arrives inside the first 8 MiB chunk, so six are left behind the first call: a 6-block context -- cross that rebuild; the
    literals behind it only decode if the tables came along (src/libzling_lz.cpp:378-386: the decoder's Reset keeps m_mtf)."""
    n = 9 * corpus.BLOCK + 300_000
    unit = corpus.get("text_64k")
    x = np.concatenate([unit] * (n // unit.size + 1))[:n].copy()
    x[5 * corpus.BLOCK + 1000: 5 * corpus.BLOCK + 201000] = corpus.get("text_700k")[:200000]     # fresh literals behind the rebuild
    z = oracle.encode(x, 0)
    assert z.size < (8 << 20)
    src = tmp_path / "in.zlng"
    z.tofile(str(src))
    for ra in ("1", "0"):
        out = tmp_path / ("out%s.bin" % ra)
        subprocess.check_call([DEMO, "d", str(src), str(out)], env=dict(os.environ, ZLNG_DECODE_READAHEAD=ra))
        assert np.array_equal(np.fromfile(str(out), dtype=np.uint8), x)
