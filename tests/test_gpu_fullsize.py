"""Full-size checks at BASELINE.json's configuration (10^9-byte stream, 60 blocks in flight)."""
import hashlib
import os
import subprocess
import time

import numpy as np
import pytest

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_enwik9_shape_e0_matches_oracle_blockwise(oracle, manifest):
    """The whole 10^9-byte e0 stream equals the CPU oracle's, compared block by block (a checksum per block,
    so a mismatch names the block) -- the same stream bench.py times -- and its SHA-256 and size are the REAL reference's
    (tests/golden/manifest.json "config3_enwik9_shape", pinned by make_golden.py --configs from oracle/_ref)."""
    import libzling_amd as zl
    from oracle_py import textgen
    n = 1_000_000_000
    x = textgen(n, 0)
    nb = (n + zl.BLOCK - 1) // zl.BLOCK
    with zl.Stream(0, 0, True, nb) as s:
        z = s.encode(x)
        ends = s.block_ends
    pin = manifest["config3_enwik9_shape"]
    assert hashlib.sha256(x.tobytes()).hexdigest() == pin["input_sha256"]
    assert z.size == pin["zlng_bytes"] and hashlib.sha256(z.tobytes()).hexdigest() == pin["sha256"]
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


def test_enwik8_shape_e0_serial_blocks_and_cli(tmp_path, manifest, capsys):
    """BASELINE configs 1 and 2 at their own size: exactly 100,000,000 bytes (five full blocks + 16,113,920 bytes) at e0,
    (i) through a context of ONE block, block by block ("serial 16 MB blocks": one block in flight, the literal tables carried by
    the context), (ii) through the drop-in CLI (tools/zling_demo e0, then d) -- the harness of benchmark/benchmark.sh:22-45.
    Both streams must be the reference's (manifest "config12_enwik8_shape", pinned from oracle/_ref)."""
    import libzling_amd as zl
    from libzling_amd import build
    from oracle_py import textgen
    build.build_all()
    pin = manifest["config12_enwik8_shape"]
    n = pin["bytes"]
    x = textgen(n, 0)
    assert hashlib.sha256(x.tobytes()).hexdigest() == pin["input_sha256"]
    parts = []
    with zl.Stream(0, 0, True, 1) as s:
        s.encode(x[: 1 << 20].copy())                 # start-up outside the clock (and a state to reset)
        with zl.Stream(0, 0, True, 1) as fresh:
            st0, lv0 = fresh.get_state()
        s.set_state(st0, lv0)
        t0 = time.perf_counter()
        for off in range(0, n, zl.BLOCK):
            parts.append(s.encode(x[off: off + zl.BLOCK]))
        dt = time.perf_counter() - t0
    z = np.concatenate(parts)
    assert z.size == pin["zlng_bytes"] and hashlib.sha256(z.tobytes()).hexdigest() == pin["sha256"]
    with capsys.disabled():
        print("\n[config 2] 100,000,000 B at e0, one 16 MiB block in flight, host to host: %.2f s = %.1f MB/s" % (dt, n / dt / 1e6))
    src, enc, dec = str(tmp_path / "enwik8.shape"), str(tmp_path / "o.zlng"), str(tmp_path / "o.bin")
    x.tofile(src)
    demo = os.environ.get("ZLNG_DEMO") or os.path.join(ROOT, "tools", "zling_demo")
    t0 = time.perf_counter(); subprocess.check_call([demo, "e0", src, enc], stderr=subprocess.DEVNULL); te = time.perf_counter() - t0
    zc = np.fromfile(enc, dtype=np.uint8)
    assert zc.size == pin["zlng_bytes"] and hashlib.sha256(zc.tobytes()).hexdigest() == pin["sha256"]
    t0 = time.perf_counter(); subprocess.check_call([demo, "d", enc, dec], stderr=subprocess.DEVNULL); td = time.perf_counter() - t0
    assert np.array_equal(np.fromfile(dec, dtype=np.uint8), x)
    with capsys.disabled():
        print("[config 1] zling_demo e0 %.2f s, d %.2f s (file to file, process start included), round trip PASS" % (te, td))


def test_config4_share_e4_512_blocks_through_four_contexts(manifest):
    """BASELINE config 4, one GPU's share: 8 GiB = 512 blocks of the synthetic stream at e4 through sharding.RangeEncoder
    (4 contexts of 128 blocks, all parsing at once; MTF tables + current_level handed context to context).  The first 512 MiB
    are compared byte for byte with the real reference when oracle/_ref is present (else the oracle); the whole stream's
    SHA-256 and size are the reference's own (tests/golden/manifest.json "config4_share": pinned from a full reference run)."""
    import torch
    import libzling_amd as zl
    from libzling_amd import sharding
    from oracle_py import Oracle, Reference, textgen
    pin = manifest["config4_share"]
    n, level = pin["bytes"], pin["level"]
    x = textgen(n, 0)
    nb = n // zl.BLOCK
    d_in = torch.empty(n + 512, dtype=torch.uint8, device="cuda")
    step = 1 << 30
    for o in range(0, n, step):
        d_in[o:o + step].copy_(torch.from_numpy(x[o:o + step]))
    d_in[n:].zero_()
    enc = sharding.RangeEncoder(lambda blocks: zl.Stream(0, level, True, blocks), nb, 128)
    assert len(enc.parts) == 4 and all(p == 128 for p in enc.parts)
    cap = zl.encode_bound(n) + 4 * len(enc.parts)
    d_out = torch.empty(cap, dtype=torch.uint8, device="cuda")
    d_state = torch.zeros(sharding.STATE_BUF, dtype=torch.uint8, device="cuda")
    st0, lv0 = enc.streams[0].get_state()
    d_state[:zl.MTF_STATE].copy_(torch.from_numpy(st0))
    torch.cuda.synchronize()
    enc.parse(d_in.data_ptr(), n)
    segs, _lv = enc.finish(d_out.data_ptr(), cap, d_state.data_ptr(), lv0)
    torch.cuda.synchronize()
    assert sum(k for _, k in segs) == pin["zlng_bytes"]
    h = hashlib.sha256()
    head = []
    for o, k in segs:
        part = d_out[o:o + k].cpu().numpy()
        h.update(part.tobytes())
        if sum(p.size for p in head) < (600 << 20):
            head.append(part)
    assert h.hexdigest() == pin["sha256"]
    # a whole-block prefix of the input encodes to a prefix of the stream (state only flows forward)
    sample = 512 << 20
    cpu = Reference() if Reference.available() else Oracle()
    z = cpu.encode(x[:sample], level)
    got = np.concatenate(head)[: z.size]
    assert got.size == z.size and np.array_equal(got, z)
    enc.close()
