"""Host-side logic of bench.py that needs no GPU: the identity of the kernel sources a PMC profile belongs to and the rule
that a committed traffic figure is quoted only for those very sources and that very workload."""
import json
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import bench


def test_kernel_source_sha_is_a_stable_hash_of_the_encode_kernels():
    a, b = bench.kernel_source_sha(), bench.kernel_source_sha()
    assert a == b and len(a) == 16 and int(a, 16) >= 0
    for f in bench.ENCODE_KERNEL_SOURCES:
        assert os.path.exists(os.path.join(ROOT, "libzling_amd", "csrc", f)), f


def test_traffic_is_quoted_only_for_the_same_sources_and_workload(tmp_path, monkeypatch):
    prof = tmp_path / "profiles"
    prof.mkdir()
    entry = {"kernel_source_sha": "0" * 16, "workload": {"bytes": 1000, "level": 0, "blocks": 1},
             "kernels": {"k_rolz_parse_wave": {"hbm_bytes_corrected": 4242}}}
    (prof / "r09_z_pmc_traffic.json").write_text(json.dumps(entry))
    monkeypatch.setattr(bench, "ROOT", str(tmp_path))
    monkeypatch.setattr(bench, "kernel_source_sha", lambda: "f" * 16)
    t, src = bench.traffic_of("k_rolz_parse_wave", 1000, 0, "synthetic", 1)
    assert t is None and src.startswith("stale")                      # other kernel sources
    monkeypatch.setattr(bench, "kernel_source_sha", lambda: "0" * 16)
    t, src = bench.traffic_of("k_rolz_parse_wave", 1000, 0, "synthetic", 1)
    assert t == 4242 and "r09_z_pmc_traffic.json" in src
    assert bench.traffic_of("k_rolz_parse_wave", 2000, 0, "synthetic", 1) == (None, None)     # another size
    assert bench.traffic_of("k_rolz_parse_wave", 1000, 4, "synthetic", 1) == (None, None)     # another level
    assert bench.traffic_of("k_rolz_parse_wave", 1000, 0, "synthetic", 2) == (None, None)     # several ranks


def test_the_committed_profile_names_its_kernel_sources():
    """Every committed PMC profile carries the hash of the sources it was taken on; a stale newest one is reported (bench.py
    then prints `traffic: null` with the reason) but is not an error of the code."""
    import warnings
    files = sorted(f for f in os.listdir(os.path.join(ROOT, "profiles")) if f.endswith("_pmc_traffic.json"))
    assert files
    newest = json.load(open(os.path.join(ROOT, "profiles", files[-1])))
    assert len(newest["kernel_source_sha"]) == 16
    if newest["kernel_source_sha"] != bench.kernel_source_sha():
        warnings.warn("profiles/%s was taken on other kernel sources: re-run scripts/profile_round.sh" % files[-1])


def test_schedule_model_arithmetic():
    """sharding.schedule_model / schedule_model_ranks (what bench.py prints as amdahl.model_ms) on hand-made stage times."""
    from libzling_amd import sharding
    # one context: parse, then rank + Huffman
    assert sharding.schedule_model([(500.0, 600.0, 2.0)]) == 1102.0
    # four equal contexts, all parses at once (rounds 1-3): one parse, then four finishes in a row
    st = [(3000.0, 1300.0, 3.0)] * 4
    assert sharding.schedule_model(st, 0) == 3000.0 + 4 * 1303.0
    # two parses in flight: parse ends 2000, 2000, 4000, 4000; finishes 3303, 4606, 5909, 7212
    st = [(2000.0, 1300.0, 3.0)] * 4
    assert sharding.schedule_model(st, 2) == 7212.0
    # a slow third parse makes its finish wait for it: parse ends 1000, 1000, 6000, 2000(+1000 start) ...
    st = [(1000.0, 100.0, 0.0), (1000.0, 100.0, 0.0), (5000.0, 100.0, 0.0), (1000.0, 100.0, 0.0)]
    assert sharding.schedule_model(st, 2) == 6000.0 + 100.0 + 100.0          # ctx 3 parsed long before, waits for ctx 2's finish
    # two ranks, one context each: both parse at once, rank 1's finish follows rank 0's
    assert sharding.schedule_model_ranks([[(600.0, 650.0, 2.0)], [(600.0, 650.0, 2.0)]], 2) == 600.0 + 2 * 652.0
    # ... and a rank whose parse is late holds its own finish up, not the earlier rank's
    assert sharding.schedule_model_ranks([[(100.0, 50.0, 0.0)], [(900.0, 50.0, 0.0)]], 2) == 950.0


def test_schedule_model_with_time_staggered_parses():
    """stagger = (first, gap_ms): context k is queued (k - first + 1) * gap_ms into the step."""
    from libzling_amd import sharding
    st = [(1900.0, 1300.0, 0.0), (2000.0, 1300.0, 0.0), (2000.0, 1300.0, 0.0), (2000.0, 1300.0, 0.0)]
    # parses end 1900, 1200+2000 = 3200, 4400, 5600; finishes 3200, 4500, 5800, 7100
    assert sharding.schedule_model(st, 2, (1, 1200.0)) == 7100.0
    # a gap shorter than the chain's pace changes nothing once the chain is the bottleneck ...
    assert sharding.schedule_model(st, 2, (1, 1000.0)) == 7100.0
    # ... a longer one makes the last finish wait for its parse: 3 * 1500 + 2000 + 1300
    assert sharding.schedule_model(st, 2, (1, 1500.0)) == 7800.0
    # first = 2: two contexts at once, then one per gap
    assert sharding.schedule_model(st, 0, (2, 1200.0)) == max(2000.0, 1900.0 + 1300.0) + 1300.0 + 1300.0 + 1300.0


def test_schedule_model_with_explicit_launch_times():
    from libzling_amd import sharding
    st = [(1500.0, 600.0, 0.0), (2000.0, 1300.0, 0.0), (2000.0, 1300.0, 0.0)]
    # parses end 1500, 300 + 2000, 1300 + 2000; finishes 2100, 3600, 4900
    assert sharding.schedule_model(st, 2, ("at", [0.0, 300.0, 1300.0])) == 4900.0


def test_block_ends_and_expected_ranges_slice_the_cpu_stream_at_block_ends(oracle, monkeypatch):
    """bench.py's PASS/FAIL column at N > 1: the bytes a rank must produce are the slice of the whole stream's .zlng between
    the block ends that bound its range (here with 64 KiB "blocks" so that the CPU suite stays fast)."""
    import hashlib
    import numpy as np
    from oracle_py import textgen
    x = textgen(200_000, 5)
    z = oracle.encode(x, 0)
    ends = bench.zlng_block_ends(z)
    assert ends == [z.size]                                         # one block: one terminator, at the very end
    z2 = np.concatenate([z, z])                                     # two framed blocks back to back walk as two
    assert bench.zlng_block_ends(z2) == [z.size, 2 * z.size]

    class A:                                                        # the argparse fields expected_ranges reads
        level, strong, size, no_cpu_baseline = 0, False, 200_000, False
    monkeypatch.setattr(bench, "load_input", lambda n, c: (x[:n], "synthetic"))
    monkeypatch.setattr(bench, "cpu_encoder", lambda: (oracle, "port"))
    want, src = bench.expected_ranges(A, 1, False, "synthetic", [(0, 200_000)], 1 << 30)
    assert want == [(int(z.size), hashlib.sha256(z.tobytes()).hexdigest())] and src.startswith("port-live")
    assert bench.expected_ranges(A, 1, False, "synthetic", [(0, 200_000)], 1000)[0] is None        # above the live limit
    A.no_cpu_baseline = True
    assert bench.expected_ranges(A, 1, False, "synthetic", [(0, 200_000)], 1 << 30)[0] is None


def test_pinned_ranges_cover_the_driver_scaling_runs(manifest):
    """tests/golden/manifest.json `sharded_ranges`: the per-rank size + SHA-256 of bench.py --gpus N for N = 1, 2, 4, 8 (weak: N x
    10^9 bytes; strong: 10^9 bytes), from the REAL reference; the ranges are sharding.plan's and the slices add up."""
    from libzling_amd import sharding
    pins = manifest["sharded_ranges"]
    per = pins["per_gpu_bytes"]
    assert per == 1_000_000_000 and pins["level"] == 0
    for w in (1, 2, 3, 4, 8):
        e = pins["weak"][str(w)]
        assert [(r["offset"], r["bytes"]) for r in e["ranks"]] == sharding.plan(per * w, w, per_rank_bytes=per)
        assert sum(r["zlng_bytes"] for r in e["ranks"]) == e["zlng_bytes"] and len(e["sha256"]) == 64
    assert pins["weak"]["1"]["sha256"] == manifest["config3_enwik9_shape"]["sha256"]
    for w in (2, 3, 4, 8):
        e = pins["strong"][str(w)]
        assert [(r["offset"], r["bytes"]) for r in e["ranks"]] == sharding.plan(per, w)
        assert sum(r["zlng_bytes"] for r in e["ranks"]) == manifest["config3_enwik9_shape"]["zlng_bytes"]
    # the weak streams are prefixes of one another: rank 0's range (60 whole blocks) has the same bytes at every N > 1
    assert len({pins["weak"][str(w)]["ranks"][0]["sha256"] for w in (2, 3, 4, 8)}) == 1


def test_expected_ranges_uses_the_pins_for_the_driver_configuration(manifest):
    class A:
        level, strong, size, no_cpu_baseline = 0, False, 1_000_000_000, True
    from libzling_amd import sharding
    for w in (1, 2, 4, 8):
        rg = sharding.plan(A.size * w, w, per_rank_bytes=A.size) if w > 1 else [(0, A.size)]
        want, src = bench.expected_ranges(A, w, w > 1, "synthetic", rg, 0)
        assert src.startswith("pins") and len(want) == w
        assert want == [(r["zlng_bytes"], r["sha256"]) for r in manifest["sharded_ranges"]["weak"][str(w)]["ranks"]]
    A.strong = True
    want, src = bench.expected_ranges(A, 4, True, "synthetic", sharding.plan(A.size, 4), 0)
    assert src.startswith("pins") and sum(k for k, _ in want) == manifest["config3_enwik9_shape"]["zlng_bytes"]
    assert bench.expected_ranges(A, 4, True, "enwik9", sharding.plan(A.size, 4), 0)[0] is None      # a real file has no pins


def test_bench_makes_its_workload_without_the_checker():
    """bench.py binds the text generator from libzling_amd itself; oracle/ is imported only by the legs that
    report the checker (cpu_baseline*, rank_chain's host column, live parity).  A fresh interpreter whose import system refuses
    anything from oracle/ imports bench, makes the workload and resolves the pinned parity column."""
    import subprocess
    code = r'''
import sys, os, importlib.abc
ROOT = %r
class Refuse(importlib.abc.MetaPathFinder):
    def find_spec(self, name, path=None, target=None):
        if name in ("oracle_py", "oracle"):
            raise ImportError("the checker is not available in this process: " + name)
sys.meta_path.insert(0, Refuse())
sys.path.insert(0, ROOT)
import bench
assert not any(os.path.basename(p.rstrip("/")) == "oracle" for p in sys.path), sys.path
x, src = bench.load_input(1 << 20, 0)
assert src == "synthetic" and x.size == 1 << 20 and x[:64].tobytes().isascii()
y, _ = bench.load_input(1 << 16, 7)
assert y.size == 1 << 16 and not (x[: 1 << 16] == y).all()
class A: level = 0; size = 1_000_000_000; strong = False; no_cpu_baseline = True
want, why = bench.expected_ranges(A, 1, True, "synthetic", [(0, 1_000_000_000)], 0)
assert want and len(want[0][1]) == 64 and why.startswith("pins"), why
print("ok")
''' % ROOT
    r = subprocess.run([sys.executable, "-c", code], capture_output=True, text=True, timeout=300)
    assert r.returncode == 0 and r.stdout.strip().endswith("ok"), r.stderr[-2000:]


def test_gpu_multistream_host_logic_with_a_stand_in_stream(oracle, monkeypatch):
    """bench.gpu_multistream (K contexts driven by K host threads, per-stream SHA-256 against the pin) has never met a GPU when
    this was written; its host logic -- buffers, threads, state reset, timing fields, the parity column -- runs here against a
    stand-in for zl.Stream that encodes with the checker into the buffer it is handed.  The kernels are not involved."""
    import ctypes
    import hashlib

    import numpy as np
    import torch

    class FakeStream:
        made = []

        def __init__(self, device, level, is_encode, nblocks):
            self.level, self.state, self.closed = level, (np.zeros(65536, np.uint8), 0), False
            FakeStream.made.append(self)

        def get_state(self):
            return self.state

        def set_state(self, mtf, level):
            self.state = (mtf, level)

        def encode_device(self, d_in, n, d_out, cap):
            x = np.frombuffer((ctypes.c_uint8 * n).from_address(d_in), np.uint8)
            z = oracle.encode(x, self.level)
            assert z.size <= cap
            ctypes.memmove(d_out, z.ctypes.data, z.size)
            return int(z.size)

        def timings(self):
            return [("rolz_parse", 1.0), ("mtf_chain", 2.0)]

        def close(self):
            self.closed = True
    monkeypatch.setattr(bench.zl, "Stream", FakeStream)
    from libzling_amd.textgen import textgen
    x = textgen(300_000, 1)
    want = hashlib.sha256(oracle.encode(x, 0).tobytes()).hexdigest()
    g = bench.gpu_multistream(3, x, 0, 0, want, 10.0, device=torch.device("cpu"))
    assert g["streams"] == 3 and g["parity"] is True and g["zlng_sha256_per_stream"] == [want] * 3 and g["value"] > 0
    assert g["stage_ms_per_stream"] == {"rolz_parse": [1.0] * 3, "mtf_chain": [2.0] * 3} and g["speedup_over_one_stream"] > 0
    assert len(FakeStream.made) == 3 and all(s.closed for s in FakeStream.made)
    g = bench.gpu_multistream(2, x, 0, 0, "0" * 64, 10.0, device=torch.device("cpu"))
    assert g["parity"] is False
    assert bench.gpu_multistream(2, x, 0, 0, None, 0.0, device=torch.device("cpu"))["parity"] is None
