"""bench.py at N > 1 must be self-validating: no line without the PASS/FAIL column (the reference's harness never prints a
time without `cmp`, benchmark/benchmark.sh:29-39).  A one-GPU box cannot run RCCL with two ranks on one device, so these tests
drive the SAME control flow (sharding.run_handoff: parse at once, 64 KiB state hand-off rank to rank, finishes in stream order)
with all ranks on cuda:0 and the few collectives over gloo (ZLNG_BENCH_ONE_DEVICE=1), through torch.distributed.run exactly
as the driver launches the multi-GPU bench."""
import json
import os
import socket
import subprocess
import sys

import pytest

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _free_port():
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        return s.getsockname()[1]


def _run(nproc, extra, timeout=900):
    env = dict(os.environ, ZLNG_BENCH_ONE_DEVICE="1", MASTER_ADDR="127.0.0.1")
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", str(nproc), "--master-addr", "127.0.0.1",
           "--master-port", str(_free_port()), os.path.join(ROOT, "bench.py"), "--gpus", str(nproc)] + extra
    p = subprocess.run(cmd, cwd=ROOT, env=env, capture_output=True, text=True, timeout=timeout)
    assert p.returncode == 0, p.stdout[-2000:] + p.stderr[-4000:]
    lines = [ln for ln in p.stdout.splitlines() if ln.startswith("{")]
    assert len(lines) == 1, p.stdout[-2000:]                       # ONE JSON line, from rank 0
    return json.loads(lines[0])


def _check_line(d, nproc):
    assert d["n_gpus"] == nproc and len(d["zlng_per_rank"]) == nproc
    assert d["parity"] is True, d
    assert d["parity_ranges"]["ok"] is True and "live" in d["parity_ranges"]["source"]
    assert d["cpu_baseline"]["value"] > 0 and d["cpu_baseline"]["kind"] in ("reference", "port")
    assert "byte-for-byte: True" in d["cpu_baseline"]["sample"]
    r = d["roofline"]
    assert r["bound"] == "hbm" and r["achieved"] > 0 and 0 < r["frac"] < 1 and r["algorithmic_bytes"] > 0
    assert sum(q["zlng_bytes"] for q in d["zlng_per_rank"]) == d["config"]["zlng_bytes_total"]
    assert d["zlng_sha256_rank0"] == d["zlng_per_rank"][0]["sha256"]
    alt = d["alt_host_rank_chains"]
    assert alt["identical_bytes"] is True and alt["value"] > 0


def test_bench_two_ranks_every_line_carries_parity_roofline_and_cpu_baseline():
    """2 ranks x 12 blocks (one stream of 402,653,184 B): every rank's size + SHA-256 against the CPU encoder's slice for that
    rank's range, rank 0's prefix byte for byte, the hybrid's bytes equal to the all-device run's."""
    d = _run(2, ["--steps", "1", "--warmup", "1", "--size", "201326592", "--no-multistream"])
    _check_line(d, 2)
    assert d["scaling"] == "weak" and d["config"]["input_bytes_total"] == 2 * 201326592


def test_bench_three_ranks_strong_split_with_a_ragged_last_range():
    """3 ranks, --strong: ONE stream of 250,000,000 B (14 full blocks + a ragged one) split 5 + 5 + 5 blocks."""
    d = _run(3, ["--steps", "1", "--warmup", "0", "--size", "250000000", "--strong", "--no-multistream"])
    _check_line(d, 3)
    assert d["scaling"] == "strong" and d["config"]["input_bytes_total"] == 250000000
