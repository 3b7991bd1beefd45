#!/usr/bin/env python3
"""K INDEPENDENT streams at once on ONE GPU -- a number about K files, the counterpart of bench.py's `cpu_baseline_multistream`
(the reference on K host threads), never the metric (BASELINE's metric is ONE stream, whose rank chain is one serial chain).

    python scripts/multi_stream_probe.py [K=4] [bytes per stream=1000000000] [level=0]

Every stream has its own context (its own literal tables, HIP stream and pools); K host threads call zlng_encode_blocks_device at the
same time, so the K parses share the chip's CUs (60 workgroups each) and the K rank chains run side by side.  Stream 0 is the
benchmark stream itself (its SHA-256 is the real reference's pin); the first 64 MiB of every other stream are compared byte for
byte with the CPU encoder.  Prints one JSON line."""
import hashlib
import json
import os
import sys
import threading
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "oracle"))
os.environ.setdefault("GPU_MAX_HW_QUEUES", "8")
import numpy as np  # noqa: E402
import torch  # noqa: E402

import libzling_amd as zl  # noqa: E402
from oracle_py import Oracle, Reference, textgen  # noqa: E402


def main():
    k = int(sys.argv[1]) if len(sys.argv) > 1 else 4
    n = int(sys.argv[2]) if len(sys.argv) > 2 else 1_000_000_000
    level = int(sys.argv[3]) if len(sys.argv) > 3 else 0
    nb = (n + zl.BLOCK - 1) // zl.BLOCK
    cap = zl.encode_bound(n)
    xs = [textgen(n, 4096 * i) for i in range(k)]
    d_in, d_out, ctx = [], [], []
    for x in xs:
        t = torch.empty(n + 512, dtype=torch.uint8, device="cuda")
        t[:n].copy_(torch.from_numpy(x)); t[n:].zero_()
        d_in.append(t)
        d_out.append(torch.empty(cap, dtype=torch.uint8, device="cuda"))
        ctx.append(zl.Stream(0, level, True, nb))
    st0 = [c.get_state() for c in ctx]
    lens = [0] * k

    def one(i):
        ctx[i].set_state(*st0[i])
        lens[i] = ctx[i].encode_device(d_in[i].data_ptr(), n, d_out[i].data_ptr(), cap)

    def all_at_once():
        th = [threading.Thread(target=one, args=(i,)) for i in range(k)]
        for t in th:
            t.start()
        for t in th:
            t.join()
        torch.cuda.synchronize()

    all_at_once()                                                    # warm-up
    t0 = time.perf_counter()
    reps = 2
    for _ in range(reps):
        all_at_once()
    dt = (time.perf_counter() - t0) / reps
    one(0); torch.cuda.synchronize()                                 # the same stream alone, for the ratio
    t1 = time.perf_counter(); one(0); torch.cuda.synchronize(); alone = time.perf_counter() - t1
    all_at_once()                                                    # the outputs checked below come from a concurrent run
    cpu = Reference() if Reference.available() else Oracle()
    ok = []
    for i in range(k):
        got = d_out[i][: lens[i]].cpu().numpy()
        z = cpu.encode(xs[i][: 4 * zl.BLOCK], level)
        ok.append(bool(np.array_equal(got[: z.size], z)))
    sha0 = hashlib.sha256(d_out[0][: lens[0]].cpu().numpy().tobytes()).hexdigest()
    pin = None
    try:
        man = json.load(open(os.path.join(ROOT, "tests", "golden", "manifest.json")))["config3_enwik9_shape"]
        if n == man["bytes"] and level == man["level"]:
            pin = bool(sha0 == man["sha256"])
    except Exception:
        pass
    print(json.dumps({"what": "EXTRA, not the metric: %d independent streams of %d B at e%d at once on one GPU (one context and one host thread each)" % (k, n, level),
                      "value": round(k * n / dt / 1e6, 2), "unit": "MB/s", "seconds_per_round": round(dt, 4), "one_stream_alone_s": round(alone, 4),
                      "prefix_parity_per_stream": ok, "stream0_sha256_is_the_reference_pin": pin, "zlng_bytes": [int(v) for v in lens]}))
    for c in ctx:
        c.close()


if __name__ == "__main__":
    main()
