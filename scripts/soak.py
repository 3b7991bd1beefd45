"""Soak parity: several large streams (different seeds / levels) GPU vs the reference encoder, byte for byte."""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "oracle"))
import numpy as np
import libzling_amd as zl
from oracle_py import textgen, Oracle, Reference
ref = Reference() if Reference.available() else Oracle()
cases = [(256, 0, 1000), (256, 0, 2000), (192, 0, 3000), (96, 1, 4000), (96, 2, 5000), (64, 3, 6000), (64, 4, 7000)]
if len(sys.argv) > 1:                    # python scripts/soak.py N: N further rounds of the same mix with other seeds
    base = list(cases)
    for r in range(1, int(sys.argv[1]) + 1):
        cases += [(m, lv, sd + 17 * r) for m, lv, sd in base]
rng = np.random.Generator(np.random.PCG64(77))
bad = 0
for mib, lv, seed in cases:
    n = (mib << 20) - int(rng.integers(0, 100000))
    x = textgen(n, seed)
    # sprinkle structure: a few binary / repetitive patches
    for _ in range(20):
        o = int(rng.integers(0, n - 70000)); k = int(rng.integers(100, 60000))
        kind = int(rng.integers(0, 3))
        if kind == 0: x[o:o + k] = rng.integers(0, 256, k, dtype=np.uint8)
        elif kind == 1: x[o:o + k] = x[o - k:o] if o >= k else 0
        else: x[o:o + k] = np.resize(rng.integers(97, 123, int(rng.integers(1, 9)), dtype=np.uint8), k)
    nb = (n + zl.BLOCK - 1) // zl.BLOCK
    t = time.time()
    with zl.Stream(0, lv, True, nb) as s:
        z = s.encode(x)
    tg = time.time() - t
    t = time.time(); r = ref.encode(x, lv); tr = time.time() - t
    ok = z.size == r.size and np.array_equal(z, r)
    bad += not ok
    print("e%d %4d MiB seed %d: gpu %.2fs ref %.2fs  %d -> %d  %s" % (lv, mib, seed, tg, tr, n, z.size, "OK" if ok else "MISMATCH at %d" % int(np.argmax(z[:min(z.size, r.size)] != r[:min(z.size, r.size)]))), flush=True)
print("soak done, mismatches:", bad)
sys.exit(1 if bad else 0)
