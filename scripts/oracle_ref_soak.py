"""CPU-only differential soak of the ORACLE against the REAL reference (oracle/_ref, compiled in place by oracle/Makefile) on the
structured generators of tests/test_gpu_fuzz.py at all five levels, plus the oracle's decoder on every stream.  Test infrastructure
checking test infrastructure: no product code runs here.
    python scripts/oracle_ref_soak.py <first seed> <seeds> <max bytes>        e.g. 7000 60 1500000"""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "oracle")); sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np
from oracle_py import textgen, Oracle, Reference
from test_gpu_fuzz import make_input
o = Oracle(); r = Reference()
text = textgen(3_000_000, 200)
s0 = int(sys.argv[1]); n = int(sys.argv[2]); cap = int(sys.argv[3])
bad = total = 0; t0 = time.time()
for seed in range(s0, s0 + n):
    rng = np.random.Generator(np.random.PCG64(seed))
    for kind in range(8):
        m = int(rng.integers(1, cap))
        x = np.ascontiguousarray(make_input(rng, kind, m, text))
        for lv in range(5):
            z = o.encode(x, lv); zr = r.encode(x, lv)
            ok = np.array_equal(z, zr)
            rc, back, fl = o.decode_ex(z, x.size + 16)
            ok = ok and rc == 0 and np.array_equal(back, x)
            total += 1
            if not ok:
                bad += 1; print("MISMATCH seed %d kind %d level %d n %d" % (seed, kind, lv, x.size), flush=True)
print("cpu soak seeds %d..%d: %d encodes (oracle vs real reference, 5 levels) + oracle decodes, %d mismatches, %.0f s" % (s0, s0 + n - 1, total, bad, time.time() - t0))
