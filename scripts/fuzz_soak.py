"""Many-seed differential of the GPU encoder against the oracle on the structured generators of tests/test_gpu_fuzz.py
and, since round 3 (the replay got 2x faster), the GPU decoder on every stream it produced.
python scripts/fuzz_soak.py [seeds=40] [max_bytes=1200000]"""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "oracle")); sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np
import libzling_amd as zl
from oracle_py import textgen, Oracle
from test_gpu_fuzz import make_input
o = Oracle()
text = textgen(3_000_000, 200)
seeds = int(sys.argv[1]) if len(sys.argv) > 1 else 40
cap = int(sys.argv[2]) if len(sys.argv) > 2 else 1_200_000
bad = 0; total = 0; t0 = time.time()
with zl.Stream(0, 0, True, 2) as s0, zl.Stream(0, 2, True, 2) as s2, zl.Stream(0, 4, True, 2) as s4, zl.Stream(0, 0, False, 2) as dec:
    streams = {0: s0, 2: s2, 4: s4}
    st = {lv: s.get_state() for lv, s in streams.items()}
    dst = dec.get_state()
    for seed in range(1000, 1000 + seeds):
        rng = np.random.Generator(np.random.PCG64(seed))
        for kind in range(8):
            n = int(rng.integers(1, cap))
            x = np.ascontiguousarray(make_input(rng, kind, n, text))
            for lv, s in streams.items():
                s.set_state(*st[lv])
                z = s.encode(x)
                ok = np.array_equal(z, o.encode(x, lv))
                dec.set_state(*dst)
                ok = ok and np.array_equal(dec.decode(z, x.size), x)
                total += 1
                if not ok:
                    bad += 1
                    print("MISMATCH seed %d kind %d level %d n %d" % (seed, kind, lv, x.size), flush=True)
print("fuzz soak: %d encodes + GPU decodes, %d mismatches, %.0f s" % (total, bad, time.time() - t0))
sys.exit(1 if bad else 0)
