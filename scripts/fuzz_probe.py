import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "oracle")); sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np
import libzling_amd as zl
from oracle_py import textgen, Oracle
from test_gpu_fuzz import make_input
o = Oracle()
text = textgen(3_000_000, 200)
n = int(sys.argv[1]) if len(sys.argv) > 1 else 300_000
for kind in range(8):
    for lv in (0, 4):
        rng = np.random.Generator(np.random.PCG64(kind))
        x = np.ascontiguousarray(make_input(rng, kind, n, text))
        t = time.time(); z = zl.encode(x, lv); te = time.time() - t
        ok = np.array_equal(z, o.encode(x, lv))
        t = time.time()
        with zl.Stream(0, 0, False, 1) as d:
            back = d.decode(z, x.size)
        td = time.time() - t
        print("kind %d lv %d n %d -> %d  enc %.2fs dec %.2fs ok %s rt %s" % (kind, lv, x.size, z.size, te, td, ok, np.array_equal(back, x)), flush=True)
