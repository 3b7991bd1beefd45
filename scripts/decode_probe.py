"""python scripts/decode_probe.py MiB [level] [text|rand] -- encode on GPU, decode on GPU, timings.
`rand` (incompressible bytes: literal tokens only) gives the replay's cost per literal."""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "oracle"))
import numpy as np
import libzling_amd as zl
from oracle_py import textgen
mib = int(sys.argv[1]) if len(sys.argv) > 1 else 32
level = int(sys.argv[2]) if len(sys.argv) > 2 else 0
kind = sys.argv[3] if len(sys.argv) > 3 else "text"
n = mib << 20
x = textgen(n, 0) if kind == "text" else np.random.default_rng(1).integers(0, 256, n, dtype=np.uint8)
nb = (n + zl.BLOCK - 1) // zl.BLOCK
with zl.Stream(0, level, True, nb) as s:
    t = time.time(); z = s.encode(x); te = time.time() - t
    print("encode e%d: %d -> %d  %.2f s  %.1f MB/s (host buffers, incl. PCIe)" % (level, n, z.size, te, n / te / 1e6), s.timings())
with zl.Stream(0, 0, False, nb) as d:
    t = time.time(); back = d.decode(z, n); td = time.time() - t
    tm = dict(d.timings())
    print("decode %s: %.2f s  %.1f MB/s  ok=%s  replay %.1f ms = %.1f ns per byte" % (kind, td, n / td / 1e6, np.array_equal(back, x), tm["rolz_decode"], tm["rolz_decode"] * 1e6 / n), d.timings())
