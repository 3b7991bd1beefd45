"""python scripts/parity_probe.py MiB [level] -- GPU vs reference/oracle on MiB of synthetic text; reports first difference."""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "oracle"))
import numpy as np
import libzling_amd as zl
from oracle_py import textgen, Oracle, Reference
mib = int(sys.argv[1]); level = int(sys.argv[2]) if len(sys.argv) > 2 else 0
n = mib << 20
x = textgen(n, 0)
nb = (n + zl.BLOCK - 1) // zl.BLOCK
s = zl.Stream(0, level, True, nb)
z = s.encode(x)
ends = s.block_ends
ref = (Reference() if Reference.available() else Oracle()).encode(x, level)
print("gpu", z.size, "ref", ref.size, "equal", np.array_equal(z, ref))
if not np.array_equal(z, ref):
    m = min(z.size, ref.size)
    d = int(np.argmax(z[:m] != ref[:m])) if (z[:m] != ref[:m]).any() else m
    blk = next((i for i, e in enumerate(ends) if d < e), -1)
    print("first diff at byte", d, "in block", blk, "block ends", ends[:blk + 2])
    # which blocks differ when encoded alone?
    for b in range(nb):
        xb = x[b * zl.BLOCK:(b + 1) * zl.BLOCK]
        zb = zl.Stream(0, level, True, 1).encode(xb)
        rb = Oracle().encode(xb, level)
        print("block", b, "alone equal:", np.array_equal(zb, rb))
