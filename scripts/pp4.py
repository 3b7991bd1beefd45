import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "oracle"))
import numpy as np
import libzling_amd as zl
from oracle_py import textgen, Oracle
nb = 5
n = nb * zl.BLOCK
x = textgen(n, 0)
ref = Oracle().encode(x, 0)
# sub-block offsets in ref
offs = []; p = 0
while p < ref.size:
    if ref[p] == 0: p += 1; continue
    ol = int.from_bytes(ref[p+9:p+13].tobytes(), "big"); offs.append((p, ol)); p += 13 + ol
for it in range(3):
    s = zl.Stream(0, 0, True, nb)
    z = s.encode(x)
    bad = np.nonzero(z != ref)[0]
    print("run", it, "ndiff bytes", bad.size, "span", bad[:1], bad[-1:], "len", (bad[-1]-bad[0]+1) if bad.size else 0)
    if bad.size:
        k = max(i for i, (o, l) in enumerate(offs) if o <= bad[0])
        o, l = offs[k]
        print("  sub-block #%d at %d olen %d; diff starts %d bytes into it (payload+13); ends %d" % (k, o, l, bad[0]-o, bad[-1]-o))
        print("  gpu", z[bad[0]-4:bad[0]+12].tolist()); print("  ref", ref[bad[0]-4:bad[0]+12].tolist())
        # per-4-byte word alignment
        print("  first bad %4 =", bad[0] % 4, " distinct words:", np.unique(bad // 4).size)
