import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "oracle"))
import numpy as np, ctypes as C
import libzling_amd as zl
from oracle_py import textgen, Oracle
nb = 5
n = nb * zl.BLOCK
x = textgen(n, 0)
o = Oracle()
st = o.lib.zo_stream_new(0)
otoks = []
for b in range(nb):
    t, cuts = o.parse_block(x[b * zl.BLOCK:(b + 1) * zl.BLOCK], 0, apply_mtf=True, stream=st)
    otoks.append((t, cuts))
s = zl.Stream(0, 0, True, nb)
z = s.encode(x)
for b in range(nb):
    t, cuts = s.block_tokens(b)
    ot, ocuts = otoks[b]
    same = t.size == ot.size and np.array_equal(t, ot)
    print("block", b, "ntok", t.size, ot.size, "equal", same)
    if not same and t.size == ot.size:
        bad = np.nonzero(t != ot)[0]
        print("  ndiff", bad.size, "first", bad[:10], [hex(int(v)) for v in t[bad[:6]]], [hex(int(v)) for v in ot[bad[:6]]])
        ctxs = np.unique(ot[bad] >> 16)
        print("  contexts of differing tokens:", ctxs[:20], "count", ctxs.size)
        print("  first bad idx %% 512 = %d, %% 64 = %d" % (bad[0] % 512, bad[0] % 64))
