import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "oracle"))
import numpy as np
import libzling_amd as zl
from oracle_py import textgen, Oracle
nb = 5
n = nb * zl.BLOCK
x = textgen(n, 0)
o = Oracle()
s = zl.Stream(0, 0, True, nb)
z = s.encode(x)
for b in range(nb):
    tok, cuts = s.block_tokens(b)
    ns = cuts.shape[0]
    freq = s.debug_fetch(2, b, np.uint32, 80 * 546).reshape(80, 546)
    lens = s.debug_fetch(3, b, np.uint8, 80 * 546).reshape(80, 546)
    codes = s.debug_fetch(6, b, np.uint16, 80 * 546).reshape(80, 546)
    olen = s.debug_fetch(4, b, np.uint32, 80)
    off = s.debug_fetch(7, b, np.uint64, 80)
    for k in range(ns):
        t = tok[cuts[k, 0]:cuts[k, 1]]
        f1, f2 = o.histogram(t)
        l1 = o.length_table(f1, 15); l2 = o.length_table(f2, 8)
        c1 = o.encode_table(l1, 15); c2 = o.encode_table(l2, 8)
        okf = np.array_equal(freq[k, :514], f1) and np.array_equal(freq[k, 514:], f2)
        okl = np.array_equal(lens[k, :514], l1) and np.array_equal(lens[k, 514:], l2)
        okc = np.array_equal(codes[k, :514], c1) and np.array_equal(codes[k, 514:], c2)
        pay = o.pack(t, l1, l2)
        g = z[int(off[k]) + 13: int(off[k]) + 13 + int(olen[k])]
        okp = pay.size == olen[k] and np.array_equal(g, pay)
        if not (okf and okl and okc and okp):
            nbad = int((g != pay).sum()) if pay.size == g.size else -1
            print("blk %d sub %d: freq %s lens %s codes %s pack %s (bad bytes %d, first %d) off %d" % (b, k, okf, okl, okc, okp, nbad, int(np.argmax(g != pay)) if nbad > 0 else -1, off[k]))
print("done")
