import os, sys, ctypes as C
sys.path.insert(0, '/root/repo'); sys.path.insert(0, '/root/repo/oracle'); sys.path.insert(0,'/root/repo/scripts')
import numpy as np, torch
import libzling_amd as zl
from oracle_py import textgen
real = "--real" in sys.argv
n = 512 << 20
if real:
    from real_text_soak import gather
    x, _ = gather(n); n = x.size
else:
    x = textgen(n, 0)
nb = (n + zl.BLOCK - 1) // zl.BLOCK
s = zl.Stream(0, 0, True, max(nb, 22))
z = s.encode(x)
print(dict(s.timings()))
buf = (C.c_ulonglong * (24 * max(nb, 22)))()
zl.lib().zlng_debug_counters(C.c_void_p(s._h), buf, max(nb, 22))
tot = s.debug_fetch(8, 0, np.uint32, 256)
rows = sorted(((buf[c], c) for c in range(256)), reverse=True)[:8]
for cyc, c in rows:
    nev = buf[256 + c]
    print("ctx %3d %r: %7.1f ms  literals %9d  slow steps (literal outside the table front, rank >= 60): %7d (%.2f%% of the literals)  %.1f ns per literal" % (
        c, chr(c), cyc / 2.4e6, tot[c], nev, 100.0 * nev / max(tot[c], 1), cyc / 2.4 / max(tot[c], 1)))
