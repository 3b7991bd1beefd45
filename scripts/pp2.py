import os, sys
ROOT = "/root/repo" if os.path.exists("/root/repo/bench.py") else os.getcwd()
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "oracle"))
import numpy as np, hashlib
import libzling_amd as zl
from oracle_py import textgen, Oracle
n = 80 << 20
x = textgen(n, 0)
ref = Oracle().encode(x, 0)
for it in range(3):
    z = zl.Stream(0, 0, True, 5).encode(x)
    eq = np.array_equal(z, ref)
    d = -1 if eq else int(np.argmax(z[:min(z.size, ref.size)] != ref[:min(z.size, ref.size)]))
    print(os.environ.get("ZLNG_PARSER", "wave"), it, z.size, ref.size, eq, d)
