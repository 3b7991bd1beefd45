"""Same-box comparison of builds: `python scripts/ab_lib.py <file name of a libzlng_hip variant under libzling_amd/>` prints the
parse time of the 60-block benchmark batch with that library (boxes differ by +-2 %, so variants are compared inside one
gpurun call: build each, copy libzlng_hip.so aside under another name, run this per variant)."""
import os, sys, time
ROOT = os.environ.get("GRAFT_REPO_ROOT", "/root/repo")
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "oracle"))
import numpy as np, torch
import libzling_amd as zl
zl.HIP_SO = os.path.join(ROOT, "libzling_amd", sys.argv[1])
from oracle_py import textgen
n = 960 << 20
x = textgen(n, 0)
nb = n // zl.BLOCK
dx = torch.cat([torch.from_numpy(x).cuda(), torch.zeros(512, dtype=torch.uint8, device="cuda")])
cap = zl.encode_bound(n); dout = torch.empty(cap, dtype=torch.uint8, device="cuda")
s = zl.Stream(0, 0, True, nb); st0, lv0 = s.get_state()
for it in range(2):
    s.set_state(st0, lv0); torch.cuda.synchronize()
    m = s.encode_device(dx.data_ptr(), n, dout.data_ptr(), cap); torch.cuda.synchronize()
print(sys.argv[1], dict(s.timings())["rolz_parse"], m)
