"""Host-buffer entry point (zlng_encode_blocks) vs device-resident (zlng_encode_blocks_device) on the same 10^9-byte stream."""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "oracle"))
import numpy as np, torch
import libzling_amd as zl
from oracle_py import textgen
n = int(sys.argv[1]) if len(sys.argv) > 1 else 1_000_000_000
x = textgen(n, 0)
nb = (n + zl.BLOCK - 1) // zl.BLOCK
s = zl.Stream(0, 0, True, nb)
st, lv = s.get_state()
for it in range(2):
    s.set_state(st, lv)
    t = time.perf_counter(); z = s.encode(x); dt = time.perf_counter() - t
    print("host buffers (pageable H2D + D2H included): %.3f s  %.1f MB/s  -> %d B" % (dt, n / dt / 1e6, z.size))
dx = torch.empty(n + 512, dtype=torch.uint8, device="cuda"); dx[:n].copy_(torch.from_numpy(x)); dx[n:].zero_()
do = torch.empty(zl.encode_bound(n), dtype=torch.uint8, device="cuda")
for it in range(2):
    s.set_state(st, lv); torch.cuda.synchronize()
    t = time.perf_counter(); m = s.encode_device(dx.data_ptr(), n, do.data_ptr(), do.numel()); torch.cuda.synchronize(); dt = time.perf_counter() - t
    print("device resident: %.3f s  %.1f MB/s" % (dt, n / dt / 1e6))
