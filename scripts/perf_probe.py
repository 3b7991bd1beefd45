"""Stage timing probe: python scripts/perf_probe.py [MiB] [level] [--check] [--real] -- prints per-stage device ms
(--real: text files of this image, see real_text_soak.py, instead of the synthetic generator)."""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "oracle"))
import numpy as np
import torch
import libzling_amd as zl
from oracle_py import textgen, Oracle

mib = int(sys.argv[1]) if len(sys.argv) > 1 else 64
level = int(sys.argv[2]) if len(sys.argv) > 2 else 0
check = "--check" in sys.argv
n = mib << 20
if "--real" in sys.argv:
    sys.path.insert(0, os.path.join(ROOT, "scripts"))
    from real_text_soak import gather
    x, _ = gather(n)
    n = x.size
else:
    x = textgen(n, 0)
nb = (n + zl.BLOCK - 1) // zl.BLOCK
dx = torch.from_numpy(x).cuda()
dx = torch.cat([dx, torch.zeros(512, dtype=torch.uint8, device="cuda")])
cap = zl.encode_bound(n)
dout = torch.empty(cap, dtype=torch.uint8, device="cuda")
s = zl.Stream(0, level, True, nb)
st0, lv0 = s.get_state()
for it in range(2):
    s.set_state(st0, lv0)
    torch.cuda.synchronize()
    t = time.time()
    m = s.encode_device(dx.data_ptr(), n, dout.data_ptr(), cap)
    torch.cuda.synchronize()
    dt = time.time() - t
    print("iter %d: %d -> %d bytes, %.1f ms, %.1f MB/s" % (it, n, m, dt * 1e3, n / dt / 1e6))
    for name, ms in s.timings():
        print("   %-14s %10.3f ms" % (name, ms))
if check:
    z = dout[:m].cpu().numpy()
    ref = Oracle().encode(x, level)
    print("bit-exact vs oracle:", np.array_equal(z, ref))
    check = False
if os.environ.get("ZLNG_PROFILE") == "1":
    import ctypes as C
    SL = 24
    buf = (C.c_ulonglong * (SL * nb))()
    zl.lib().zlng_debug_counters(C.c_void_p(s._h), buf, nb)
    if os.environ.get("ZLNG_PARSER", "wg") == "wg":
        for b in range(min(nb, 4) if "--all" not in sys.argv else nb):
            d = buf[SL * b: SL * b + 24]
            r = max(d[3], 1)
            print("blk %2d wg: %5.0f Mcyc  rounds %6d tokens %7d  positions/round %.1f iterations/round %.2f serial tokens/round %.3f cut rounds %d | cycles per round: phase1 %5.0f tables %5.0f iterate %5.0f commit %5.0f | per serial token %5.0f" % (
                b, (d[0] + d[1] + d[2] + d[9] + d[8]) / 1e6, d[3], d[4], d[7] / r, d[5] / r, d[6] / r, d[11], d[0] / r, d[1] / r, d[2] / r, d[9] / r, d[8] / max(d[6], 1)))
            i = max(d[5], 1)
            print("        per iteration: chase+rank+deposit %5.0f  E (+closure) %5.0f  exchange+limit %5.0f | hard rounds %d | chase %5.0f rank+pty %5.0f deposit %5.0f" % (d[12] / i, d[13] / i, d[14] / i, d[10], d[15] / i, d[16] / i, d[17] / i))
            mk = [d[18 + k] & 0xFFFFFFFF for k in range(6)] + [d[18 + k] >> 32 for k in range(6)]
            print("        tables per round: rows(CAS) %5.0f closure %5.0f B1-wait %5.0f find_row %5.0f | E per iteration: rows+fix %5.0f lazy %5.0f mru %5.0f finish %5.0f closure-if-changed %5.0f Be-wait %5.0f (32-bit counters: may wrap)" % (
                mk[0] / r, mk[1] / r, mk[2] / r, mk[3] / r, mk[4] / i, mk[5] / i, mk[6] / i, mk[7] / i, mk[8] / i, mk[9] / i))
        sys.exit(0)
