"""CLI wall time on small inputs (process start + context creation dominate).  python scripts/small_probe.py"""
import os, subprocess, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "oracle"))
from oracle_py import textgen
demo = os.path.join(ROOT, "tools", "zling_demo")
for name, n in (("1 MB", 1_000_000), ("100 MB", 100_000_000)):
    src = "/tmp/sp_in.bin"
    textgen(n, 0).tofile(src)
    for bb in (64, 8, 1):
        best = 1e9
        for rep in range(2):
            t = time.time()
            subprocess.check_call([demo, "e0", src, "/tmp/sp_out.zlng"], env=dict(os.environ, ZLNG_BATCH_BLOCKS=str(bb)), stderr=subprocess.DEVNULL)
            best = min(best, time.time() - t)
        print("%7s  ZLNG_BATCH_BLOCKS=%-3d %.2f s" % (name, bb, best), flush=True)
