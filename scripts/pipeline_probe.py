"""Long-stream throughput of the C++ API (tools/zling_demo) with and without the two-context pipeline of the shim.
python scripts/pipeline_probe.py [blocks=192] [level=0]"""
import os, subprocess, sys, time, hashlib
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "oracle"))
from oracle_py import textgen
blocks = int(sys.argv[1]) if len(sys.argv) > 1 else 192
level = int(sys.argv[2]) if len(sys.argv) > 2 else 0
n = blocks << 24
src = "/tmp/pipe_in.bin"
textgen(n, 0).tofile(src)
demo = os.path.join(ROOT, "tools", "zling_demo")
digests = []
for mode in ("0", "1", "0", "1"):
    env = dict(os.environ, ZLNG_PIPELINE=mode)
    t = time.time()
    subprocess.check_call([demo, "e%d" % level, src, "/tmp/pipe_out.zlng"], env=env, stderr=subprocess.DEVNULL)
    dt = time.time() - t
    h = hashlib.sha256(open("/tmp/pipe_out.zlng", "rb").read()).hexdigest()
    digests.append(h)
    print("ZLNG_PIPELINE=%s: %d B in %.2f s = %.1f MB/s  (sha256 %s)" % (mode, n, dt, n / dt / 1e6, h[:16]))
assert len(set(digests)) == 1
