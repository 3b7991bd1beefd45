"""Wall-clock table in the spirit of the reference's benchmark/benchmark.sh:41-48 (encode time, decode time, size, round
trip) for the GPU path through tools/zling_demo, the reference CPU codec (oracle/_ref, in-process) and gzip/bzip2/xz,
on an enwik8-sized (10^8 B) slice of the synthetic text.   python scripts/benchmark_table.py [bytes=100000000]"""
import os, subprocess, sys, time, shutil
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "oracle"))
import numpy as np
from oracle_py import textgen, Reference
n = int(sys.argv[1]) if len(sys.argv) > 1 else 100_000_000
src, enc, dec = "/tmp/bt_in.bin", "/tmp/bt_enc", "/tmp/bt_dec"
x = textgen(n, 0)
x.tofile(src)
demo = os.path.join(ROOT, "tools", "zling_demo")
cpu = open("/proc/cpuinfo").read().split("model name")[1].split(":")[1].split("\n")[0].strip()
print("CPU: %s   input: %d B synthetic enwik-shaped text" % (cpu, n))
print("%-26s | %9s | %9s | %11s | %s" % ("codec", "encode s", "decode s", "size", "round trip"))
def row(name, te, td, size, ok):
    print("%-26s | %9.2f | %9.2f | %11d | %s" % (name, te, td, size, "PASS" if ok else "FAIL"), flush=True)
def shell(name, ecmd, dcmd):
    t = time.time(); subprocess.check_call("%s < %s > %s" % (ecmd, src, enc), shell=True, stderr=subprocess.DEVNULL); te = time.time() - t
    t = time.time(); subprocess.check_call("%s < %s > %s" % (dcmd, enc, dec), shell=True, stderr=subprocess.DEVNULL); td = time.time() - t
    row(name, te, td, os.path.getsize(enc), subprocess.call(["cmp", "-s", src, dec]) == 0)
for lv in range(5):
    shell("zling_demo e%d (MI355X)" % lv, "%s e%d" % (demo, lv), "%s d" % demo)
ref = Reference()
for lv in range(5):
    t = time.time(); z = ref.encode(x, lv); te = time.time() - t
    t = time.time(); rc, y, _ = ref.decode(z, n); td = time.time() - t
    row("reference e%d (1 thread)" % lv, te, td, z.size, rc == 0 and np.array_equal(x, y))
for name in ("gzip", "bzip2", "xz"):
    if shutil.which(name):
        shell(name, "%s -c" % name, "%s -d" % name)
