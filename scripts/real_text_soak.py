"""Parity and rate on REAL text from the image (no enwik file exists on any box): source and documentation text
that ships with ROCm and the Python packages -- C/C++ headers, .py, .md, .rst, .txt, .json -- concatenated in sorted path
order (deterministic for one image), up to `limit` bytes.  GPU encode at e0 and e4 vs the reference, byte for byte.

    python scripts/real_text_soak.py [limit_mib=1024] [levels=0,4] [decode_mib=0]
(decode_mib > 0: the first decode_mib MiB are also decoded on the GPU from the GPU's own stream and compared with the input)
"""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "oracle"))
import numpy as np
import libzling_amd as zl
from oracle_py import Oracle, Reference

ROOTS = ["/opt/rocm/include", "/usr/local/lib/python3.10/dist-packages", "/usr/lib/python3/dist-packages", "/usr/share/doc"]
EXT = (".h", ".hpp", ".hip", ".inc", ".py", ".pyi", ".md", ".rst", ".txt", ".json", ".cmake", ".yaml", ".yml", ".cfg", ".html")


def gather(limit):
    parts, total, nfiles = [], 0, 0
    for r in ROOTS:
        for d, dn, fn in os.walk(r):
            dn.sort()
            for f in sorted(fn):
                if not f.endswith(EXT):
                    continue
                p = os.path.join(d, f)
                try:
                    if os.path.islink(p) or os.path.getsize(p) > (8 << 20):
                        continue
                    b = np.fromfile(p, dtype=np.uint8)
                except OSError:
                    continue
                parts.append(b); total += b.size; nfiles += 1
                if total >= limit:
                    return np.concatenate(parts)[:limit], nfiles
    return np.concatenate(parts)[:limit], nfiles


def main():
    limit = (int(sys.argv[1]) if len(sys.argv) > 1 else 1024) << 20
    levels = [int(v) for v in (sys.argv[2] if len(sys.argv) > 2 else "0,4").split(",")]
    dec = (int(sys.argv[3]) if len(sys.argv) > 3 else 0) << 20
    x, nfiles = gather(limit)
    print("real text: %d files, %d bytes" % (nfiles, x.size), flush=True)
    ref = Reference() if Reference.available() else Oracle()
    nb = (x.size + zl.BLOCK - 1) // zl.BLOCK
    bad = 0
    for lv in levels:
        with zl.Stream(0, lv, True, min(nb, 240)) as s:
            s.encode(x[: 2 * zl.BLOCK])
        with zl.Stream(0, lv, True, min(nb, 240)) as s:
            t = time.time(); z = s.encode(x); tg = time.time() - t
            passes = s.passes()
            stages = ", ".join("%s %.0f" % (k, v) for k, v in s.timings() if v >= 1.0)
        t = time.time(); r = ref.encode(x, lv); tr = time.time() - t
        ok = z.size == r.size and np.array_equal(z, r)
        bad += not ok
        print("e%d: %d -> %d (ratio %.4f)  GPU %.2f s = %.0f MB/s host to host (%d parse pass%s), reference %.2f s = %.0f MB/s  %s" % (
            lv, x.size, z.size, z.size / x.size, tg, x.size / tg / 1e6, passes, "" if passes == 1 else "es", tr, x.size / tr / 1e6,
            "OK" if ok else "MISMATCH at %d" % int(np.argmax(z[:min(z.size, r.size)] != r[:min(z.size, r.size)]))), flush=True)
        print("    device stages (ms): " + stages, flush=True)
        if dec:
            n = min(dec, x.size)
            with zl.Stream(0, lv, True, 8) as s:
                zs = s.encode(x[:n])
            with zl.Stream(0, lv, False, 8) as d:
                t = time.time(); y = d.decode(zs, n); td = time.time() - t
            okd = y.size == n and np.array_equal(y, x[:n])
            bad += not okd
            print("    GPU decode of the first %d bytes: %.2f s  %s" % (n, td, "round trip OK" if okd else "ROUND TRIP MISMATCH"), flush=True)
    return 1 if bad else 0


if __name__ == "__main__":
    sys.exit(main())
