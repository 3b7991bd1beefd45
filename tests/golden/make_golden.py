#!/usr/bin/env python3
"""Generate tests/golden/ from the REAL reference (oracle/_ref/libzling_ref.so).

Run in the build container only (it needs /root/reference to have been compiled by
oracle/Makefile):   python tests/golden/make_golden.py
Writes:
  <name>.bin / <name>.e<level>.zlng   for tests/corpus.py SMALL (inputs and reference streams)
  manifest.json                        SHA-256 of every input and of every reference stream,
                                       stream sizes, per-sub-block (encpos, rlen, olen) lists,
                                       SHA-256 of the reference ROLZ u16 token stream per block
  huff_tables.npz                      tie-heavy frequency tables + the reference's code lengths
The files are data (inputs and expected outputs); no reference source is stored.
"""
import json
import os
import sys

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.dirname(HERE))
import corpus  # noqa: E402
from oracle_py import Reference  # noqa: E402


def walk(z):
    """(encpos, rlen, olen) per sub-block, -1 marks a block end."""
    out, p = [], 0
    while p < len(z):
        f = z[p]; p += 1
        if f == 0:
            out.append(-1)
            continue
        e, r, o = (int.from_bytes(z[p + 4 * k: p + 4 * k + 4].tobytes(), "big") for k in range(3))
        p += 12 + o
        out.append([e, r, o])
    assert p == len(z)
    return out


def config4_share():
    """BASELINE config 4's per-GPU share through the reference: SHA-256 and size of the e4 .zlng of the first 8 GiB of the
    synthetic stream (needs ~11 GB of memory and a couple of minutes); updates manifest.json in place."""
    import hashlib
    from oracle_py import textgen
    n = 8 << 30
    x = textgen(n, 0)
    z = Reference().encode(x, 4)
    man = json.load(open(os.path.join(HERE, "manifest.json")))
    man["config4_share"].update({"bytes": n, "level": 4, "zlng_bytes": int(z.size), "sha256": hashlib.sha256(z.tobytes()).hexdigest()})
    json.dump(man, open(os.path.join(HERE, "manifest.json"), "w"), indent=1)
    print("config4_share:", man["config4_share"]["zlng_bytes"], man["config4_share"]["sha256"])


def config_streams():
    """BASELINE configs 1-3 and 5 at their own sizes through the reference: SHA-256 and size of the e0 .zlng of the first
    100,000,000 and 1,000,000,000 bytes of the synthetic stream (enwik8 / enwik9 shaped; about 15 s of one host core)."""
    import hashlib
    from oracle_py import textgen
    man = json.load(open(os.path.join(HERE, "manifest.json")))
    ref = Reference()
    for key, n in (("config12_enwik8_shape", 100_000_000), ("config3_enwik9_shape", 1_000_000_000)):
        x = textgen(n, 0)
        z = ref.encode(x, 0)
        man[key] = {"what": "first %d bytes of the synthetic text stream (textgen chunk 0) at e0" % n, "bytes": n, "level": 0,
                    "input_sha256": hashlib.sha256(x.tobytes()).hexdigest(), "zlng_bytes": int(z.size),
                    "sha256": hashlib.sha256(z.tobytes()).hexdigest(),
                    "provenance": "the REAL reference (oracle/_ref) over the whole input in this container: tests/golden/make_golden.py --configs"}
        print(key, man[key]["zlng_bytes"], man[key]["sha256"])
    json.dump(man, open(os.path.join(HERE, "manifest.json"), "w"), indent=1)


def block_ends(z):
    """Offset behind every block of a .zlng stream (Appendix A: a block is its sub-blocks followed by one 0x00)."""
    ends, p, n = [], 0, len(z)
    while p < n:
        if z[p] == 0:
            p += 1
            ends.append(p)
            continue
        p += 13 + int.from_bytes(z[p + 9: p + 13].tobytes(), "big")
    assert p == n and (n == 0 or ends[-1] == n)
    return ends


def sharded_ranges():
    """What every rank of `bench.py --gpus N` must produce, from the REAL reference: one stream of N x 10^9 bytes (weak, the
    driver's scaling runs) resp. 10^9 bytes (--strong), split by libzling_amd.sharding.plan into contiguous block ranges.  The
    .zlng of a block range is the slice of the reference's stream between the block ends that bound it (the reference never
    starts a block's bytes before it has pushed the previous block's terminator, src/libzling.cpp:187-284).  Pins size + SHA-256
    per rank for N = 1, 2, 4, 8 (and 3, which the one-device control-flow test runs); about 3 minutes of one host core and
    11 GB of memory.  The weak streams are prefixes of one another up to the last, ragged block of each, so each N is run whole."""
    import hashlib
    from oracle_py import textgen
    sys.path.insert(0, os.path.dirname(os.path.dirname(HERE)))
    from libzling_amd import sharding
    man = json.load(open(os.path.join(HERE, "manifest.json")))
    ref = Reference()
    per = 1_000_000_000
    out = {"what": "per-rank .zlng of bench.py's sharded streams at e0 (textgen chunk 0 onwards), sharding.plan ranges",
           "provenance": "the REAL reference (oracle/_ref) over each whole stream in this container, sliced at its block ends: tests/golden/make_golden.py --ranges",
           "per_gpu_bytes": per, "level": 0, "weak": {}, "strong": {}}

    def pins(z, ranges):
        ends = [0] + block_ends(z)
        res = []
        for off, n in ranges:
            b0, b1 = off // corpus.BLOCK, (off + n + corpus.BLOCK - 1) // corpus.BLOCK
            seg = z[ends[b0]: ends[b1]]
            res.append({"offset": int(off), "bytes": int(n), "zlng_bytes": int(seg.size), "sha256": hashlib.sha256(seg.tobytes()).hexdigest()})
        assert sum(r["zlng_bytes"] for r in res) == z.size
        return res
    for world in (1, 2, 3, 4, 8):
        x = textgen(per * world, 0)
        z = ref.encode(x, 0)
        out["weak"][str(world)] = {"stream_bytes": int(x.size), "zlng_bytes": int(z.size), "sha256": hashlib.sha256(z.tobytes()).hexdigest(),
                                   "ranks": pins(z, sharding.plan(per * world, world, per_rank_bytes=per))}
        print("weak", world, z.size, out["weak"][str(world)]["sha256"], flush=True)
        if world == 1:
            assert out["weak"]["1"]["sha256"] == man["config3_enwik9_shape"]["sha256"]
            for w in (2, 3, 4, 8):
                out["strong"][str(w)] = {"stream_bytes": per, "zlng_bytes": int(z.size), "ranks": pins(z, sharding.plan(per, w))}
        del x, z
    man["sharded_ranges"] = out
    json.dump(man, open(os.path.join(HERE, "manifest.json"), "w"), indent=1)


def main():
    if "--config4" in sys.argv:
        return config4_share()
    if "--ranges" in sys.argv:
        return sharded_ranges()
    if "--configs" in sys.argv:
        return config_streams()
    ref = Reference()
    man = {"inputs": {}, "streams": {}, "rolz": {}}
    try:
        old = json.load(open(os.path.join(HERE, "manifest.json")))
        for k in ("config4_share", "config12_enwik8_shape", "config3_enwik9_shape", "sharded_ranges"):      # expensive: kept unless --config4 / --configs / --ranges
            if k in old:
                man[k] = old[k]
    except Exception:
        pass
    for name in corpus.ALL:
        x = corpus.get(name)
        man["inputs"][name] = {"size": int(x.size), "sha256": corpus.sha(x)}
        small = name in corpus.SMALL
        if small:
            x.tofile(os.path.join(HERE, name + ".bin"))
        levels = range(5) if (small or name in ("mixed_e4", "text_700k")) else (0, 4)
        for lv in levels:
            z = ref.encode(x, lv)
            rc, back, _ = ref.decode(z, x.size)
            assert rc == 0 and np.array_equal(back, x), (name, lv)
            key = "%s.e%d" % (name, lv)
            man["streams"][key] = {"size": int(z.size), "sha256": corpus.sha(z)}
            if small:
                z.tofile(os.path.join(HERE, key + ".zlng"))
            else:
                man["streams"][key]["subblocks"] = walk(z)
        if name in ("text_700k", "rand_1m", "abc_1m", "skew_400k"):
            for lv in (0, 4):
                t, cuts = ref.rolz_block(x[: corpus.BLOCK], lv)
                man["rolz"]["%s.e%d" % (name, lv)] = {"sha256_u16": corpus.sha(t), "cuts": cuts}
    # Huffman length tables: tie-heavy distributions
    rng = np.random.Generator(np.random.PCG64(99))
    freqs, lens, meta = [], [], []
    for n, limit in ((514, 15), (32, 8)):
        for k in range(150):
            kind = k % 6
            if kind == 0: f = rng.integers(0, 4, n)
            elif kind == 1: f = rng.integers(0, 2, n) * rng.integers(1, 3, n)
            elif kind == 2: f = (rng.zipf(1.3, n) % 100000) * (rng.random(n) < 0.7)
            elif kind == 3: f = np.where(rng.random(n) < 0.3, 1, 0) + (np.arange(n) < 3) * rng.integers(1000, 100000)
            elif kind == 4: f = (2.0 ** rng.integers(0, 24, n)).astype(np.int64) * (rng.random(n) < 0.5)
            else: f = rng.integers(0, 262144, n) * (rng.random(n) < 0.2)
            f = np.asarray(f, dtype=np.uint32)
            if k == 7: f[:] = 0
            if k == 8: f[:] = 0; f[n // 2] = 5
            full = np.zeros(514, np.uint32); full[:n] = f
            l = np.zeros(514, np.uint32); l[:n] = ref.length_table(f, limit)
            freqs.append(full); lens.append(l); meta.append((n, limit))
    np.savez_compressed(os.path.join(HERE, "huff_tables.npz"), freq=np.array(freqs), len=np.array(lens),
                        meta=np.array(meta))
    with open(os.path.join(HERE, "manifest.json"), "w") as f:
        json.dump(man, f, indent=0, separators=(",", ":"))
    print("wrote", len(man["inputs"]), "inputs,", len(man["streams"]), "streams")


if __name__ == "__main__":
    main()
