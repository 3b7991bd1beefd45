#!/usr/bin/env python3
"""HBM traffic per kernel launch from two rocprofv3 --pmc passes (FETCH_SIZE, WRITE_SIZE; --kernel-trace only).

    python scripts/pmc_traffic.py fetch_results.db write_results.db bytes level blocks > profiles/rNN_pmc_traffic.json
Units and the gfx950 correction follow /opt/skills/guides/MI355X_MICROARCH.md (HBM / rocprofv3 section): the counters are in
KB (1024 B) and FETCH_SIZE counts 128-B requests as 64 B, so hbm_bytes = (2 * FETCH + WRITE) * 1024 per launch."""
import glob, hashlib, json, os, re, sqlite3, sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


ENCODE_KERNEL_SOURCES = ("zlng_common.h", "zlng_kernels.h", "rolz_dev.h", "rolz_wg.hip", "mtf_rank.hip", "huffman.hip")


def kernel_source_sha():
    """Same identity bench.py computes: the profile is only quoted for the encode-kernel sources it was taken on."""
    h = hashlib.sha256()
    for f in ENCODE_KERNEL_SOURCES:
        h.update(f.encode())
        h.update(open(os.path.join(ROOT, "libzling_amd", "csrc", f), "rb").read())
    return h.hexdigest()[:16]


def per_kernel(path, counter):
    db = sqlite3.connect(path)
    out = {}
    q = "select kernel_name, dispatch_id, sum(value), max(duration) from counters_collection where counter_name = ? group by dispatch_id"
    for name, _, val, dur in db.execute(q, (counter,)):
        short = re.sub(r"\(.*$", "", name).replace("void ", "").replace("zlng::", "")
        short = re.sub(r"^(k_rolz_parse_wave|k_rolz_parse_wg)<.*>$", r"\1", short)
        o = out.setdefault(short, {"n": 0, "v": 0.0, "ns": 0.0})
        o["n"] += 1; o["v"] += val; o["ns"] += dur
    return {k: (o["v"] / o["n"], o["ns"] / o["n"] / 1e6) for k, o in out.items()}

f, w = per_kernel(sys.argv[1], "FETCH_SIZE"), per_kernel(sys.argv[2], "WRITE_SIZE")
kern = {}
for k in f:
    if k not in w or not k.startswith("k_"): continue
    kern[k] = {"FETCH_SIZE_KB": f[k][0], "dur_ms_fetch": round(f[k][1], 6), "WRITE_SIZE_KB": w[k][0], "dur_ms_write": round(w[k][1], 6),
               "hbm_bytes_corrected": int((2 * f[k][0] + w[k][0]) * 1024)}
print(json.dumps({
    "note": "rocprofv3 --pmc FETCH_SIZE / --pmc WRITE_SIZE (separate passes, --kernel-trace only), per-launch averages. "
            "Unit KB (1024 B); gfx950 FETCH_SIZE counts 128-B requests as 64 B, so hbm_bytes = (2*FETCH + WRITE)*1024 as the guide "
            "prescribes (calibrated in round 1 on k_dict_reset / k_pack / k_histogram); for the parser's narrow random reads the x2 is an upper estimate.",
    "kernel_source_sha": kernel_source_sha(),
    "workload": {"bytes": int(sys.argv[3]), "level": int(sys.argv[4]), "blocks": int(sys.argv[5])},
    "kernels": kern}, indent=1))
