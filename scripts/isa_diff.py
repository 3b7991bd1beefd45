#!/usr/bin/env python3
"""Which kernels' MACHINE CODE differs between this tree and a commit -- no GPU needed (hipcc cross-compiles to gfx950 assembly).

    python scripts/isa_diff.py b703640            # the last commit before the GPU pool closed (round 5; profiles/r05_d_gpu_tests.txt: 134 passed)
    python scripts/isa_diff.py --pin b703640 tests/golden/kernel_isa_last_gpu_run.json      # what tests/test_isa_hygiene.py compares with

For every .hip file: compile both trees' sources to assembly (-S --cuda-device-only), split at the kernel symbols, strip comments,
labels' numbering and debug directives, and compare instruction streams kernel by kernel.  A kernel whose stream is identical is, as
far as the device is concerned, the code that ran then -- whatever moved in the source around it."""
import hashlib
import os
import re
import subprocess
import sys
import tempfile

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
HIPCC = "/opt/rocm/bin/hipcc"


def kernels(csrc, name):
    with tempfile.TemporaryDirectory() as t:
        out = os.path.join(t, "x.s")
        r = subprocess.run([HIPCC, "--offload-arch=gfx950", "-O3", "-std=c++17", "-munsafe-fp-atomics", "-S", "--cuda-device-only", "-Wno-everything",
                            "-o", out, os.path.join(csrc, name)], stderr=subprocess.PIPE, text=True, cwd=csrc)
        if r.returncode != 0:
            return None
        text = open(out).read()
    res, cur = {}, None
    for ln in text.splitlines():
        m = re.match(r"^(_Z\w+|k_\w+):\s*(;.*)?$", ln)
        if m:
            cur = subprocess.run(["c++filt", m.group(1)], stdout=subprocess.PIPE, text=True).stdout.strip()
            cur = re.sub(r"\(.*", "", cur).replace("void ", "").replace("zlng::", "")
            # k_rolz_parse_wg's sixth argument (kRingRule, round 6) defaults to false: such an instantiation IS the five-argument kernel of before
            cur = re.sub(r"^(k_rolz_parse_wg<\d+(?:, (?:true|false)){4}), false>$", r"\1>", cur)
            res[cur] = []
            continue
        if cur is None:
            continue
        s = ln.split(";")[0].strip()
        if s.startswith(".Lfunc_end"):
            cur = None
            continue
        if not s or s.startswith((".loc", ".file", ".cfi", ".p2align", ".size", ".type", ".section", ".text", ".globl", ".protected", ".weak", ".hidden")):
            continue
        s = re.sub(r"\.LBB\d+_", ".LBB_", s)         # block labels carry the function's index in the file: new instantiations shift it
        if s.startswith((".Lfunc_begin", ".Ltmp")):
            continue
        if s.startswith(".amdhsa_kernel "):           # carries the mangled name (a defaulted template argument more changes it, not the code)
            s = ".amdhsa_kernel"
        if s.startswith(".amdhsa_kernarg_size"):     # an argument appended BEHIND the ones a kernel reads changes this and nothing else
            continue
        res[cur].append(s)
    return {k: v for k, v in res.items() if v}


def stream_hash(lines):
    return hashlib.sha256("\n".join(lines).encode()).hexdigest()[:16]


def hipcc_version():
    out = subprocess.run([HIPCC, "--version"], stdout=subprocess.PIPE, text=True).stdout
    return " | ".join(ln.strip() for ln in out.splitlines()[:2])


def pin(commit, path):
    """Write {kernel: hash of its instruction stream} of every kernel of `commit` (tests/golden/kernel_isa_*.json)."""
    import json
    t = tempfile.mkdtemp()
    subprocess.check_call("git -C %s archive %s libzling_amd/csrc include | tar -x -C %s" % (ROOT, commit, t), shell=True)
    there = os.path.join(t, "libzling_amd", "csrc")
    out = {"commit": commit, "hipcc": hipcc_version(), "flags": "--offload-arch=gfx950 -O3 -std=c++17 -munsafe-fp-atomics", "kernels": {}}
    for name in sorted(f for f in os.listdir(there) if f.endswith(".hip")):
        for k, v in (kernels(there, name) or {}).items():
            out["kernels"]["%s:%s" % (name, k)] = {"instructions": len(v), "sha": stream_hash(v)}
    json.dump(out, open(path, "w"), indent=1, sort_keys=True)
    print("pinned %d kernels of %s -> %s" % (len(out["kernels"]), commit, path))


def main():
    if sys.argv[1] == "--pin":
        return pin(sys.argv[2], sys.argv[3])
    commit = sys.argv[1]
    t = tempfile.mkdtemp()
    subprocess.check_call("git -C %s archive %s libzling_amd/csrc include | tar -x -C %s" % (ROOT, commit, t), shell=True)
    here, there = os.path.join(ROOT, "libzling_amd", "csrc"), os.path.join(t, "libzling_amd", "csrc")
    same = diff = 0
    for name in sorted(f for f in os.listdir(here) if f.endswith(".hip")):
        a, b = kernels(here, name), (kernels(there, name) if os.path.exists(os.path.join(there, name)) else None)
        if a is None:
            print("%s: does not compile here" % name); continue
        if not a:
            print("%-16s (host code only: no kernels)" % name); continue
        for k in sorted(a):
            ha = hashlib.sha256("\n".join(a[k]).encode()).hexdigest()[:12]
            if b is None or k not in b:
                print("%-16s %-62s %6d instr  NEW (not in %s)" % (name, k[:62], len(a[k]), commit)); diff += 1; continue
            hb = hashlib.sha256("\n".join(b[k]).encode()).hexdigest()[:12]
            if ha == hb:
                same += 1
                print("%-16s %-62s %6d instr  identical" % (name, k[:62], len(a[k])))
            else:
                diff += 1
                print("%-16s %-62s %6d instr  DIFFERS (%d there)" % (name, k[:62], len(a[k]), len(b[k])))
    print("%d kernels identical to %s, %d differ or are new" % (same, commit, diff))


if __name__ == "__main__":
    main()
