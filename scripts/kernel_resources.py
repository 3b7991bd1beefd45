#!/usr/bin/env python3
"""Register / LDS / occupancy figures of every kernel of a HIP source as the compiler reports them (hipcc -Rpass-analysis=
kernel-resource-usage), for one or two trees -- no GPU needed.

    python scripts/kernel_resources.py rolz_wg.hip                 # this tree
    python scripts/kernel_resources.py rolz_wg.hip 1a2de5a~1       # ... against that commit (git archive into /tmp)

Used for VERDICT r5 item 2: what the ring rule's plumbing costs the parser's e1-e4 instantiation while the rule is off."""
import os
import re
import subprocess
import sys
import tempfile

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def usage(csrc, name):
    with tempfile.TemporaryDirectory() as t:
        r = subprocess.run(["/opt/rocm/bin/hipcc", "--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-munsafe-fp-atomics", "-c",
                            os.path.join(csrc, name), "-o", os.path.join(t, "x.o"), "-Rpass-analysis=kernel-resource-usage"],
                           stderr=subprocess.PIPE, text=True, cwd=csrc)
    out, cur = {}, None
    for line in r.stderr.splitlines():
        m = re.search(r"Function Name: (\S+)", line)
        if m:
            cur = subprocess.run(["c++filt", m.group(1)], stdout=subprocess.PIPE, text=True).stdout.strip()
            cur = re.sub(r"\(.*", "", cur).replace("void ", "").replace("zlng::", "")
            out[cur] = {}
            continue
        m = re.search(r"remark:\s+([A-Za-z ]+?)(?: \[[^\]]*\])?: (\S+)", line)
        if m and cur:
            out[cur][m.group(1).strip()] = m.group(2)
    return out


def main():
    name = sys.argv[1]
    here = usage(os.path.join(ROOT, "libzling_amd", "csrc"), name)
    other = None
    if len(sys.argv) > 2:
        t = tempfile.mkdtemp()
        subprocess.check_call("git -C %s archive %s libzling_amd/csrc include | tar -x -C %s" % (ROOT, sys.argv[2], t), shell=True)
        other = usage(os.path.join(t, "libzling_amd", "csrc"), name)
    cols = ("VGPRs", "TotalSGPRs", "SGPRs Spill", "VGPRs Spill", "ScratchSize", "Occupancy", "LDS Size")
    print("%-58s %s" % ("kernel (%s)" % name, "  ".join("%11s" % c for c in cols)))
    for k in sorted(here):
        row = here[k]
        def cell(c):
            v = next((row[x] for x in row if x.startswith(c)), "?")
            if other and k in other:
                w = next((other[k][x] for x in other[k] if x.startswith(c)), "?")
                if w != v:
                    return "%s<-%s" % (v, w)
            return v
        print("%-58s %s" % (k[:58], "  ".join("%11s" % cell(c) for c in cols)))
    if other:
        print("(a<-b: this tree <- %s)" % sys.argv[2])


if __name__ == "__main__":
    main()
