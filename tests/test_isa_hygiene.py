"""Properties of the BUILT gfx950 code that the sources rely on but the compiler does not promise (no GPU needed: hipcc
cross-compiles to assembly here).

* M0 is a reserved register: an inline-asm statement that writes it cannot be honoured by the register allocator (round 3 wrote M0
  for v_writelane and argued from an ISA inspection that nothing else used it).  Round 4 removed every M0 write from the rank
  stage; this test keeps it that way after a toolchain or source change.
* The tile statements of k_mtf_chain switch EXEC lane 0 off and must switch every lane on again on every way out.
"""
import os
import re
import shutil
import subprocess

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
CSRC = os.path.join(ROOT, "libzling_amd", "csrc")
HIPCC = shutil.which("hipcc") or "/opt/rocm/bin/hipcc"


def isa_of(tmp_path, name):
    out = str(tmp_path / (name + ".s"))
    subprocess.check_call([HIPCC, "--offload-arch=gfx950", "-O3", "-std=c++17", "-S", "--cuda-device-only", "-Wno-everything",
                           "-o", out, os.path.join(CSRC, name + ".hip")])
    with open(out) as f:
        return [ln.split(";")[0].strip() for ln in f]


@pytest.mark.skipif(not os.path.exists(HIPCC), reason="hipcc not installed")
def test_rank_stage_never_touches_m0(tmp_path):
    lines = isa_of(tmp_path, "mtf_rank")
    hits = [ln for ln in lines if re.search(r"\bm0\b", ln)]
    assert hits == [], hits[:5]


@pytest.mark.skipif(not os.path.exists(HIPCC), reason="hipcc not installed")
def test_chain_tile_restores_exec(tmp_path):
    lines = isa_of(tmp_path, "mtf_rank")
    off = [i for i, ln in enumerate(lines) if ln == "s_mov_b64 exec, -2"]
    on = [i for i, ln in enumerate(lines) if ln == "s_mov_b64 exec, -1"]
    assert off and len(off) == len(on)
    for a, b in zip(off, on):                 # each statement: lane 0 off ... every exit runs through the one restore at its end
        assert a < b
        body = lines[a + 1: b]
        assert not any(ln.startswith("s_endpgm") or ln.startswith("s_setpc_b64 s[30:31]") for ln in body)
        assert not any("s_mov_b64 exec" in ln for ln in body)
