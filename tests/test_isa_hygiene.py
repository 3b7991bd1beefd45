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


# ---- gfx940 / gfx950: a VALU that writes an SGPR or VCC and a VALU that reads it need two wait states between them ----------------
# (LLVM GCNHazardRecognizer, checkVALUHazards under hasVDecCoExecHazard: VALUWriteSGPRVALUReadWaitstates = 2; hipcc pads its own
# code -- `v_cmp; s_nop 0 + one more instruction; v_cndmask` on gfx950, nothing on gfx90a -- but never looks inside an asm statement.)
_SREG = re.compile(r"\b(vcc(?:_lo|_hi)?\b|s\[(\d+):(\d+)\]|s(\d+)\b)")


def _sregs(text):
    """Scalar registers named in an operand string, as a set of ints (vcc = {1000, 1001})."""
    out = set()
    for m in _SREG.finditer(text):
        if m.group(1).startswith("vcc"):
            out |= {1000} if m.group(1) == "vcc_lo" else ({1001} if m.group(1) == "vcc_hi" else {1000, 1001})
        elif m.group(2) is not None:
            out |= set(range(int(m.group(2)), int(m.group(3)) + 1))
        else:
            out.add(int(m.group(4)))
    return out


def valu_sgpr_hazards(lines, need=2):
    """[(line number, writer, reader)] of VALU-writes-SGPR -> VALU-reads pairs fewer than `need` wait states apart, scanning the
    assembly in text order (a taken branch only lengthens the distance)."""
    recent = []                                       # [(registers written, wait states since, text, line)]
    bad = []
    for ln, text in enumerate(lines):
        if not text or text.endswith(":") or text.startswith(".") or text.startswith(";"):
            continue
        parts = text.split(None, 1)
        op, args = parts[0], (parts[1] if len(parts) > 1 else "")
        if not re.match(r"^[svdgb]_|^global_|^buffer_|^ds_|^flat_|^scratch_", op):
            continue
        ops = [a.strip() for a in args.split(",")]
        if op.startswith("v_"):
            carry = "_co_" in op or op.startswith("v_div_scale")
            ndst = 2 if carry else 1
            reads = _sregs(", ".join(ops[ndst:]))
            if op.startswith("v_div_fmas"):
                reads |= {1000, 1001}
            for regs, dist, wtext, wln in recent:
                if dist < need and regs & reads:
                    bad.append((ln, wtext, text))
            writes = _sregs(", ".join(ops[:ndst]))
            for r in recent:
                r[1] += 1
            if writes:
                recent.append([writes, 0, text, ln])
        else:
            step = 1
            m = re.match(r"s_nop\s+(\d+)", text)
            if m:
                step = int(m.group(1)) + 1
            for r in recent:
                r[1] += step
        recent = [r for r in recent if r[1] < need]
    return bad


def test_the_hazard_scanner_sees_what_it_should():
    assert valu_sgpr_hazards(["v_cmp_ne_u32_e32 vcc, v1, v2", "v_cndmask_b32_e32 v0, v1, v2, vcc"])
    assert valu_sgpr_hazards(["v_cmp_ne_u32_sdwa vcc, v1, v2 src0_sel:BYTE_0 src1_sel:DWORD", "s_cbranch_scc0 .L1",
                              "v_cndmask_b32_dpp v0, v0, v0, vcc wave_shr:1 row_mask:0xf bank_mask:0xf"])      # round 4's step: one state
    assert not valu_sgpr_hazards(["v_cmp_ne_u32_sdwa vcc, v1, v2 src0_sel:BYTE_0 src1_sel:DWORD", "s_cbranch_scc0 .L1", "s_nop 0",
                                  "v_cndmask_b32_dpp v0, v0, v0, vcc wave_shr:1 row_mask:0xf bank_mask:0xf"])
    assert valu_sgpr_hazards(["v_readlane_b32 s5, v1, 3", "v_mov_b32_e32 v2, s5"])
    assert not valu_sgpr_hazards(["v_readlane_b32 s5, v1, 3", "s_nop 1", "v_mov_b32_e32 v2, s5"])
    assert valu_sgpr_hazards(["v_cmp_eq_u32_e64 s[4:5], v1, v2", "s_mov_b32 s9, 0", "v_cndmask_b32_e64 v0, v1, v2, s[4:5]"])
    assert not valu_sgpr_hazards(["v_cmp_eq_u32_e64 s[4:5], v1, v2", "v_cndmask_b32_e64 v0, v1, v2, s[6:7]"])
    assert not valu_sgpr_hazards(["s_ashr_i64 vcc, vcc, 1", "v_cndmask_b32_e32 v0, v1, v2, vcc"])               # a scalar writer owes nothing


@pytest.mark.skipif(not os.path.exists(HIPCC), reason="hipcc not installed")
@pytest.mark.parametrize("name", ["mtf_rank", "decode", "rolz_wg", "rolz_parse", "huffman"])
def test_no_valu_reads_an_sgpr_a_valu_wrote_less_than_two_wait_states_ago(tmp_path, name):
    bad = valu_sgpr_hazards(isa_of(tmp_path, name))
    assert not bad, (len(bad), bad[:8])


# ---- the kernels at HEAD against the kernels that last ran on a GPU -------------------------------------------------------------
# tests/golden/kernel_isa_last_gpu_run.json (scripts/isa_diff.py --pin b703640): the instruction stream of every kernel of b703640, the
# last commit before the GPU pool closed to this repository in the middle of round 5 (its kernel sources are those of d0b3819, 06:31 that
# day; the GPU suite -- 134 passed, profiles/r05_d_gpu_tests.txt -- the hostile soak and the config-4 runs of that morning were taken on
# them or on the working tree of the minutes before; everything the round-5 review lists as "never executed on a GPU" came after).  What
# CAN be shown without a GPU is that the machine code the default configuration executes is still that code, instruction for
# instruction: EVERY kernel of that commit -- all parser instantiations at every level, both rank kernels, the Huffman kernels, the three
# decode kernels -- compiles to the same stream at HEAD (labels renumbered, the mangled name and .amdhsa_kernarg_size apart), and the only
# kernels HEAD has on top are the six instantiations that carry the ring rule (kRingRule = true: launched only under ZLNG_RING_FIX=1,
# never run).  Two things in the sources exist for this: ParseArgs::ring_fix sits LAST in the argument block (the other fields keep
# their kernarg offsets), and the rule is a template argument, not a run-time branch, so none of its code or registers is in the default
# instantiations.
RING_RULE_PARSER = {"rolz_wg.hip:k_rolz_parse_wg<%d, false, %s, false, false, true>" % (nw, p) for nw in (2, 4, 8) for p in ("false", "true")}


@pytest.mark.skipif(not os.path.exists(HIPCC), reason="hipcc not installed")
def test_every_kernel_of_the_last_gpu_run_is_instruction_identical_at_head():
    import json
    import sys
    sys.path.insert(0, os.path.join(ROOT, "scripts"))
    import isa_diff
    pins = json.load(open(os.path.join(ROOT, "tests", "golden", "kernel_isa_last_gpu_run.json")))
    if pins["hipcc"] != isa_diff.hipcc_version():
        pytest.skip("another hipcc than the one the pins were taken with: %s" % isa_diff.hipcc_version())
    here = {}
    for name in sorted(f for f in os.listdir(CSRC) if f.endswith(".hip")):
        for k, v in (isa_diff.kernels(CSRC, name) or {}).items():
            here["%s:%s" % (name, k)] = isa_diff.stream_hash(v)
    assert len(pins["kernels"]) == 37
    differ = sorted(k for k in pins["kernels"] if here.get(k) != pins["kernels"][k]["sha"])
    assert differ == [], differ
    assert set(here) - set(pins["kernels"]) == RING_RULE_PARSER, sorted(set(here) ^ set(pins["kernels"]))
