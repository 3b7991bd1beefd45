"""The rank chain's CPU model (scripts/experiments/mtf_chain_model.c: the round-4 step of csrc/mtf_rank.hip executed lane by lane --
chain layout, EXEC lane 0 off, DPP write suppression, arithmetic shift of the ne-mask, head repairs, slow steps) against the
reference's literal ranks on literal streams the oracle's parse produces.  No GPU needed: it pins the ALGORITHM the kernel implements,
the GPU tests pin the kernel."""
import os
import subprocess

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_chain_model_matches_the_reference_ranks(tmp_path, oracle):
    exe = str(tmp_path / "chain_model")
    subprocess.check_call(["gcc", "-O2", "-o", exe, os.path.join(ROOT, "scripts", "experiments", "mtf_chain_model.c")])
    from oracle_py import textgen
    rng = np.random.Generator(np.random.PCG64(5))
    text = textgen(1_500_000, 31)
    # text, then text with a wide alphabet after every blank (ranks beyond the front: slow steps and couplings)
    wide = text.copy()
    blanks = np.flatnonzero(wide[:-1] == 32)
    wide[blanks + 1] = rng.integers(33, 255, blanks.size, dtype=np.uint8)
    x = np.concatenate([text, wide])
    tok, _ = oracle.parse_block(x, level=0)
    sym, aux = tok & 0xFFFF, tok >> 16
    m = (sym < 256) & (aux < 256)
    np.save(str(tmp_path / "ctx.npy"), aux[m].astype(np.uint8))
    np.save(str(tmp_path / "lit.npy"), sym[m].astype(np.uint8))
    np.array(oracle.mtfinit(), dtype=np.uint8).tofile(str(tmp_path / "mtfinit.bin"))
    p = subprocess.run([exe, "mtfinit.bin", "ctx.npy", "lit.npy"], cwd=str(tmp_path), stdout=subprocess.PIPE, check=True)
    out = p.stdout.decode()
    assert "ALLOWX = 0x3f3fff9fffffffff" in out                      # the constant csrc/mtf_rank.hip static_asserts
    assert " 0 mismatches" in out, out
    slow = int(out.split("slow steps ")[1].split(" ")[0])
    heads = int(out.split("head repairs ")[1].split(" ")[0])
    assert slow > 1000 and heads > 1000, out                          # both out-of-line paths were exercised
