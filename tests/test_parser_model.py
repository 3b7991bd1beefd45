"""scripts/experiments/wg_parser_model.c -- the CPU model of the window parser (csrc/rolz_wg.hip) -- in the three forms DESIGN.md quotes
numbers from: the kernel's own (floating windows, phase 1 inside the round), and round 5's two pipelined candidates (phase 1 of the
next window evaluated before the commits of this one, the tokens committed since taking part as "ghosts"; grid and floating
windows).  The model checks every token against the oracle's parse itself; this test only asks that every form is exact, at e0 and
e4, on text, incompressible bytes, runs and a skewed alphabet -- so that "the scheme is exact, it just does not pay" stays a checked
statement."""
import os
import subprocess

import numpy as np
import pytest

import corpus

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.fixture(scope="module")
def model(tmp_path_factory):
    d = tmp_path_factory.mktemp("wgm")
    exe = str(d / "wgm")
    subprocess.check_call(["gcc", "-O2", "-w", "-o", exe, os.path.join(ROOT, "scripts", "experiments", "wg_parser_model.c")])
    x = np.concatenate([corpus.get("text_700k")[:400_000], corpus.get("mixed_e4")[200_000:500_000], corpus.get("abc_1m")[:100_000],
                        corpus.get("skew_400k")[:150_000]])
    f = str(d / "in.bin")
    x.tofile(f)
    return exe, f


@pytest.mark.parametrize("level", [0, 4])
@pytest.mark.parametrize("mode", [[], ["1", "1", "128"], ["2", "1", "128", "160", "352"], ["1", "1", "16"]],
                         ids=["kernel", "grid-stale", "floating-stale", "grid-stale-small-ghost-cap"])
def test_window_parser_model_is_exact(model, level, mode):
    exe, f = model
    args = [exe, f, "256", str(level), "1", "99999999999", "1", "0", "64", "0", "0"] + mode
    p = subprocess.run(args, capture_output=True, text=True, timeout=600)
    assert p.returncode == 0 and ": exact |" in p.stdout, p.stdout[-600:] + p.stderr[-600:]


@pytest.mark.parametrize("level", [1, 2, 3, 4])
def test_ring_rule_is_exact_in_the_model(model, level):
    """Levels 1-4, the rule the kernel carries behind ZLNG_RING_FIX=1: a chain node whose ring slot a token of the same round has
    taken over ends the walk in front of it (the reference's chain-end test, src/libzling_lz.cpp:265) -- the match is the best over
    the nodes before it -- instead of making the token a serially replayed one.  Exact, and it must actually fire."""
    exe, f = model
    p = subprocess.run([exe, f, "256", str(level), "1", "99999999999", "1", "0", "64"], capture_output=True, text=True, timeout=600,
                       env=dict(os.environ, RING_FIX="1"))
    assert p.returncode == 0 and ": exact |" in p.stdout, p.stdout[-600:] + p.stderr[-600:]
    fired = int(p.stdout.split("ring_fix: ")[1].split()[0])
    assert fired > 100, p.stdout[-400:]


@pytest.mark.parametrize("level", [2, 4])
def test_lazy_rule_is_exact_in_the_model(model, level):
    """Model only (LABNOTES, round 5's last session): a token whose match length is no longer the speculation's has its lazy probes
    walked again under the new length (src/libzling_lz.cpp:291-316) instead of going hard.  Exact on top of the ring rule; it buys
    -2.6 % rounds at e4 for about one more probe walk per round, which is why no kernel carries it."""
    exe, f = model
    p = subprocess.run([exe, f, "256", str(level), "1", "99999999999", "1", "0", "64"], capture_output=True, text=True, timeout=600,
                       env=dict(os.environ, RING_FIX="1", LAZY_FIX="1"))
    assert p.returncode == 0 and ": exact |" in p.stdout, p.stdout[-600:] + p.stderr[-600:]
    fired = int(p.stdout.split("lazy_fix: ")[1].split()[0])
    assert fired > 100, p.stdout[-400:]
