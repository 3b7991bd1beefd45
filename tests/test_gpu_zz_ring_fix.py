"""Both settings of the parser's ring rule (ZLNG_RING_FIX, levels 1-4; csrc/rolz_wg.hip, src/libzling_lz.cpp:240-267: a chain node
whose ring slot a token of the same round has taken over ends the walk in front of it) produce the reference's bytes.

The library reads the variable once per process (zlng_api.hip), so every setting runs in a process of its own; whichever is the
default, the OTHER one keeps its coverage here: every golden stream at e1-e4 (small set byte for byte, large and multi-block ones
by size + SHA-256 from the real reference), two seeds of the structured fuzz (tests/test_gpu_fuzz.py) at random levels, and a
48 MiB three-block e4 stream against the oracle.

Written in round 6 while the GPU pool was closed to this repository (the rule itself was built in round 5 and is exact in the CPU
model, tests/test_parser_model.py): sorted behind the rest of the suite so that a first-run surprise cannot hide it under `-x`."""
import os
import subprocess
import sys

import pytest

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))

BODY = r'''
import json, os, sys
import numpy as np
ROOT = %r
for p in (ROOT, os.path.join(ROOT, "oracle"), os.path.join(ROOT, "tests")):
    sys.path.insert(0, p)
import corpus, test_gpu_fuzz
import libzling_amd as zl
from oracle_py import Oracle, textgen
assert zl.lib().zlng_device_count() >= 1
o = Oracle()
G = os.path.join(ROOT, "tests", "golden")
manifest = json.load(open(os.path.join(G, "manifest.json")))
n = 0
for name in sorted(corpus.SMALL):
    x = np.fromfile(os.path.join(G, name + ".bin"), dtype=np.uint8)
    for lv in (1, 2, 3, 4):
        want = np.fromfile(os.path.join(G, "%%s.e%%d.zlng" %% (name, lv)), dtype=np.uint8)
        assert np.array_equal(zl.encode(x, lv), want), (name, lv)
        n += 1
for key, meta in sorted(manifest["streams"].items()):
    name, lv = key[:-3], int(key[-1])
    if name in corpus.SMALL or lv == 0:
        continue
    z = zl.encode(corpus.get(name), lv)
    assert z.size == meta["size"] and corpus.sha(z) == meta["sha256"], key
    n += 1
for seed in (2, 5):
    rng = np.random.Generator(np.random.PCG64(1000 + seed))
    text = textgen(3_000_000, 200 + seed)
    for it in range(8):
        kind = (seed + it) %% 8
        m = int(rng.integers(1, 900_000)) if it else int(rng.integers(1, 600))
        x = np.ascontiguousarray(test_gpu_fuzz.make_input(rng, kind, m, text))
        lv = int(rng.integers(1, 5))
        assert np.array_equal(zl.encode(x, lv), o.encode(x, lv)), (seed, it, kind, lv)
        n += 1
x = textgen(3 * zl.BLOCK - 777, 61)
assert np.array_equal(zl.encode(x, 4), o.encode(x, 4))
print("ring_fix=%%s: %%d streams bit-exact" %% (os.environ.get("ZLNG_RING_FIX"), n + 1))
'''


@pytest.mark.parametrize("setting", ["0", "1"])
def test_both_settings_of_the_ring_rule_are_bit_exact(setting):
    r = subprocess.run([sys.executable, "-c", BODY % ROOT], env=dict(os.environ, ZLNG_RING_FIX=setting), stdout=subprocess.PIPE,
                       stderr=subprocess.PIPE, text=True, timeout=1200)
    assert r.returncode == 0 and "bit-exact" in r.stdout, (r.stdout[-500:], r.stderr[-3000:])
