"""CPU-only, test infrastructure checking test infrastructure: a wider-seed run of tests/test_oracle_hostile.py's differential -- the ORACLE's decoder
against the REAL reference (oracle/_ref, in a forked child: it may crash or hang) on hostile mutants of tests/hostile.py.
    python scripts/oracle_ref_hostile_soak.py <first seed> <seeds> <mutants per seed> [big]      e.g. 31000 2 1500
`big`: the streams that reach the decoder's big structures (hostile.big_mutants: wrapped rings, full blocks, far sources, crafted bodies of thousands of tokens)."""
import os, sys, hashlib, collections, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "oracle")); sys.path.insert(0, os.path.join(ROOT, "tests"))
import hostile
from oracle_py import Oracle, Reference
from test_oracle_hostile import ref_verdict
o = Oracle(); ref = Reference()
seed0, nseeds, count = int(sys.argv[1]), int(sys.argv[2]), int(sys.argv[3])
big = len(sys.argv) > 4 and sys.argv[4] == "big"
bad = []; total = 0; held = 0; dev = collections.Counter(); t0 = time.time()
for seed in range(seed0, seed0 + nseeds):
    for name, m, cap in (hostile.big_mutants if big else hostile.mutants)(o, seed, count):
        total += 1
        rc, y, flags = o.decode_ex(m, cap)
        if flags & 32:
            dev["self-copy (reference hangs)"] += 1; continue
        got = ref_verdict(ref, m, cap, timeout=10 if big else 3)
        if flags:
            dev["rule %d: %s" % (flags, "same verdict" if got[0] == rc else "reference %s" % (got[0],))] += 1; continue
        held += 1
        mine = (rc, int(y.size), hashlib.sha256(y.tobytes()).hexdigest())
        if tuple(got) != mine:
            bad.append((seed, name, mine[:2], tuple(got)[:2]))
print("hostile cpu soak%s seeds %d..%d: %d mutants, %d held against the real reference (verdict class, bytes, SHA-256), %d mismatches, %d under a documented rejection rule, %.0f s"
      % (" (BIG set)" if big else "", seed0, seed0 + nseeds - 1, total, held, len(bad), total - held, time.time() - t0))
for k, v in sorted(dev.items()): print("  %5d  %s" % (v, k))
for b in bad[:10]: print("  MISMATCH", b)
