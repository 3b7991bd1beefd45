#!/usr/bin/env python3
"""Hostile-stream soak of the HIP decoder against the oracle (the long form of tests/test_gpu_hostile.py):

    python scripts/hostile_soak.py [count=40000] [seed=1]

Prints the verdict classes reached and every disagreement, grouped; exit status 1 if there is one.  The oracle's own verdicts are
held against the real reference by tests/test_oracle_hostile.py on the CPU side (same generator, tests/hostile.py)."""
import collections
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (ROOT, os.path.join(ROOT, "oracle"), os.path.join(ROOT, "tests")):
    sys.path.insert(0, p)
import numpy as np  # noqa: E402

import corpus  # noqa: E402
import hostile  # noqa: E402
import libzling_amd as zl  # noqa: E402
from oracle_py import Oracle  # noqa: E402
from test_gpu_hostile import CODE, gpu_verdict  # noqa: E402


def main():
    count = int(sys.argv[1]) if len(sys.argv) > 1 else 40000
    seed = int(sys.argv[2]) if len(sys.argv) > 2 else 1
    o = Oracle()
    good_x = corpus.get("text_64k")
    good_z = o.encode(good_x, 0)
    seen, bad = collections.Counter(), collections.Counter()
    first = {}
    t0 = time.time()
    with zl.Stream(0, 0, False, 4) as s:
        init, _ = s.get_state()
        for i, (name, m, cap) in enumerate(hostile.mutants(o, seed, count)):
            rc, y, flags = o.decode_ex(m, cap)
            code, got = gpu_verdict(zl, s, init, m, cap)
            seen[(name.split(":")[0], code)] += 1
            if code != CODE[rc] or got.size != y.size or not np.array_equal(got, y):
                k = (name.split(":")[0], "oracle %d" % rc, "gpu %d" % code, "bytes %s" % ("same" if got.size == y.size and np.array_equal(got, y) else "%d vs %d" % (y.size, got.size)))
                bad[k] += 1
                first.setdefault(k, (i, name, flags))
            if i % 1000 == 999:
                code, got = gpu_verdict(zl, s, init, good_z, good_x.size)
                assert code == 0 and np.array_equal(got, good_x), "context broken after mutant %d (%s)" % (i, name)
    print("%d mutants (seed %d) in %.1f s" % (count, seed, time.time() - t0))
    print("verdicts reached (class, ZLNG code): " + ", ".join("%s %d: %d" % (k[0], k[1], v) for k, v in sorted(seen.items())))
    for k, v in sorted(bad.items()):
        print("DISAGREE x%d: %s  first: %s" % (v, k, first[k]))
    print("disagreements: %d" % sum(bad.values()))
    return 1 if bad else 0


if __name__ == "__main__":
    sys.exit(main())
