"""Debug aid: first token where the GPU parse differs from the oracle, with the input position."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "oracle")); sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np
import libzling_amd as zl
import corpus
from oracle_py import Oracle
name = sys.argv[1]; lv = int(sys.argv[2]) if len(sys.argv) > 2 else 0
x = corpus.get(name)[: zl.BLOCK]
o = Oracle()
ot, oc = o.parse_block(x, lv, apply_mtf=True)
with zl.Stream(0, lv, True, 1) as s:
    s.encode(x)
    t, c = s.block_tokens(0)
n = min(t.size, ot.size)
bad = np.nonzero(t[:n] != ot[:n])[0]
print("ntok gpu %d oracle %d first diff %s" % (t.size, ot.size, bad[:1]))
if bad.size:
    i = int(bad[0])
    def adv(tok):
        sym = tok & 0xFFFF
        return np.where(sym >= 258, sym - 258 + 4, np.where(sym >= 256, 2, 1))
    pos = int(adv(ot[:i]).sum())
    print("token %d at input pos %d: gpu %#x oracle %#x" % (i, pos, t[i], ot[i]))
    print("context bytes:", bytes(x[max(0, pos - 12): pos + 24]))
    for j in range(max(0, i - 6), min(n, i + 3)):
        print("  tok %d gpu %#010x oracle %#010x" % (j, t[j], ot[j]))
