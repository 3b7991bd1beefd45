"""Bring-up / regression probe of the parser against the oracle, token by token:
    python scripts/wg_probe.py [small|text|real|all] [levels=0,4]
Prints one line per (input, level): OK or the first differing token with its input position."""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "oracle")); sys.path.insert(0, os.path.join(ROOT, "tests")); sys.path.insert(0, os.path.join(ROOT, "scripts"))
import numpy as np
import libzling_amd as zl
import corpus
from oracle_py import Oracle, textgen

what = sys.argv[1] if len(sys.argv) > 1 else "small"
levels = [int(v) for v in (sys.argv[2] if len(sys.argv) > 2 else "0,4").split(",")]
o = Oracle()


def adv(tok):
    sym = tok & 0xFFFF
    return np.where(sym >= 258, sym - 258 + 4, np.where(sym >= 256, 2, 1))


def check(name, x, lv):
    nb = max(1, (x.size + zl.BLOCK - 1) // zl.BLOCK)
    t0 = time.time()
    try:
        with zl.Stream(0, lv, True, nb) as s:
            z = s.encode(x)
            dt = time.time() - t0
            stages = dict(s.timings())
            ref = o.encode(x, lv)
            if z.size == ref.size and np.array_equal(z, ref):
                print("%-14s e%d OK   %9d B  %.2f s  parse %.1f ms" % (name, lv, x.size, dt, stages.get("rolz_parse", 0.0)), flush=True)
                return True
            for b in range(nb):
                xb = x[b * zl.BLOCK:(b + 1) * zl.BLOCK]
                ot, oc = o.parse_block(xb, lv, apply_mtf=False)
                t, c = s.block_tokens(b)
                # the device tokens are ranked; compare kinds, lengths and match fields (literal symbols differ by the rank stage)
                n = min(t.size, ot.size)
                lit_t = (t[:n] & 0xFFFF) < 256
                lit_o = (ot[:n] & 0xFFFF) < 256
                same = np.where(lit_t & lit_o, (t[:n] >> 16) == (ot[:n] >> 16), t[:n] == ot[:n])
                bad = np.nonzero(~same)[0]
                if bad.size or t.size != ot.size:
                    i = int(bad[0]) if bad.size else n
                    pos = int(adv(ot[:i]).sum())
                    print("%-14s e%d MISMATCH block %d: ntok gpu %d oracle %d, first diff token %d at input pos %d (pos %% 256 = %d): gpu %#010x oracle %#010x" % (
                        name, lv, b, t.size, ot.size, i, pos, pos % 256, t[i] if i < t.size else 0, ot[i] if i < ot.size else 0), flush=True)
                    print("     text:", bytes(xb[max(0, pos - 8): pos + 16]), flush=True)
                    for j in range(max(0, i - 3), min(n, i + 3)):
                        print("       tok %d gpu %#010x oracle %#010x" % (j, t[j], ot[j]))
                    print("     cuts gpu", c[:3].tolist(), "oracle", oc[:3])
                    return False
            print("%-14s e%d MISMATCH in bytes only (tokens agree)" % (name, lv), flush=True)
            return False
    except zl.ZlngError as e:
        print("%-14s e%d ERROR %s" % (name, lv, e), flush=True)
        return False


cases = []
if what in ("small", "all"):
    for nm in ("text_1000", "text_64k", "rand_4k", "zeros_20k", "abc_30k", "runs_ab", "bytes_ff", "skew_24k", "text_274", "text_280", "text_5"):
        cases.append((nm, corpus.get(nm)))
if what in ("text", "all"):
    for nm in ("text_700k", "rand_1m", "zeros_1m", "abc_1m", "skew_400k", "skew2_600k", "mixed_e4"):
        cases.append((nm, corpus.get(nm)))
    cases.append(("syn_20m", textgen(20 << 20, 5)))
if what in ("real", "all"):
    from real_text_soak import gather
    cases.append(("real_40m", gather(40 << 20)[0]))
bad = 0
for nm, x in cases:
    for lv in levels:
        bad += not check(nm, x, lv)
print("failures:", bad)
sys.exit(1 if bad else 0)
