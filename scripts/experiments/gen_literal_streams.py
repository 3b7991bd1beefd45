"""Research tool: per-context literal byte streams of N synthetic 16 MiB blocks, from the oracle's parse.
Writes ctx.npy / lit.npy / lit_<ctx>.bin into the current directory."""
import sys, numpy as np
sys.path.insert(0, '/root/repo')
from oracle import oracle_py as op
O = op.Oracle()
nblk = int(sys.argv[1]) if len(sys.argv) > 1 else 4
data = op.textgen(nblk * (1 << 24))
ctxs = []; lits = []
for b in range(nblk):
    tok, cuts = O.parse_block(data[b << 24:(b + 1) << 24], level=0)
    sym = tok & 0xFFFF; aux = tok >> 16
    m = (sym < 256) & (aux < 256)
    ctxs.append(aux[m].astype(np.uint8)); lits.append(sym[m].astype(np.uint8))
    print(b, tok.size, m.sum(), flush=True)
c = np.concatenate(ctxs); l = np.concatenate(lits)
np.save('ctx.npy', c); np.save('lit.npy', l)
for hot in (32, 101, 116):
    l[c == hot].tofile('lit_%d.bin' % hot)
h = np.bincount(c, minlength=256)
print('top ctx', [(int(i), int(h[i])) for i in np.argsort(-h)[:8]], l.size)
