"""Research tool (decode): distribution of match source distances and of the contexts matches occur in,
for one synthetic 16 MiB block (replays the decoder's ring on the oracle's token stream)."""
import sys, numpy as np
sys.path.insert(0, '/root/repo'); sys.path.insert(0, '/root/repo/oracle')
import oracle_py as op
O = op.Oracle()
x = op.textgen(1 << 24)
tok, cuts = O.parse_block(x, 0)
sym = (tok & 0xFFFF).tolist(); aux = (tok >> 16).tolist()
xb = x.tolist()
ring = [[0] * 4096 for _ in range(256)]
head = [0] * 256
pos = 0
d = []
ctxhits = {}
for i in range(len(sym)):
    s = sym[i]
    if pos < 2:
        pos += 1; continue
    c = xb[pos - 1]
    h = (head[c] + 1) & 4095; head[c] = h; ring[c][h] = pos
    if s < 256: pos += 1
    elif s < 258: pos += 2
    else:
        src = ring[c][(h - aux[i]) & 4095]
        d.append(pos - src)
        ctxhits[c] = ctxhits.get(c, 0) + 1
        pos += s - 258 + 4
d = np.array(d)
print('matches', d.size, 'tokens', len(sym))
for lim in (4096, 16384, 32768, 65536, 131072, 1 << 20):
    print(lim, round(float((d < lim).mean()), 3))
tot = sum(ctxhits.values())
top = sorted(ctxhits.items(), key=lambda kv: -kv[1])[:8]
print([(chr(c), round(n / tot, 3)) for c, n in top])
