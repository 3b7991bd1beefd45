#!/usr/bin/env python3
"""Could a block's parse be cut into segments that start from a dictionary warmed up on the W bytes in front of them (the only way to
shorten the 0.55 s a block's parse takes whatever else the chip does)?  CPU probe with the checker's parser: the second half of a
16 MiB block of the benchmark text parsed from position 8 MiB - W with an empty dictionary, compared token by token with the true parse.
Result (LABNOTES, round 6): token STARTS re-synchronise (99.0 % with W = 1 MiB), token WORDS do not -- 31 % of the second half's tokens
differ with W = 1 MiB, 24 % with W = 4 MiB, spread evenly over the segment: ROLZ's sources lie megabytes back (a rare context's ring
keeps old starts), so a truncated dictionary finds other matches or none.  The parse of a block stays one chain.

    python scripts/experiments/segment_parse_probe.py"""
import sys, ctypes as C, numpy as np, time
import os
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))), 'oracle'))
from oracle_py import Oracle, textgen, _ptr
o=Oracle(); L=o.lib
BLOCK=1<<24
x=textgen(BLOCK, 0)
def parse_from(start, level=0):
    s=L.zo_stream_new(level); L.zo_reset_buckets(s)
    ib=np.zeros(x.size+275,np.uint8); ib[:x.size]=x
    enc=C.c_int(start); rl=C.c_int(0); buf=np.empty(262144,np.uint32)
    toks=[]; 
    while enc.value < x.size:
        nt=L.zo_parse_subblock(s, level, _ptr(ib), x.size, C.byref(enc), buf.ctypes.data, C.byref(rl), 0)
        toks.append(buf[:nt].copy())
    L.zo_stream_free(s)
    t=np.concatenate(toks)
    sym=t&0xFFFF
    ln=np.where(sym<256,1,np.where(sym<258,2,sym.astype(np.int64)-258+4))
    pos=start+np.concatenate([[0],np.cumsum(ln)[:-1]])
    return t,pos,ln
t0=time.time()
T,P,Ln=parse_from(0)
print('true parse', T.size, 'tokens', '%.1fs'%(time.time()-t0))
true_at=dict()  # position -> token word
idx=np.searchsorted(P, np.arange(0))  # noop
Pset=set(P.tolist())
for B in (8<<20,):
  for W in (1<<16, 1<<18, 1<<20, 2<<20, 4<<20, 8<<20):
    t,p,ln=parse_from(B-W)
    sel=p>=B
    t,p=t[sel],p[sel]
    # align: tokens of the speculative parse whose start position is a true token start AND same word
    i=np.searchsorted(P,p)
    i=np.clip(i,0,P.size-1)
    samepos=P[i]==p
    sameword=samepos & (T[i]==t)
    # literals: aux is ctx -> same; matches: idx must equal
    first_sync=np.argmax(samepos) if samepos.any() else -1
    n=t.size
    print('W=%8d: %d tokens in 2nd half; same start %.3f%%; same start+word %.3f%%; mismatching tokens %d; last mismatch at +%d bytes of %d' % (W, n, 100*samepos.mean(), 100*sameword.mean(), int((~sameword).sum()), int(p[np.nonzero(~sameword)[0][-1]]-B) if (~sameword).any() else 0, x.size-B))
    # distribution of mismatches over the segment in 8 bins
    bins=np.histogram(p[~sameword], bins=8, range=(B,x.size))[0]
    print('   mismatches per eighth of the segment:', bins.tolist())
