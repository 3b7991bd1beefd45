"""Hostile .zlng streams for the decoder differentials (the reference's six guards: src/libzling.cpp:316, 327,
382, 392, 399, 407 and src/libzling_lz.cpp:363-373).

`mutants(oracle, seed, count)` yields (name, bytes, cap): valid reference streams (single- and multi-block, e0 and e4) damaged
by class --
  bit / byte / span     random damage anywhere
  header                encpos, rlen, olen of a sub-block at and around their limits (0, 1, the 262,144 / 393,216 limits +- 1,
                        2^24 +- 1, 2^31 - 1, 2^31, 2^32 - 1, the true value +- a little)
  flag                  a flag byte turned into 2..255, a terminator into a continuation and back
  table                 the 273 length-table bytes: all zero, one symbol, over- and under-subscribed sets, lengths over the
                        limit of alphabet 2, random nibbles, one nibble off
  trunc                 cuts at every structural offset (behind a flag, inside a header, inside the tables, inside the bits, in
                        front of the terminator) and bytes appended behind the end
  crafted               sub-blocks assembled from token lists with the oracle's own packer: ring indices that name slots no one
                        wrote, index 0, lengths that over- and undershoot encpos, match and word symbols in the two block-opening
                        entries, word symbols on empty MRU slots, rlen smaller and larger than the tokens, empty sub-blocks
used by tests/test_oracle_hostile.py (oracle against the REAL reference, CPU), tests/test_gpu_hostile.py (HIP path against the
oracle) and scripts/hostile_soak.py.  Everything is a pure function of the seed.
"""
import numpy as np

import corpus

CLASSES = ("bit", "byte", "span", "header", "flag", "table", "trunc", "crafted")
SPECIAL = (0, 1, 2, 272, 273, 274, 262143, 262144, 262145, 393215, 393216, 393217, (1 << 24) - 1, 1 << 24, (1 << 24) + 1,
           (1 << 31) - 1, 1 << 31, (1 << 32) - 1)


def walk(z):
    """[(flag_off, payload_off, encpos, rlen, olen)] of every sub-block and the offsets of the block terminators."""
    subs, ends, p = [], [], 0
    while p < len(z):
        if z[p] == 0:
            ends.append(p)
            p += 1
            continue
        e, r, o = (int.from_bytes(z[p + 1 + 4 * k: p + 5 + 4 * k].tobytes(), "big") for k in range(3))
        subs.append((p, p + 13, e, r, o))
        p += 13 + o
    return subs, ends


def bases(oracle):
    """[(name, stream, output capacity)]: small valid streams; the multi-block ones are blocks of different streams back to
    back -- a valid FORMAT (the decoder's literal tables simply carry over, so the later blocks decode to other bytes than
    their encoder saw, but every length still lands on its encpos)."""
    out = []
    enc = {}
    for name, lv in (("text_1000", 0), ("text_64k", 0), ("text_64k", 4), ("rand_4k", 0), ("abc_30k", 0), ("runs_ab", 2),
                     ("skew_24k", 0), ("zeros_20k", 3), ("text_280", 0), ("text_5", 0)):
        x = corpus.get(name)
        enc[(name, lv)] = (oracle.encode(x, lv), x.size)
        out.append(("%s.e%d" % (name, lv),) + enc[(name, lv)])
    x = corpus.get("text_700k")                                       # two sub-blocks in one block
    z = oracle.encode(x, 0)
    out.append(("text_700k.e0", z, x.size))
    for combo in ((("text_64k", 0), ("rand_4k", 0), ("text_64k", 4)), (("abc_30k", 0), ("text_1000", 0)),
                  (("skew_24k", 0), ("runs_ab", 2), ("text_280", 0), ("zeros_20k", 3))):
        zz = np.concatenate([enc[c][0] for c in combo])
        out.append(("+".join("%s.e%d" % c for c in combo), zz, sum(enc[c][1] for c in combo)))
    return out


def _be(v):
    return np.frombuffer(int(v & 0xFFFFFFFF).to_bytes(4, "big"), np.uint8)


def _sub(tokens, encpos, oracle, rlen=None, tables=None):
    """One framed sub-block from token words (sym | aux << 16) with Huffman tables built from their own histogram."""
    tok = np.asarray(tokens, dtype=np.uint32)
    f1, f2 = oracle.histogram(tok)
    l1, l2 = oracle.length_table(f1, 15), oracle.length_table(f2, 8)
    if tables is not None:
        l1, l2 = tables(l1, l2)
    pay = oracle.pack(tok, l1, l2)
    n16 = int(np.sum(1 + ((tok & 0xFFFF) >= 258))) if rlen is None else rlen
    return np.concatenate([np.array([1], np.uint8), _be(encpos), _be(n16), _be(pay.size), pay])


def crafted(oracle, rng):
    """(name, stream): one block assembled from tokens.  lit(r) = literal of rank r, W0 / W1 = word-MRU symbols, M(len, idx)."""
    def M(ln, idx):
        return (258 + ln - 4) | idx << 16
    W0, W1 = 256, 257
    head = [65, 66]                                                   # the block's two raw opening bytes
    body = [int(r) for r in rng.integers(0, 40, 30)]                  # thirty literals of small rank
    end = np.array([0], np.uint8)
    k = int(rng.integers(0, 18))
    far = int(rng.integers(1, 4096))
    ln = int(rng.integers(4, 260))
    if k == 0:      # a ring slot nobody wrote: offset 0, a copy from the block's start (legal, src/libzling_lz.cpp:388-399)
        t = head + body + [M(ln, far)]; return "never-written-slot", np.concatenate([_sub(t, 32 + ln, oracle), end])
    if k == 1:      # index 0: the slot the token itself has just written
        t = head + body + [M(ln, 0)]; return "index-0", np.concatenate([_sub(t, 32 + ln, oracle), end])
    if k == 2:      # the lengths overshoot encpos (src/libzling_lz.cpp:363-365)
        t = head + body + [M(ln, 1)]; return "overshoot", np.concatenate([_sub(t, 32 + ln - int(rng.integers(1, ln)), oracle), end])
    if k == 3:      # ... or stop short of it (:371-373)
        t = head + body + [M(ln, 1)]; return "undershoot", np.concatenate([_sub(t, 32 + ln + int(rng.integers(1, 300)), oracle), end])
    if k == 4:      # a match symbol in the FIRST opening entry
        t = [M(ln, far)] + body; return "opening-match-0", np.concatenate([_sub(t, 2 + 30, oracle), end])
    if k == 5:      # ... in the second
        t = [65, M(ln, far)] + body; return "opening-match-1", np.concatenate([_sub(t, 2 + 30, oracle), end])
    if k == 6:      # word symbols as opening entries: raw bytes 0 / 1 (the u16 entry truncated, :327-328)
        t = [W0, W1] + body + [W0, W1, M(ln, 1)]; return "opening-words", np.concatenate([_sub(t, 2 + 30 + 4 + ln, oracle), end])
    if k == 7:      # word symbols on empty MRU slots (zero words), then literals and matches over them
        t = head + [W0, W1, W0] + body + [W1, M(ln, 2), W0]; return "empty-mru", np.concatenate([_sub(t, 2 + 6 + 30 + 2 + ln + 2, oracle), end])
    if k == 8:      # rlen smaller than the tokens: the decode stops early and misses encpos
        t = head + body + [M(ln, 1)]; return "rlen-short", np.concatenate([_sub(t, 32 + ln, oracle, rlen=int(rng.integers(0, 33))), end])
    if k == 9:      # rlen larger: the reader runs into the zero padding behind the payload
        t = head + body + [M(ln, 1)]; return "rlen-long", np.concatenate([_sub(t, 32 + ln, oracle, rlen=34 + int(rng.integers(1, 50))), end])
    if k == 10:     # rlen = last entry is a match symbol without its index entry (src/libzling.cpp:398)
        t = head + body + [M(ln, 1)]; return "rlen-splits-match", np.concatenate([_sub(t, 32 + ln, oracle, rlen=33), end])
    if k == 11:     # empty sub-blocks around a real one; encpos repeats
        s0 = np.concatenate([np.array([1], np.uint8), _be(0), _be(0), _be(273), np.zeros(273, np.uint8)])
        t = head + body
        s2 = np.concatenate([np.array([1], np.uint8), _be(32), _be(0), _be(273), np.zeros(273, np.uint8)])
        return "empty-subblocks", np.concatenate([s0, _sub(t, 32, oracle), s2, end])
    if k == 12:     # the opening entries split over two sub-blocks (rlen 1, then the rest)
        return "opening-split", np.concatenate([_sub([65], 1, oracle), _sub([66] + body + [M(ln, 1)], 32 + ln, oracle), end])
    if k == 13:     # ... and the second opening entry, in the next sub-block, is a match
        return "opening-split-match", np.concatenate([_sub([65], 1, oracle), _sub([M(ln, 1)] + body, 2 + 30, oracle), end])
    if k == 14:     # encpos runs backwards between sub-blocks
        return "encpos-backwards", np.concatenate([_sub(head + body, 32, oracle), _sub(body, 20, oracle), end])
    if k == 15:     # encpos beyond the block size with lengths that really add up to it: 64 K matches of 259 over a 2-byte period
        nm = (16777216 - 2) // 259 + 2 + int(rng.integers(0, 3))
        t = head + [M(259, 1)] * nm
        return "encpos-over-block", np.concatenate([_sub(t, 2 + 259 * nm, oracle), end])
    if k == 16:     # over-subscribed alphabet 1 with a consistent bitstream for the encoder's (first-fit) codes
        def over(l1, l2):
            l1 = l1.copy(); l1[l1 == 0] = np.where(rng.random(int((l1 == 0).sum())) < 0.2, rng.integers(1, 16, int((l1 == 0).sum())), 0); return l1, l2
        t = head + body + [M(ln, 1)]; return "tables-oversubscribed", np.concatenate([_sub(t, 32 + ln, oracle, tables=over), end])
    # a long self-overlapping run and a far match, closed by a second block that opens with word symbols
    t = head + [M(259, 1)] * 40 + body + [M(ln, far)]
    b2 = _sub([W1, 7] + body, 32, oracle)
    return "two-blocks", np.concatenate([_sub(t, 2 + 259 * 40 + 30 + ln, oracle), end, b2, end])


def mutate(z, cls, rng, oracle):
    """One mutant of the valid stream z."""
    subs, ends = walk(z)
    m = z.copy()
    if cls == "bit":
        for _ in range(int(rng.integers(1, 4))):
            m[int(rng.integers(0, m.size))] ^= 1 << int(rng.integers(0, 8))
        return m
    if cls == "byte":
        for _ in range(int(rng.integers(1, 4))):
            m[int(rng.integers(0, m.size))] = int(rng.choice([0, 1, 255, int(rng.integers(0, 256))]))
        return m
    if cls == "span":
        a = int(rng.integers(0, m.size))
        n = int(min(m.size - a, rng.integers(2, 65)))
        how = int(rng.integers(0, 4))
        if how == 0: m[a:a + n] = rng.integers(0, 256, n)
        elif how == 1: m[a:a + n] = 0
        elif how == 2: m[a:a + n] = 255
        else:
            b = int(rng.integers(0, m.size - n + 1)); m[a:a + n] = z[b:b + n]
        return m
    f, pay, e, r, o = subs[int(rng.integers(0, len(subs)))]
    if cls == "header":
        field = int(rng.integers(0, 3))
        true = (e, r, o)[field]
        v = int(rng.choice(SPECIAL)) if rng.random() < 0.6 else true + int(rng.integers(-3, 4)) or true + 1
        m[f + 1 + 4 * field: f + 5 + 4 * field] = _be(v)
        return m
    if cls == "flag":
        how = int(rng.integers(0, 3))
        if how == 0: m[f] = int(rng.integers(2, 256))
        elif how == 1 and ends: m[ends[int(rng.integers(0, len(ends)))]] = int(rng.choice([1, 2, 255]))
        else: m[f] = 0
        return m
    if cls == "table":
        t1, t2 = pay, pay + 257
        how = int(rng.integers(0, 10))
        if how == 0: m[t1:t1 + 257] = 0
        elif how == 1: m[t2:t2 + 16] = 0
        elif how == 2: m[t1:t1 + 257] = 0; m[t1 + int(rng.integers(0, 257))] = int(rng.choice([0x10, 0x01, 0xF0, 0x11]))
        elif how == 3: m[t1:t1 + 257] = int(rng.choice([0x11, 0xFF, 0x88, 0x2F]))               # every symbol the same length(s)
        elif how == 4: m[t2:t2 + 16] = int(rng.choice([0x11, 0xFF, 0x88, 0x99, 0x9F]))          # alphabet 2 incl. lengths over its limit of 8
        elif how == 5: m[t1:t1 + 273] = rng.integers(0, 256, 273)
        elif how == 6:                                                # one nibble off: over- or under-subscribed by one code
            a = t1 + int(rng.integers(0, 273)); m[a] = (int(m[a]) + int(rng.choice([1, 15, 16, 240]))) & 255
        elif how == 7:                                                # a few short codes added to a complete set: over-subscribed
            for a in rng.integers(t1, t1 + 257, 6): m[a] |= int(rng.choice([0x10, 0x01, 0x20, 0x03]))
        elif how == 8:                                                # codes removed: holes in the decode table
            for a in rng.integers(t1, t1 + 273, 12): m[a] &= int(rng.choice([0x0F, 0xF0]))
        else: m[t1:t1 + 257] = np.where(rng.random(257) < 0.5, m[t1:t1 + 257], 0)
        return m
    if cls == "trunc":
        how = int(rng.integers(0, 9))
        cuts = [f + 1, f + 1 + int(rng.integers(1, 12)), pay, pay + int(rng.integers(1, 273)), pay + 273 + int(rng.integers(0, max(1, o - 273))),
                pay + o, m.size - 1, int(rng.integers(0, m.size))]
        if how < 8:
            return m[: min(cuts[how], m.size)].copy()
        tail = [np.array([int(rng.integers(2, 256))], np.uint8), np.array([0], np.uint8), np.array([1, 0, 0], np.uint8),
                rng.integers(0, 256, int(rng.integers(1, 40))).astype(np.uint8)][int(rng.integers(0, 4))]
        return np.concatenate([m, tail])
    raise ValueError(cls)


def mutants(oracle, seed, count, classes=CLASSES):
    """count x (name, bytes, output capacity), class by class in turn."""
    rng = np.random.Generator(np.random.PCG64(seed))
    bs = bases(oracle)
    for i in range(count):
        cls = classes[i % len(classes)]
        if cls == "crafted":
            name, m = crafted(oracle, rng)
            if rng.random() < 0.3:                                    # a valid stream in front: the good block must be reported first
                _, z0, cap0 = bs[int(rng.integers(0, len(bs)))]
                yield "crafted:%s behind a good stream" % name, np.concatenate([z0, m]), cap0 + (3 << 24)
            else:
                yield "crafted:" + name, m, 3 << 24
            continue
        bname, z, cap = bs[int(rng.integers(0, len(bs)))]
        yield "%s:%s" % (cls, bname), mutate(z, cls, rng, oracle), cap + (2 << 24)


# ---------------------------------------------------------------------------------------------------------------------------------
# Streams that reach the decoder's BIG structures ("what's weak" 1b): a ring that has wrapped (more than 4,096 token
# starts in one context), a full 16 MiB block, match sources further back than the replay kernel's 64 KiB LDS window and copies that
# straddle its wrap, generic-level multi-sub-block streams, crafted bodies of thousands of tokens, a block with more entries than a
# block can hold.  `big_mutants` damages them with the same classes as above, weighted away from the classes the decoder's own
# rejection rules decide (trunc, the header values >= 2^31): the share of mutants decided by a ZO_DEV_* rule is reported by the
# tests and kept under 35 %.

def M(ln, idx):
    return (258 + ln - 4) | idx << 16


def tok_bytes(tokens):
    """Decoded length of a token list BEHIND the block's two raw opening entries (lengths are in the tokens, not in the data)."""
    t = np.asarray(tokens, dtype=np.uint32) & 0xFFFF
    return int(np.sum(np.where(t < 256, 1, np.where(t < 258, 2, t.astype(np.int64) - 258 + 4))))


def frame_block(subs_tokens, oracle, opening=2):
    """One block from a list of per-sub-block token lists; the first list starts with the block's two raw opening entries."""
    parts, pos = [], 0
    for k, t in enumerate(subs_tokens):
        t = [int(v) for v in t]
        if k == 0:
            pos += opening + tok_bytes(t[opening:])
        else:
            pos += tok_bytes(t)
        parts.append(_sub(t, pos, oracle))
    return np.concatenate(parts + [np.array([0], np.uint8)]), pos


def big_crafted(oracle, rng, kind=None):
    """(name, stream, decoded size) -- VALID streams (every length lands on its encpos) built from thousands of tokens."""
    k = int(rng.integers(0, 5)) if kind is None else kind
    head = [65, 32]                                                   # opens in the blank's context
    if k == 0:
        # ring wrap: runs of rank-0 literals keep the context at the table's front symbol (the blank, src/tables/gen.py: mtfinit[0] = 32),
        # so every token start lands in ONE bucket; matches name slots all over the ring -- before the 4,096th insert an index beyond
        # the insert count names a slot nobody wrote (offset 0, a legal copy from the block's start), behind it every index is live
        n = int(rng.integers(4300, 9000))
        t = list(head)
        for i in range(n):
            r = rng.random()
            if r < 0.90: t.append(0)
            elif r < 0.94: t.append(int(rng.integers(2, 6)))          # another symbol: the context leaves the blank and comes back at once
                                                                      # (rank 0 there is the blank too; rank 1 would move the blank off the front)
            else: t.append(M(int(rng.integers(4, 40)), int(rng.choice([1, 2, 63, 64, 4094, 4095, int(rng.integers(1, 4096))]))))
        z, size = frame_block([t], oracle)
        return "big:ring-wrap", z, size
    if k == 1:
        # LDS window: a 2-byte period pushed across 64 KiB and 128 KiB by maximal matches, copies that start a few bytes in front of
        # every 64 KiB boundary, then matches whose ring slot was written more than 128 KiB earlier (index = inserts since then)
        t = list(head) + [7, 9, 11]
        pos = 2 + 3
        far_at = len(t)                                               # a token start early in the block, in the blank's bucket
        while pos < 200000 + int(rng.integers(0, 70000)):
            to_edge = 65536 - pos % 65536
            if 4 <= to_edge - 3 <= 259 and rng.random() < 0.9:
                ln = to_edge - int(rng.integers(1, 4)); t.append(M(max(4, ln), 1)); pos += max(4, ln)       # stops just short of the boundary
                t.append(M(259, 1)); pos += 259                                                             # ... and the next copy straddles it
            else:
                t.append(M(259, 1)); pos += 259
        inserts = len(t) - far_at                                     # every token since was a start in the same bucket (period of blanks)
        for _ in range(20):
            idx = min(4095, max(1, inserts - int(rng.integers(0, 4))))
            t.append(M(int(rng.integers(4, 260)), idx)); inserts += 1
        z, size = frame_block([t], oracle)
        return "big:lds-window", z, size
    if k == 2:
        # three sub-blocks of random tokens: literals of every rank, word symbols, matches of every length and index
        subs = []
        for s in range(3):
            n = int(rng.integers(20000, 60000))
            kind_ = rng.random(n)
            toks = np.where(kind_ < 0.55, rng.integers(0, 256, n),
                            np.where(kind_ < 0.62, rng.integers(256, 258, n),
                                     (258 + rng.integers(0, 256, n)) | (rng.integers(1, 4096, n) << 16))).astype(np.uint32)
            subs.append(([65, 66] if s == 0 else []) + [int(v) for v in toks])
        z, size = frame_block(subs, oracle)
        if size > 16777216:
            return big_crafted(oracle, rng, 0)
        return "big:random-tokens-3-subblocks", z, size
    if k == 3:
        # two blocks: the literal tables carry over (rank r names another byte in block 2), the ring does not
        b1, s1 = frame_block([list(head) + [int(v) for v in rng.integers(0, 50, 5000)] + [M(200, 5)] * 300], oracle)
        b2, s2 = frame_block([[66, 67] + [int(v) for v in rng.integers(0, 50, 5000)] + [M(100, 4000), M(259, 1)] * 200], oracle)
        return "big:two-blocks-carry", np.concatenate([b1, b2]), s1 + s2
    # k == 4: more u16 entries than a block can hold: 64 sub-blocks of 262,144 blanks fill the 16 MiB exactly, a 65th sub-block
    # crosses the limit.  With rlen <= 64 it is an ordinary sub-block whose lengths overshoot (lzdecode failed / or bad code1 with
    # an empty table); beyond that the decoder rejects at its header (ZO_DEV_ENTRIES)
    full = [list(head) + [0] * 262142] + [[0] * 262144 for _ in range(63)]
    z, size = frame_block(full, oracle)
    assert size == 16777216
    extra = int(rng.choice([1, 64, 65, 100, 262144]))
    tail = _sub([0] * extra, 16777216, oracle)
    if rng.random() < 0.5:
        tail[13: 13 + 257] = 0                                        # no code for any symbol: "bad code1" wherever the stream is read
    return "big:too-many-entries(+%d)" % extra, np.concatenate([z[:-1], tail, np.array([0], np.uint8)]), 16777216


_BIG = {}


def big_bases(oracle):
    """[(name, stream, output capacity)], built once per process: encoder-made streams at full size."""
    if _BIG:
        return _BIG["v"]
    from oracle_py import textgen
    out = []
    x = textgen(16777216 + 700000, 3)                                  # a FULL block and a second one: rings wrapped thousands of times
    out.append(("text_16m+.e0", oracle.encode(x, 0), x.size))
    x = textgen(3000000, 5)                                            # generic level, several sub-blocks
    out.append(("text_3m.e4", oracle.encode(x, 4), x.size))
    # far sources: a rare context byte (0x01) in front of a phrase, 150-700 KB apart: its ring keeps the old starts, the matches
    # reach back beyond the 64 KiB window and beyond 128 KiB; two blocks of it
    rng = np.random.Generator(np.random.PCG64(99))
    y = textgen(2 * 16777216 - 3000000, 7)
    phrase = np.frombuffer(b"\x01the quick brown fox jumps over the lazy dog 0123456789 ABCDEFGHIJKLMNOPQRSTUVWXYZ", np.uint8)
    p = 1000
    while p + phrase.size < y.size:
        y[p: p + phrase.size] = phrase
        p += int(rng.integers(150000, 700000))
    out.append(("far_sources_2blk.e0", oracle.encode(y, 0), y.size))
    _BIG["v"] = out
    return out


BIG_CLASSES = ("bit", "byte", "span", "header", "flag", "table", "crafted", "bit", "span", "table", "crafted", "trunc")


def big_mutants(oracle, seed, count):
    """count x (name, bytes, output capacity): big crafted streams as they are (they must decode) and damaged, big encoder-made
    streams damaged.  A pure function of the seed."""
    rng = np.random.Generator(np.random.PCG64(seed))
    bs = big_bases(oracle)
    for i in range(count):
        cls = BIG_CLASSES[i % len(BIG_CLASSES)]
        if cls == "crafted":
            name, z, size = big_crafted(oracle, rng, i // len(BIG_CLASSES) % 5 if i < 5 * len(BIG_CLASSES) else None)
            if rng.random() < 0.5 or name.startswith("big:too-many"):
                yield name, z, size + (1 << 20)
            else:
                c2 = ("bit", "byte", "span", "header", "table")[int(rng.integers(0, 5))]
                yield "%s:%s" % (c2, name), mutate(z, c2, rng, oracle), size + (2 << 24)
            continue
        bname, z, cap = bs[int(rng.integers(0, len(bs)))]
        m = mutate(z, cls, rng, oracle)
        if cls == "header" and rng.random() < 0.7:                     # mostly near-true values: the damage is met deep inside, not by a size rule
            subs, _ = walk(z)
            f, pay, e, r, o = subs[int(rng.integers(0, len(subs)))]
            m = z.copy()
            field = int(rng.integers(0, 2))
            m[f + 1 + 4 * field: f + 5 + 4 * field] = _be((e, r)[field] + int(rng.choice([-2, -1, 1, 2, 259, -259])))
        yield "%s:%s" % (cls, bname), m, cap + (2 << 24)
