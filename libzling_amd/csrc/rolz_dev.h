// rolz_dev.h -- device helpers shared by the ROLZ parser kernels (rolz_parse.hip, rolz_pipe.hip): the exact
// MatchAndUpdate / MatchLazy restatements, the per-position speculation of a window, and the small lane utilities.
// Internal to libzlng_hip.so.
#pragma once
#include "zlng_common.h"
#include "zlng_kernels.h"

namespace zlng {

// ------------------------------------------------------------------------------ helpers
__device__ __forceinline__ uint32_t ld32u(const uint8_t* p) { uint32_t v; __builtin_memcpy(&v, p, 4); return v; }

// HashContext, src/libzling_lz.cpp:55-57
__device__ __forceinline__ uint32_t hash4(const uint8_t* p) {
    uint32_t w = ld32u(p);
    return w + ((w >> 16) & 0xFF) * 137u + (w >> 24) * 13337u;
}

// GetCommonLength, src/libzling_lz.cpp:66-89: 0 unless 4 bytes agree, else byte-wise LCP capped at 259.
__device__ __forceinline__ int common_len(const uint8_t* a, const uint8_t* b) {
    if (ld32u(a) != ld32u(b)) return 0;
    int n = 4;
    while (n + 4 <= kMatchMax) {
        uint32_t x = ld32u(a + n) ^ ld32u(b + n);
        if (x) return n + (__ffs((int)x) - 1) / 8;
        n += 4;
    }
    while (n < kMatchMax && a[n] == b[n]) n++;
    return n;
}

// GetCommonLength for a wave-uniform pair, all 64 lanes helping: lane t compares bytes 4t .. 4t+3, lane 0 also the
// last three (256 .. 258), one round trip instead of one per 4 bytes.  Every lane must be active and hold the same a, b.
__device__ __forceinline__ int common_len_wave(const uint8_t* a, const uint8_t* b) {
    const uint32_t t4 = 4u * (threadIdx.x & 63u);
    const uint32_t x = ld32u(a + t4) ^ ld32u(b + t4);
    const uint32_t xt = (ld32u(a + 256) ^ ld32u(b + 256)) & 0xFFFFFFu;
    const unsigned long long m = __ballot(x != 0u);
    if (m) {
        const int f = (int)__builtin_ctzll(m);
        if (f == 0) return 0;
        return 4 * f + (__ffs((int)__builtin_amdgcn_readlane((int)x, f)) - 1) / 8;
    }
    const uint32_t xl = (uint32_t)__builtin_amdgcn_readfirstlane((int)xt);
    return xl ? 256 + (__ffs((int)xl) - 1) / 8 : kMatchMax;
}

// One context's dictionary plane.  Fields are addressed as (wave-uniform base) + (32-bit byte offset): the whole
// per-block dictionary is 10 MiB, so the offset fits a VGPR and every access is a global_load/store with an SGPR
// base ("saddr") -- no 64-bit per-lane pointer arithmetic.
template <class T, uint32_t kStride = (uint32_t)sizeof(T)> struct BktField {
    uint8_t* d; uint32_t o;
    __device__ __forceinline__ T& operator[](uint32_t i) const { return *reinterpret_cast<T*>(d + (o + kStride * i)); }
};
// Both forms of a bucket keep a ring slot in 8 bytes at b + 8 i (zlng_common.h):
//   wide   {own word, copy of the linked slot's word}, the link itself (suffix) in its own plane behind the slots;
//   paired {own word, link (u16), unused}: a chain hop of the generic walk reads a node's word AND its link from one
//          8-byte record -- one memory line per node instead of two (round 3; the "compact" form it replaces kept the links in a
//          second plane: with >= 128 blocks in flight the e1-e4 parse is bound by the lines it touches, profiles/r03_h).
template <bool kWide> struct BucketT {
    static constexpr uint32_t kSlot = 8u;
    BktField<uint32_t, kSlot> offset;          // a slot's own word
    BktField<unsigned long long, 8> slot;      // wide form only: own word + the linked slot's word as of the time the link was made
    BktField<uint16_t, kWide ? 2u : 8u> suffix; BktField<uint16_t> hash;
    __device__ __forceinline__ BucketT(uint8_t* dict, uint32_t ctx) {
        const uint32_t b = ctx * kBktBytes;
        offset = {dict, b};
        slot   = {dict, b};
        suffix = {dict, kWide ? b + kSlot * kRing : b + 4u};
        hash   = {dict, b + kSlot * kRing + 2u * kRing};
    }
    // a node's word and its link: ONE 8-byte load in the paired form (a scattered load instruction costs the compute unit's one
    // address path ~64 cycles per wavefront whatever its width, and a chain hop of the generic walk is bound by just that)
    __device__ __forceinline__ void node(uint32_t i, uint32_t& word, uint32_t& link) const {
        if (kWide) { word = offset[i]; link = suffix[i]; }
        else { const unsigned long long r = slot[i]; word = (uint32_t)r; link = (uint32_t)(r >> 32) & 0xFFFFu; }
    }
};
using Bucket = BucketT<false>;

// MatchLazy, src/libzling_lz.cpp:291-316
template <bool kWide = false>
__device__ __forceinline__ bool lazy_probe(uint8_t* dict, const uint8_t* buf, int pos, int maxlen, int depth) {
    BucketT<kWide> B(dict, buf[pos - 1]);
    uint32_t node = B.hash[hash4(buf + pos) % kHashSlots];
    if (node == 65535) return false;
    int m = maxlen - 3;
    for (int i = 0; i < depth; i++) {
        uint32_t off = B.offset[node] & 0xFFFFFF;
        if (ld32u(buf + pos + m) == ld32u(buf + off + m)) return true;
        node = B.suffix[node];
        if (node == 65535 || off <= (B.offset[node] & 0xFFFFFF)) break;
    }
    return false;
}

// MatchAndUpdate, src/libzling_lz.cpp:211-289 (insert first, then walk <= depth chain nodes).
// `head` is the ring slot this insert takes (the caller owns the per-context head counters).
// Safe to run wave-uniformly: every lane computes the same thing, lane 0 alone stores.
// kCopy: the wide slot plane, the insert also stores the copy of its link's word (zlng_common.h).
// kWave: called by a whole wavefront with identical arguments (the wave parser's replays): the LCP is one round trip.
template <bool kCopy = false, bool kWave = false>
__device__ __forceinline__ bool match_exact(uint8_t* dict, const uint8_t* buf, int pos, const LevelCfg cfg,
                                            uint32_t head, bool writer, int& match_idx, int& match_len) {
    uint32_t h = hash4(buf + pos);
    uint32_t chk = (h / kHashSlots) & 255u;
    uint32_t hc = h % kHashSlots;
    uint32_t ctx = buf[pos - 1];
    BucketT<kCopy> B(dict, ctx);
    uint32_t node = B.hash[hc];
    const uint32_t own = (uint32_t)pos | chk << 24;
    uint32_t ov_first = 0;
    if (kCopy && node != 65535 && node != head) ov_first = B.offset[node];
    if (writer) {
        B.suffix[head] = (uint16_t)node;
        if (kCopy) B.slot[head] = (unsigned long long)own | (unsigned long long)(node == head ? own : ov_first) << 32;
        else B.offset[head] = own;
        B.hash[hc] = (uint16_t)head;
    }
    if (node == 65535 || node == head) return false;

    int maxlen = kMatchMin - 1;
    uint32_t maxnode = 0;
    for (int i = 0; i < cfg.depth; i++) {
        // the slot just written is read back as written (it is `head`, checked above for i == 0;
        // later hits on it end the chain through the position test exactly as in the reference)
        uint32_t ov = node == head ? own : (kCopy && i == 0) ? ov_first : B.offset[node];
        uint32_t off = ov & 0xFFFFFF;
        if ((ov >> 24) == chk && buf[pos + maxlen] == buf[off + maxlen]) {
            int len = kWave ? common_len_wave(buf + pos, buf + off) : common_len(buf + pos, buf + off);
            if (len > maxlen) { maxnode = node; maxlen = len; if (maxlen == kMatchMax) break; }
        }
        uint32_t nx = B.suffix[node];
        if (nx == 65535) break;
        uint32_t noff = nx == head ? (uint32_t)pos : (B.offset[nx] & 0xFFFFFF);
        if (off <= noff) break;
        node = nx;
    }
    if (maxlen < kMatchMin) return false;
    if (maxlen < kLazyLimit) {
        if (cfg.lazy1 > 0 && lazy_probe<kCopy>(dict, buf, pos + 1, maxlen, cfg.lazy1)) return false;
        if (cfg.lazy2 > 0 && lazy_probe<kCopy>(dict, buf, pos + 2, maxlen, cfg.lazy2)) return false;
    }
    match_len = maxlen;
    match_idx = (int)((head - maxnode) & (kRing - 1));
    return true;
}


constexpr int kKeyTab = 4096;                        // 64-bit lane masks, indexed by a hash of (ctx, hash13)
constexpr int kEvTab = 4096;                         // 64-bit lane masks, indexed by a hash of (key byte, word)

__device__ __forceinline__ uint32_t key_ix(uint32_t ctx, uint32_t hc) { return (hc ^ (ctx * 0x9E5u)) & (kKeyTab - 1); }
__device__ __forceinline__ uint32_t ev_ix(uint32_t key, uint32_t word) { return (word ^ (word >> 7) ^ (key * 0x2D1u)) & (kEvTab - 1); }
__device__ __forceinline__ uint32_t ring_dist(uint32_t node, uint32_t head0) { return (node - head0 - 1u) & (kRing - 1); }
__device__ __forceinline__ uint32_t rl(uint32_t v, int lane) { return (uint32_t)__builtin_amdgcn_readlane((int)v, lane); }
__device__ __forceinline__ unsigned long long rl64(unsigned long long v, int lane) {
    return (unsigned long long)rl((uint32_t)(v >> 32), lane) << 32 | rl((uint32_t)v, lane);
}
__device__ __forceinline__ uint32_t ufl(uint32_t v) { return (uint32_t)__builtin_amdgcn_readfirstlane((int)v); }
__device__ __forceinline__ uint32_t hash_of(uint32_t w) { return w + ((w >> 16) & 0xFF) * 137u + (w >> 24) * 13337u; }
__device__ __forceinline__ int top_bit(unsigned long long m) { return 63 - __clzll((long long)m); }     // m != 0

// packed speculative result of one lane
constexpr uint32_t kSpLenMask = 0x1FF;               // bits 0..8  maxlen (3 = none)
constexpr int      kSpNodeShift = 9;                 // bits 9..20 maxnode
constexpr uint32_t kSpVeto1 = 1u << 21, kSpVeto2 = 1u << 22, kSpCanMatch = 1u << 23;
constexpr uint32_t kSpRisk1 = 1u << 25, kSpRisk2 = 1u << 26;   // lazy read set near the ring head
constexpr int      kMinRestart = 12;                 // restart a round at a conflict only if it resolved >= 12 positions
constexpr uint32_t kRiskDist = 64;                   // a round hands out <= 64 slots per context

// token kinds (also the "previous token" kind carried to the next boundary)
constexpr uint32_t kTyNone = 0, kTyLit = 1, kTyW0 = 2, kTyW1 = 3, kTyMatch = 4;

struct Quad { uint32_t a, b, c, d; };
__device__ __forceinline__ Quad ld128u(const uint8_t* p) { Quad q; __builtin_memcpy(&q, p, 16); return q; }
// Index of the first differing byte of two 16-byte groups (16 = equal).
__device__ __forceinline__ uint32_t first_diff16(const Quad x, const Quad y) {
    const unsigned long long lo = (unsigned long long)(x.a ^ y.a) | ((unsigned long long)(x.b ^ y.b) << 32);
    const unsigned long long hi = (unsigned long long)(x.c ^ y.c) | ((unsigned long long)(x.d ^ y.d) << 32);
    const uint32_t dl = lo ? ((uint32_t)__ffsll((long long)lo) - 1u) >> 3 : 8u;
    const uint32_t dh = hi ? ((uint32_t)__ffsll((long long)hi) - 1u) >> 3 : 8u;
    return lo ? dl : 8u + dh;
}

// byte-wise common prefix of a and b, capped at 259, given that it is going to be compared with
// a threshold >= 3: returns 0 when the first four bytes differ (GetCommonLength, src/libzling_lz.cpp:66-89).
__device__ __forceinline__ int common_len_q(const uint8_t* a, const uint8_t* b, const Quad qa) {
    const Quad qb = ld128u(b);
    uint32_t x = qa.a ^ qb.a;
    if (x) return 0;
    x = qa.b ^ qb.b; if (x) return 4 + (__ffs((int)x) - 1) / 8;
    x = qa.c ^ qb.c; if (x) return 8 + (__ffs((int)x) - 1) / 8;
    x = qa.d ^ qb.d; if (x) return 12 + (__ffs((int)x) - 1) / 8;
    int n = 16;
    while (n < kMatchMax) {
        const uint32_t d = first_diff16(ld128u(a + n), ld128u(b + n));
        n += (int)d;
        if (d < 16u) break;
    }
    return n < kMatchMax ? n : kMatchMax;
}

// Speculative evaluation of one position as a token start (phase 1 of the parser, also run ahead of
// it by the prefetch wavefront): hash head, <= depth chain nodes with exact LCP, lazy probes.
// Read-only on the dictionary.  `dmin` = ring distance (ahead of the context's head) of the nearest
// visited node; lz*/lkix*/lctx* describe the lazy probes' read sets.
struct Spec {
    uint32_t sp, node0, head0, dmin;
    uint32_t lkix1, lkix2, lctx1, lctx2;
    uint32_t ld1, ld2;                   // ring distance (ahead of the probe bucket's head) of the nearest node a lazy probe visited
    bool lz1, lz2;
    // level 0 only, for resolving same-slot conflicts in registers: candidate length of the first chain node
    // alone, the lazy probe's source offset (bit 31: probe chain non-empty), the 16 input bytes at the position
    uint32_t len0, lsrc1;
    Quad qa;
    // level 0 in the wave parser (speculate_l0w): a lane whose compare against a chain node reached 16 bytes is
    // "open" -- its lengths are lower bounds and its lazy probe has not been evaluated -- until the parser finishes it
    // (only token starts are ever finished).  off0 / off1: sources of the two nodes; olen: len0 | len1 << 8 |
    // long0 << 16 | long1 << 17 | node 1 exists << 18 | node 1's ring slot << 19.
    uint32_t off0, off1, olen;
    bool open;
    uint32_t ov0;                        // node 0's slot word (the inserting lane stores it as its link's copy)
    uint32_t lkey1;                      // exact key of the lazy probe at +1: context << 13 | hash13 (speculate_l0w)
    // generic levels, for the workgroup parser's in-window evaluation: best (len | node << 9) over the first depth-1 / depth-2
    // nodes of the chain, and the index of the first node of each probe's chain that vetoes (its depth if none)
    uint32_t pre1, pre2, vpos1, vpos2;
    // generic levels, ring rule (ZLNG_RING_FIX): ring distance of chain node 0, and for the first three nodes BEHIND it that lie within 64
    // slots ahead of the bucket's head -- the only ones a round's <= 64 inserts can reach -- distance | (best len | node << 9 over the
    // nodes in front of that one) << 6; ntail counts all such nodes (more than three: the rule gives up)
    uint32_t d0g, tail0, tail1, tail2, ntail;
};

__device__ __forceinline__ uint32_t lcp16(const Quad qa, const Quad qb) {      // 0 if the first 4 bytes differ, 16 = all equal
    const uint32_t x0 = qa.a ^ qb.a, x1 = qa.b ^ qb.b, x2 = qa.c ^ qb.c, x3 = qa.d ^ qb.d;
    uint32_t len = x3 ? 12u + ((uint32_t)__ffs((int)x3) - 1u) / 8u : 16u;
    len = x2 ? 8u + ((uint32_t)__ffs((int)x2) - 1u) / 8u : len;
    len = x1 ? 4u + ((uint32_t)__ffs((int)x1) - 1u) / 8u : len;
    return x0 ? 0u : len;
}
// Continue an LCP of 16 up to kMatchMax, 32 bytes per memory round trip: a
// 259-byte match costs 8 dependent loads, not 63 (long matches are the common
// case on source code and markup, where every lane of a round sits inside one).
__device__ __forceinline__ uint32_t lcp_tail(const uint8_t* a, const uint8_t* b, bool active) {
    uint32_t n = 16;
    while (__any(active)) {
        if (active) {
            const Quad a0 = ld128u(a + n), b0 = ld128u(b + n);
            const Quad a1 = ld128u(a + n + 16), b1 = ld128u(b + n + 16);
            uint32_t d = first_diff16(a0, b0);
            if (d == 16u) d += first_diff16(a1, b1);
            n += d;
            if (d < 32u || n >= (uint32_t)kMatchMax) active = false;
        }
    }
    return n < (uint32_t)kMatchMax ? n : (uint32_t)kMatchMax;
}


// Generic-depth form (levels 1-4: depth 4..16, lazy depths up to 4 and 2; src/libzling_lz.cpp:131-134).  Loops run
// wave-uniformly (`while any lane is still walking`) with per-lane predicates instead of per-lane breaks: a divergent
// break costs a handful of exec-mask instructions per lane group, a uniform loop costs one scalar branch.
__device__ __forceinline__ void lazy_spec_u(uint8_t* dict, const uint8_t* buf, int ppos, uint32_t lctx, uint32_t first, uint32_t lov,
                                            uint32_t lsfx, uint32_t m, int depth, uint32_t lhead, bool active, bool& veto, uint32_t& ld, uint32_t& vpos) {
    // (n, lov = offset[n], lsfx = suffix[n]) travel together, so a chain hop is ONE round trip: the source word
    // of node n and both ring fields of its successor are requested at the same time
    Bucket B(dict, lctx);
    uint32_t n = first;
    active = active && n != 65535u;
    const uint32_t probe = ld32u(buf + ppos + m);
    for (int i = 0; i < depth && __any(active); i++) {
        if (active) ld = min(ld, ring_dist(n, lhead));
        const uint32_t off = lov & 0xFFFFFF;
        const uint32_t srcw = ld32u(buf + (active ? off + m : (uint32_t)ppos));
        const uint32_t nn = lsfx;
        uint32_t nov, nsfx;
        B.node(nn & (kRing - 1), nov, nsfx);
        if (active && probe == srcw) { veto = true; vpos = (uint32_t)i; active = false; }
        active = active && nn != 65535u;
        if (active) ld = min(ld, ring_dist(nn, lhead));
        active = active && !(off <= (nov & 0xFFFFFF));
        n = nn; lov = nov; lsfx = nsfx;
    }
}

// head0 / lhead1 / lhead2: ring heads of the position's bucket and of the two probe buckets as the caller read them;
// risk_dist: a probe that visited a node within this many slots ahead of its bucket's head gets the kSpRisk flag.
// speculate_heads: the three hash heads of a position (round trip 1 of speculate).  A caller with work that does not depend on the
// dictionary (the wg parser's row claims) asks for them first and hands them to speculate_from.
__device__ __forceinline__ void speculate_heads(uint8_t* dict, const LevelCfg cfg, const Quad qa, uint32_t ctx, uint32_t hc,
                                                uint32_t& node0, uint32_t& ln1, uint32_t& ln2) {
    const uint32_t w4 = qa.a;
    const uint32_t hh1 = hash_of(w4 >> 8 | qa.b << 24) % kHashSlots;
    const uint32_t hh2 = hash_of(w4 >> 16 | qa.b << 16) % kHashSlots;
    Bucket B(dict, ctx), B1(dict, w4 & 0xFF), B2(dict, (w4 >> 8) & 0xFF);
    node0 = B.hash[hc];
    ln1 = cfg.lazy1 > 0 ? (uint32_t)B1.hash[hh1] : 65535u;
    ln2 = cfg.lazy2 > 0 ? (uint32_t)B2.hash[hh2] : 65535u;
}

__device__ __forceinline__ void speculate_from(Spec& S, uint8_t* dict, const uint8_t* buf, uint32_t head0, uint32_t lhead1, uint32_t lhead2,
                                               uint32_t risk_dist, int pos,
                                               const LevelCfg cfg, const Quad qa, uint32_t ctx, uint32_t hc, uint32_t chk,
                                               const uint32_t node0, const uint32_t ln1, const uint32_t ln2, const bool ring_rec = false) {
    const uint32_t w4 = qa.a;                        // qa: bytes pos .. pos+15, loaded by the caller (pos+275 < ilen)
    const bool want1 = cfg.lazy1 > 0, want2 = cfg.lazy2 > 0;
    const uint32_t lctx1 = w4 & 0xFF, lctx2 = (w4 >> 8) & 0xFF;
    const uint32_t hh1 = hash_of(w4 >> 8 | qa.b << 24) % kHashSlots;
    const uint32_t hh2 = hash_of(w4 >> 16 | qa.b << 16) % kHashSlots;
    Bucket B(dict, ctx), B1(dict, lctx1), B2(dict, lctx2);
    uint32_t ov, nx;
    B.node(node0 & (kRing - 1), ov, nx);
    S.ov0 = ov;                                      // the inserting lane stores it as its link's copy
    uint32_t lov1, lov2, lsf1, lsf2;
    B1.node(ln1 & (kRing - 1), lov1, lsf1);
    B2.node(ln2 & (kRing - 1), lov2, lsf2);

    uint32_t maxlen = kMatchMin - 1, maxnode = 0, node = node0, dmin = kRing - 1;
    bool active = node0 != 65535u;
    // The walk is software-pipelined: a hop costs one round trip (a node's compare block together with both ring fields
    // of its successor), and the NEXT hop's round trip is requested before this node's bytes are compared -- the ≈70
    // instructions of a node (LCP, best-of, chain-end tests under exec masks) run while it is in flight.  The request
    // cannot know yet whether this node ends the walk with a maximum-length match; one set of loads may go unused.
    uint32_t off = ov & 0xFFFFFF;
    bool cmp = active && (ov >> 24) == chk;
    Quad qb = ld128u(buf + (cmp ? off : (uint32_t)pos));
    uint32_t nov, nnx;
    B.node(nx & (kRing - 1), nov, nnx);
    uint32_t pre1 = 0xFFFFFFFFu, pre2 = 0xFFFFFFFFu;                       // (unset: the walk ended before that many nodes)
    uint32_t d0g = kRing - 1, tail0 = 0, tail1 = 0, tail2 = 0, ntail = 0;
    for (int i = 0; i < cfg.depth && __any(active); i++) {                 // src/libzling_lz.cpp:240-267
        if (i == cfg.depth - 1) pre1 = maxlen | maxnode << kSpNodeShift;       // the first i nodes are in
        if (i == cfg.depth - 2) pre2 = maxlen | maxnode << kSpNodeShift;
        if (active) dmin = min(dmin, ring_dist(node, head0));
        if (ring_rec) {   // the ring rule's record of this node (positions fall along the walk, so do the distances: the recorded nodes are its tail)
            const uint32_t dn = ring_dist(node, head0);
            if (i == 0 && active) d0g = dn;
            const bool rec = active && i > 0 && dn < 64u;
            const uint32_t ent = dn | (maxlen | maxnode << kSpNodeShift) << 6;
            tail0 = rec && ntail == 0u ? ent : tail0;
            tail1 = rec && ntail == 1u ? ent : tail1;
            tail2 = rec && ntail == 2u ? ent : tail2;
            ntail += rec ? 1u : 0u;
        }
        const uint32_t off_n = nov & 0xFFFFFF;
        const bool more = active && nx != 65535u && !(off <= off_n);
        const bool cmp_n = more && (nov >> 24) == chk;
        const Quad qb_n = ld128u(buf + (cmp_n ? off_n : (uint32_t)pos));
        uint32_t nov_n, nnx_n;
        B.node(nnx & (kRing - 1), nov_n, nnx_n);
        uint32_t len = cmp ? lcp16(qa, qb) : 0u;
        const bool lng = cmp && len == 16u;
        if (__any(lng)) { const uint32_t t = lcp_tail(buf + pos, buf + off, lng); len = lng ? t : len; }
        if (len > maxlen) { maxlen = len; maxnode = node; }
        active = active && maxlen != (uint32_t)kMatchMax && nx != 65535u;
        if (active) dmin = min(dmin, ring_dist(nx, head0));
        active = active && !(off <= off_n);
        node = nx; ov = nov; nx = nnx;
        off = off_n; cmp = active && cmp_n; qb = qb_n; nov = nov_n; nnx = nnx_n;
    }
    uint32_t sp = maxlen | maxnode << kSpNodeShift | kSpCanMatch;
    S.pre1 = pre1 == 0xFFFFFFFFu ? (maxlen | maxnode << kSpNodeShift) : pre1;
    S.pre2 = pre2 == 0xFFFFFFFFu ? (maxlen | maxnode << kSpNodeShift) : pre2;
    S.d0g = d0g; S.tail0 = tail0; S.tail1 = tail1; S.tail2 = tail2; S.ntail = ntail;
    const bool lz = maxlen >= (uint32_t)kMatchMin && maxlen < (uint32_t)kLazyLimit;
    const uint32_t m = lz ? maxlen - 3u : 0u;
    bool v1 = false, v2 = false;
    uint32_t ld1 = kRing - 1, ld2 = kRing - 1;
    uint32_t vp1 = (uint32_t)cfg.lazy1, vp2 = (uint32_t)cfg.lazy2;
    if (want1) lazy_spec_u(dict, buf, pos + 1, lctx1, ln1, lov1, lsf1, m, cfg.lazy1, lhead1, lz, v1, ld1, vp1);
    if (want2) lazy_spec_u(dict, buf, pos + 2, lctx2, ln2, lov2, lsf2, m, cfg.lazy2, lhead2, lz, v2, ld2, vp2);
    S.vpos1 = vp1; S.vpos2 = vp2;
    if (v1) sp |= kSpVeto1;
    if (v2) sp |= kSpVeto2;
    if (ld1 < risk_dist) sp |= kSpRisk1;
    if (ld2 < risk_dist) sp |= kSpRisk2;
    S.sp = sp; S.node0 = node0; S.head0 = head0; S.dmin = dmin;
    S.lkix1 = key_ix(lctx1, hh1); S.lkix2 = key_ix(lctx2, hh2); S.lctx1 = lctx1; S.lctx2 = lctx2;
    S.ld1 = ld1; S.ld2 = ld2;
    S.lz1 = lz && want1; S.lz2 = lz && want2;
}

// Two LCP tails against the same position in one loop (level 0 compares the two newest chain nodes): both pairs
// are at the same byte count while they are alive, so the position side is loaded once and a round in a
// long-match region pays 8 round trips, not 16.
__device__ __forceinline__ void lcp_tail2(const uint8_t* a, const uint8_t* b, const uint8_t* c, bool act0, bool act1,
                                          uint32_t& r0, uint32_t& r1) {
    uint32_t n = 16;
    r0 = 16; r1 = 16;
    while (__any(act0 || act1)) {
        if (act0 || act1) {
            const uint8_t* pb = act0 ? b : a;
            const uint8_t* pc = act1 ? c : a;
            const Quad a0 = ld128u(a + n), a1 = ld128u(a + n + 16);
            const Quad b0 = ld128u(pb + n), b1 = ld128u(pb + n + 16);
            const Quad c0 = ld128u(pc + n), c1 = ld128u(pc + n + 16);
            uint32_t d0 = first_diff16(a0, b0);
            if (d0 == 16u) d0 += first_diff16(a1, b1);
            uint32_t d1 = first_diff16(a0, c0);
            if (d1 == 16u) d1 += first_diff16(a1, c1);
            if (act0) { r0 = n + d0; act0 = d0 == 32u && r0 < (uint32_t)kMatchMax; }
            if (act1) { r1 = n + d1; act1 = d1 == 32u && r1 < (uint32_t)kMatchMax; }
            n += 32;
        }
    }
    r0 = r0 < (uint32_t)kMatchMax ? r0 : (uint32_t)kMatchMax;
    r1 = r1 < (uint32_t)kMatchMax ? r1 : (uint32_t)kMatchMax;
}

// (The level-0 speculations of the one-wavefront and the pipelined parser -- speculate_l0, speculate_l0w with its "open lane"
//  convention -- left with those parsers in round 4 (git history: scripts/experiments/retired/ up to commit 0899800).  The workgroup-wide parser's level-0 form,
//  speculate_l0t, lives in rolz_wg.hip; the wide slot plane it reads -- a slot's own word + a copy of the word of the slot it
//  links to, taken when the link was made -- is described at BucketT above and in DESIGN.md, K1.)

// Ordering point for LDS traffic inside ONE wavefront (program order is execution order for a wave's LDS
// operations; this only stops the compiler from moving accesses across it).  Used where ONE wavefront of the parser's
// workgroup works alone (the serial token of a hard lane) between two workgroup barriers.
__device__ __forceinline__ void wsync() {
    __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
    __builtin_amdgcn_wave_barrier();
}

}  // namespace zlng
