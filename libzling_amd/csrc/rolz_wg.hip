// rolz_wg.hip -- K1, workgroup-wide form: the production ROLZ block parser for gfx950.
//
// Replaces ZlingRolzEncoder::Encode / EncodeImpl / MatchAndUpdate / MatchLazy (src/libzling_lz.cpp:128-316 of the
// reference).  One WORKGROUP of NW wavefronts (one per SIMD of a CU, or two) owns a 16 MiB block and advances in rounds
// over a window of NL = 64 NW consecutive input positions that starts at the next token start P; lane g of the workgroup
// stands for position P + g.  (rolz_parse.hip holds the one-wavefront form of round 1-2, kept as a cross-check.)
//
// The parse of a block is a serial chain -- dictionary inserts happen only at token starts (src/libzling_lz.cpp:159), so
// whether a position is a token start, and what its match is, depends on every earlier token of the block.  A round
// resolves that chain for its window by a FIXED-POINT ITERATION in which every step is lane-parallel:
//
//   phase 1   every lane evaluates its position AS IF it were a token start against the dictionary as of the start of the
//             round (read only): hash head, chain nodes, longest match, lazy probes (rolz_dev.h speculation).
//   tables    lane-mask tables in LDS give every lane the lanes of the window that share its hash slot (ctx, hash13), its
//             bucket, its lazy probes' slots and buckets, and its word-MRU keys (hashed tables are verified against the
//             exact keys, so the masks are exact).
//   iterate   S = the token starts reached from lane 0 under the current per-lane token lengths (pointer doubling inside a
//             wavefront, one hop per wavefront across them).  Then E(g, S): every lane re-evaluates its own token GIVEN the
//             accepted starts before it --
//               * the word MRU of its context after every boundary event of S up to g (src/libzling_lz.cpp:163-191: which
//                 boundaries push, conditionally or not, follows from the token types of S; the two slots follow from the
//                 last event and the last EFFECTIVE event of the key: all mask arithmetic);
//               * its match: an accepted earlier start with its (ctx, hash13) is now the head of its chain -- at level 0
//                 (depth 2) the chain is [newest such start, the one before it | the snapshot's head] and the candidate
//                 lengths come from the window's own text; a ring slot rewritten by a start of this round ends the chain
//                 there (the reference's position test, src/libzling_lz.cpp:265); the lazy probe (depth 1) sees the newest
//                 accepted start with ITS key, the lane's own insert included (src/libzling_lz.cpp:271);
//               * whatever is not covered exactly (the hash head's own slot rewritten, a lazy probe near the ring head,
//                 every same-slot conflict at levels 1-4) makes the lane HARD;
//             until no lane of S changed its token.  At that fixed point every lane of S was evaluated under exactly the
//             inserts the reference has made by then, so the tokens are the reference's; each iteration fixes at least the
//             first wrong lane.  (scripts/experiments/wg_parser_model.c is the same algorithm in plain C, checked against
//             the oracle on every corpus kind at every level; it also gives the statistics quoted in DESIGN.md.)
//   commit    S up to the first hard lane or the end of the sub-block: dictionary inserts, token words, MRU slots, ring heads.
//             A hard lane is then replayed by the exact serial code (match_exact = MatchAndUpdate as written).
//
// Output per block: one u32 word per token (zlng_common.h) with literals still RAW, plus the sub-block cut list.
#include "zlng_common.h"
#include "zlng_kernels.h"
#include "rolz_dev.h"

namespace zlng {

typedef unsigned long long u64;

constexpr int kWgKeyTab = 2048;                     // rows of the hashed (ctx, hash13) lane-mask table

__device__ __forceinline__ uint32_t wg_key_ix(uint32_t key21) { return ((key21 & 0x1FFFu) ^ ((key21 >> 13) * 0x9E5u) ^ (key21 >> 7)) & (kWgKeyTab - 1); }
__device__ __forceinline__ u64 uni64(u64 v) { return (u64)ufl((uint32_t)(v >> 32)) << 32 | ufl((uint32_t)v); }

// What phase 1 leaves for one lane.
struct WSpec {
    uint32_t len, node;          // longest match (3 = none) and its ring slot
    uint32_t node0, ov0;         // hash head (65535 = none) and that slot's word (the inserting lane stores it as its link's copy)
    uint32_t d0, d1, dmin;       // ring distance ahead of the bucket's head of chain node 0 / node 1 (level 0) / the nearest visited node
    bool has0, has1;             // node 0 exists / node 1's slot was read (level 0)
    uint32_t len0;               // level 0: candidate length of node 0 alone (0: absent or check byte differs)
    bool veto1, veto2, lz1, lz2; // lazy probes: speculative outcome / evaluated at all
    uint32_t lkey1, lkey2;       // exact keys of the probes: context << 13 | hash13
    uint32_t ld1, ld2;           // ring distance ahead of the PROBE bucket's head of the nearest node a probe visited (4095 = none)
    uint32_t lsrc1;              // level 0: source offset of the probe's chain head | exists << 31
};

// Level 0 (depth 2, one lazy probe of depth 1; src/libzling_lz.cpp:130), straight-line predicated code: three dependent round
// trips after the window's text with the wide slot plane (rolz_dev.h speculate_l0w explains the link copy), then -- only if some
// lane's compare ran to 16 bytes -- the tails (32 bytes per trip, both nodes in one loop) and the lazy probes of those lanes.
template <bool kWide>
__device__ __forceinline__ void speculate_l0t(WSpec& W, uint8_t* dict, const uint8_t* buf, uint32_t head0, uint32_t lhead1, uint32_t pos,
                                              const Quad qa, uint32_t t16, uint32_t ctx, uint32_t hc, uint32_t chk) {
    const uint32_t w4 = qa.a;
    const uint32_t lctx1 = w4 & 0xFF;
    const uint32_t hh1 = hash_of(w4 >> 8 | qa.b << 24) % kHashSlots;
    BucketT<kWide> B(dict, ctx), B1(dict, lctx1);
    // round trip 1: both hash heads
    const uint32_t node0 = B.hash[hc];
    const uint32_t ln1 = B1.hash[hh1];
    const bool has0 = node0 != 65535u, hasl = ln1 != 65535u;
    // round trip 2: node 0's slot (wide: own word + its link's), its link, the probe node's word
    uint32_t ov0, nov = 0;
    if (kWide) { const u64 sl0 = B.slot[node0 & (kRing - 1)]; ov0 = (uint32_t)sl0; nov = (uint32_t)(sl0 >> 32); }
    else ov0 = B.offset[node0 & (kRing - 1)];
    const uint32_t nx = B.suffix[node0 & (kRing - 1)];
    const uint32_t lov1 = B1.offset[ln1 & (kRing - 1)];
    const uint32_t off0 = ov0 & 0xFFFFFF;
    const bool has1s = has0 && nx != 65535u;
    const bool cmp0 = has0 && (ov0 >> 24) == chk;
    Quad q0, q1, ql;
    uint32_t off1;
    bool cmp1;
    if (kWide) {
        const uint32_t dnx = (nx - node0) & (kRing - 1), age0 = (head0 - node0) & (kRing - 1);
        const bool rewritten = dnx != 0u && dnx <= age0;
        off1 = nov & 0xFFFFFF;
        cmp1 = has1s && !rewritten && !(off0 <= off1) && (nov >> 24) == chk;
        q0 = ld128u(buf + (cmp0 ? off0 : pos));
        q1 = ld128u(buf + (cmp1 ? off1 : pos));
        ql = ld128u(buf + (hasl ? (lov1 & 0xFFFFFF) : pos));
    } else {
        q0 = ld128u(buf + (cmp0 ? off0 : pos));
        nov = B.offset[nx & (kRing - 1)];
        ql = ld128u(buf + (hasl ? (lov1 & 0xFFFFFF) : pos));
        off1 = nov & 0xFFFFFF;
        cmp1 = has1s && !(off0 <= off1) && (nov >> 24) == chk;
        q1 = ld128u(buf + (cmp1 ? off1 : pos));
    }
    uint32_t len0 = cmp0 ? lcp16(qa, q0) : 0u;
    uint32_t len1 = cmp1 ? lcp16(qa, q1) : 0u;
    const bool long0 = cmp0 && len0 == 16u, long1 = cmp1 && len1 == 16u;
    if (__any(long0 || long1)) {
        uint32_t t0, t1;
        lcp_tail2(buf + pos, buf + off0, buf + off1, long0, long1, t0, t1);
        len0 = long0 ? t0 : len0;
        len1 = long1 ? t1 : len1;
    }
    uint32_t maxlen = kMatchMin - 1, maxnode = 0;
    if (len0 > maxlen) { maxlen = len0; maxnode = node0; }
    const bool has1 = has1s && maxlen != (uint32_t)kMatchMax;            // the reference reads node 1's offset for the chain-end test
    if (has1 && len1 > maxlen) { maxlen = len1; maxnode = nx & (kRing - 1); }
    // the lazy probe at pos + 1 (src/libzling_lz.cpp:291-316, depth 1): position bytes m+1 .. m+4 against source bytes m .. m+3
    const bool lz1 = maxlen >= (uint32_t)kMatchMin && maxlen < (uint32_t)kLazyLimit;
    const uint32_t m = lz1 ? maxlen - 3u : 0u;
    bool veto = false;
    {   // m <= 12: both words lie in bytes 1..16 of the position and bytes 0..15 of the probe node's source
        const uint32_t x0 = __builtin_amdgcn_alignbyte(qa.b, qa.a, 1u) ^ ql.a, x1 = __builtin_amdgcn_alignbyte(qa.c, qa.b, 1u) ^ ql.b;
        const uint32_t x2 = __builtin_amdgcn_alignbyte(qa.d, qa.c, 1u) ^ ql.c, x3 = __builtin_amdgcn_alignbyte(t16, qa.d, 1u) ^ ql.d;
        const uint32_t mm = m <= 12u ? m : 0u, dw = mm >> 2;
        const uint32_t xl = dw == 0u ? x0 : dw == 1u ? x1 : dw == 2u ? x2 : x3;
        const uint32_t xh = dw == 0u ? x1 : dw == 1u ? x2 : x3;
        veto = lz1 && hasl && m <= 12u && __builtin_amdgcn_alignbyte(xh, xl, mm & 3u) == 0u;
    }
    const bool far = lz1 && hasl && m > 12u;
    if (__any(far)) {
        const uint32_t pr = ld32u(buf + (pos + 1u + (far ? m : 0u)));
        const uint32_t sr = ld32u(buf + (far ? (lov1 & 0xFFFFFF) + m : pos));
        if (far) veto = pr == sr;
    }
    W.len = maxlen; W.node = maxnode; W.node0 = node0; W.ov0 = ov0;
    W.has0 = has0; W.has1 = has1;
    W.d0 = has0 ? ring_dist(node0, head0) : (uint32_t)kRing - 1u;
    W.d1 = has1 ? ring_dist(nx, head0) : (uint32_t)kRing - 1u;
    W.dmin = min(W.d0, W.d1);
    W.len0 = len0;
    W.veto1 = veto; W.veto2 = false; W.lz1 = lz1; W.lz2 = false;
    W.lkey1 = lctx1 << 13 | hh1; W.lkey2 = 0;
    W.ld1 = hasl ? ring_dist(ln1, lhead1) : (uint32_t)kRing - 1u; W.ld2 = kRing - 1;
    W.lsrc1 = (lov1 & 0xFFFFFF) | (hasl ? 0x80000000u : 0u);
}

// GetCommonLength (src/libzling_lz.cpp:66-89) of the lane's position (first 16 bytes in qa) with another position of the block.
__device__ __forceinline__ uint32_t lcp_with(const uint8_t* buf, uint32_t pos, uint32_t src, const Quad qa, bool act) {
    const Quad qb = ld128u(buf + (act ? src : pos));
    uint32_t len = act ? lcp16(qa, qb) : 0u;
    const bool lng = act && len == 16u;
    if (__any(lng)) { const uint32_t t = lcp_tail(buf + pos, buf + src, lng); len = lng ? t : len; }
    return len;
}

// ---- NW-word lane masks: M per lane (VGPRs), U wave-uniform (SGPRs).  `wv` = this wavefront, `own` = which lanes of its
// own word count (below / below-or-equal the lane); words of later wavefronts never count.
template <int NW>
__device__ __forceinline__ int top_in(const u64* M, const u64* U, int wv, u64 own) {
    int r = -1;
#pragma unroll
    for (int w = 0; w < NW; w++) {
        if (w <= wv) {
            u64 m = M[w] & U[w];
            if (w == wv) m &= own;
            if (m) r = 64 * w + top_bit(m);
        }
    }
    return r;
}
template <int NW>
__device__ __forceinline__ uint32_t cnt_in(const u64* M, const u64* U, int wv, u64 own) {
    uint32_t c = 0;
#pragma unroll
    for (int w = 0; w < NW; w++) {
        if (w <= wv) {
            u64 m = M[w] & U[w];
            if (w == wv) m &= own;
            c += (uint32_t)__popcll(m);
        }
    }
    return c;
}
// count of (M & U) strictly below the global lane `b` (per-lane b)
template <int NW>
__device__ __forceinline__ uint32_t cnt_below_lane(const u64* M, const u64* U, int b) {
    uint32_t c = 0;
    const int bw = b >> 6;
    const u64 bm = (1ull << (b & 63)) - 1ull;
#pragma unroll
    for (int w = 0; w < NW; w++) {
        u64 m = M[w] & U[w];
        m = w < bw ? m : (w == bw ? (m & bm) : 0ull);
        c += (uint32_t)__popcll(m);
    }
    return c;
}
template <int NW>
__device__ __forceinline__ bool any_above(const u64* M, const u64* U, int wv, u64 above_own) {      // lanes after this one
    bool r = false;
#pragma unroll
    for (int w = 0; w < NW; w++) {
        if (w >= wv) {
            u64 m = M[w] & U[w];
            if (w == wv) m &= above_own;
            r = r || m != 0ull;
        }
    }
    return r;
}

// Keep of a hashed mask row only the lanes whose exact key is `want` (lanes of this wavefront's word limited to `own`, later
// wavefronts dropped).  Candidates are rare (the table has 8 rows per lane), so the loop usually does not run at all.
template <int NW>
__device__ __forceinline__ void verify_row(u64* M, const uint32_t* a_key, uint32_t want, int wv, u64 own, bool act) {
#pragma unroll
    for (int w = 0; w < NW; w++) {
        if (w > wv) { M[w] = 0ull; continue; }
        u64 c = act ? M[w] : 0ull;
        if (w == wv) c &= own;
        u64 keep = 0ull;
        while (__any(c != 0ull)) {
            if (c != 0ull) {
                const int b = (int)__builtin_ctzll(c);
                if ((a_key[64 * w + b] & 0x1FFFFFu) == want) keep |= 1ull << b;
                c &= c - 1ull;
            }
        }
        M[w] = keep;
    }
}

template <int NW, bool kAllL0, bool kProf>
__global__ __launch_bounds__(64 * NW) void k_rolz_parse_wg(ParseArgs a) {
    constexpr int NL = 64 * NW;
    __shared__ uint16_t heads[256];
    __shared__ uint32_t mru[256];                    // slot0 | slot1 << 16
    // lane-mask tables, NW words per row; the extra last row of each is a sink for lanes that must not deposit
    __shared__ u64 keytab[(kWgKeyTab + 1) * NW];
    __shared__ u64 ctxtab[257 * NW];
    __shared__ u64 ektab[257 * NW];
    __shared__ uint32_t a_key[NL];                   // per lane: key21 | chk << 21
    __shared__ uint32_t a_ev[NL];                    // per lane: event word | event key << 16
    __shared__ uint32_t a_st[NL];                    // per lane: token kind | token length << 8
    __shared__ uint32_t a_s0b[NL];                   // per lane: MRU slot 0 of its event key just before its event
    __shared__ uint32_t a_succ[NL];                  // per lane: lowest lane of S that links to it in its hash slot (NL = none)
    __shared__ uint32_t c_exit[NL];                  // closure: first chain position beyond the lane's wavefront (global lane index)
    __shared__ u64 c_mask[NL];                       // closure: chain positions inside the lane's wavefront
    __shared__ u64 u_ev[NW], u_eff[NW], u_hard[NW], u_chg[NW], u_mat[NW], u_cut[NW];
    __shared__ uint32_t u_ser[4];                    // serial token: q, opos, nt, kind

    const uint32_t blk = blockIdx.x + a.blk0;
    const size_t base = (size_t)blk * kBlockIn;
    if (base >= a.in_len) return;
    const uint8_t* buf = a.in + base;
    const int ilen = (int)((a.in_len - base) < (size_t)kBlockIn ? (a.in_len - base) : (size_t)kBlockIn);
    uint8_t* dict = a.dict + (size_t)blk * kDictBytes;
    uint32_t* tok = a.tok + (size_t)blk * a.tok_cap;
    SubCut* cuts = a.cuts + (size_t)blk * kMaxSub;
    const int tid = (int)threadIdx.x;
    const int lane = tid & 63;
    const int wv = (int)ufl((uint32_t)(tid >> 6));
    const u64 lane_bit = 1ull << lane;
    const u64 below = lane_bit - 1ull, beloweq = below | lane_bit, above = ~beloweq;
    constexpr bool kWide = kAllL0;                   // slot plane form (zlng_common.h): the launcher's reset matches

    for (int i = tid; i < 257 * NW; i += NL) { ctxtab[i] = 0; ektab[i] = 0; }
    for (int i = tid; i < (kWgKeyTab + 1) * NW; i += NL) keytab[i] = 0;
    for (int i = tid; i < 256; i += NL) heads[i] = 0;
    __syncthreads();

    uint32_t nt = 0;
    int q = 0, nsub = 0;
    bool overflow = false;
    u64 c_p1 = 0, c_tab = 0, c_it = 0, c_com = 0, c_ser = 0, n_round = 0, n_iter = 0, n_ser = 0, n_hard[4] = {0, 0, 0, 0}, n_pos = 0, n_cutr = 0;
    u64 c_a = 0, c_b = 0, c_c1 = 0, c_c2 = 0, c_c3 = 0, c_lim = 0, c_cl = 0, c_t1 = 0, c_t2 = 0;
    const bool prof = kProf && a.dbg != nullptr;

    while (q < ilen && !overflow) {                  // ---- one sub-block (one EncodeImpl call)
        const uint32_t lvl = kAllL0 ? 0u : (uint32_t)a.lvl_sched[blk * kMaxSub + (nsub < kMaxSub ? nsub : kMaxSub - 1)];
        const LevelCfg cfg = level_cfg((int)lvl);
        const bool level0 = kAllL0 || lvl == 0u;
        const uint32_t tok_begin = nt;
        int opos = 0;
        uint32_t prevty = kTyNone;                   // kind of the token that ended at q (none: the MRU starts empty)
        for (int i = tid; i < 256; i += NL) mru[i] = 0;
        __syncthreads();
        if (q == 0) {                                // src/libzling_lz.cpp:150-151
            if (tid == 0) tok[nt] = (uint32_t)buf[0] | kTokRawCtx << 16;
            nt++; q = 1; opos = 1;
            if (ilen > 1) { if (tid == 0) tok[nt] = (uint32_t)buf[1] | kTokRawCtx << 16; nt++; q = 2; opos = 2; }
        }
        bool serial_next = false;

        while (q < ilen && opos + 1 < kSubSyms) {
            q = (int)ufl((uint32_t)q); opos = (int)ufl((uint32_t)opos); nt = ufl(nt); prevty = ufl(prevty);
            if (nt + (uint32_t)NL > a.tok_cap) { overflow = true; break; }

            if (serial_next) {
                // ---------------- exact serial token at q (a hard lane): the pending boundary event, then MatchAndUpdate /
                // word MRU / literal exactly as EncodeImpl does (src/libzling_lz.cpp:158-191).  Wavefront 0 alone.
                serial_next = false;
                u64 ts = 0;
                if (prof) ts = __builtin_readcyclecounter();
                if (wv == 0) {
                    const uint32_t wq = ld32u(buf + (uint32_t)(q >= 4 ? q - 4 : 0));
                    const uint32_t wp = q >= 4 ? wq : wq << ((8u * (4u - (uint32_t)q)) & 31u);
                    const uint32_t cq = wp >> 24, xk = (wp >> 8) & 0xFF, xw = ((wp >> 16) & 0xFF) << 8 | cq;
                    if (prevty == kTyMatch) { const uint32_t m = ufl(mru[xk]); if ((m & 0xFFFF) != xw && lane == 0) mru[xk] = (m << 16) | xw; }
                    else if (prevty == kTyLit || prevty == kTyW1) { const uint32_t m = ufl(mru[xk]); if (lane == 0) mru[xk] = (m << 16) | xw; }
                    wsync();
                    bool is_match = false;
                    int mlen = 0, midx = 0;
                    if (q + kSentinel < ilen) {
                        const uint32_t head = (ufl((uint32_t)heads[cq]) + 1u) & (kRing - 1);
                        int mi = 0, ml = 0;
                        const bool hit = match_exact<kWide, true>(dict, buf, q, cfg, head, lane == 0, mi, ml);
                        is_match = __builtin_amdgcn_readfirstlane((int)hit) != 0;
                        mlen = __builtin_amdgcn_readfirstlane(ml);
                        midx = __builtin_amdgcn_readfirstlane(mi);
                        if (lane == 0) heads[cq] = (uint16_t)head;
                    }
                    uint32_t word, nq = (uint32_t)q, no = (uint32_t)opos, ty;
                    if (is_match) { word = (uint32_t)(258 + mlen - kMatchMin) | (uint32_t)midx << 16; no += 2; nq += (uint32_t)mlen; ty = kTyMatch; }
                    else {
                        const uint32_t w = (uint32_t)buf[q] << 8 | buf[q + 1];
                        const uint32_t m = ufl(mru[cq]);
                        if (q + 1 < ilen && (m & 0xFFFF) == w) { word = 256; no++; nq += 2; ty = kTyW0; }
                        else if (q + 1 < ilen && (m >> 16) == w) { word = 257; no++; nq += 2; ty = kTyW1; }
                        else { word = (w >> 8) | cq << 16; no++; nq++; ty = kTyLit; }
                    }
                    if (lane == 0) { tok[nt] = word; u_ser[0] = nq; u_ser[1] = no; u_ser[2] = nt + 1u; u_ser[3] = ty; }
                }
                __syncthreads();
                q = (int)ufl(u_ser[0]); opos = (int)ufl(u_ser[1]); nt = ufl(u_ser[2]); prevty = ufl(u_ser[3]);
                __syncthreads();                     // u_ser is free again
                if (prof) { c_ser += __builtin_readcyclecounter() - ts; n_ser++; }
                continue;
            }

            // ================================================================ one round
            const int P = q;
            const int nlive = ilen - P < NL ? ilen - P : NL;
            u64 t0 = 0, t1 = 0, t2 = 0, t3 = 0;
            if (prof) t0 = __builtin_readcyclecounter();
            // ---------------- phase 1
            const int pos = P + tid;
            const bool live = pos < ilen;
            const bool canm = pos + kSentinel < ilen;
            // lanes past the end of the block read the text of the round's first position instead (their results are never used)
            const uint32_t upos = live ? (uint32_t)pos : (uint32_t)P;
            const uint32_t wraw = ld32u(buf + (upos >= 4u ? upos - 4u : 0u));
            const Quad qtext = ld128u(buf + upos);
            const uint32_t t16 = ld32u(buf + (upos + 16u));
            const uint32_t wp = upos >= 4u ? wraw : wraw << ((8u * (4u - upos)) & 31u);
            const uint32_t w4 = qtext.a;
            const uint32_t ctx = wp >> 24;
            const uint32_t h = hash_of(w4);
            const uint32_t hc = h % kHashSlots, chk = (h / kHashSlots) & 255u;
            const uint32_t key = ctx << 13 | hc;
            const uint32_t b_m3 = (wp >> 8) & 0xFF, b_m2 = (wp >> 16) & 0xFF, b_0 = w4 & 0xFF, b_1 = (w4 >> 8) & 0xFF;
            const uint32_t cw = b_0 << 8 | b_1;                    // check: mru[ctx] vs (b0, b1)
            const uint32_t ek = b_m3, ew = b_m2 << 8 | ctx;        // event at this boundary: mru[b-3] <- (b-2, b-1)

            WSpec W;
            uint32_t lctx1 = w4 & 0xFF, lctx2 = (w4 >> 8) & 0xFF;
            if (level0) {
                speculate_l0t<kWide>(W, dict, buf, heads[ctx], heads[lctx1], upos, qtext, t16, ctx, hc, chk);
            } else {
                Spec S1;
                S1.sp = kMatchMin - 1; S1.node0 = 65535; S1.head0 = 0; S1.dmin = kRing - 1;
                S1.lkix1 = S1.lkix2 = S1.lctx1 = S1.lctx2 = 0; S1.lz1 = S1.lz2 = false; S1.ld1 = S1.ld2 = kRing - 1; S1.ov0 = 0;
                if (canm) speculate(S1, dict, buf, heads[ctx], heads[lctx1], heads[lctx2], 0u, pos, cfg, qtext, ctx, hc, chk);
                W.len = S1.sp & kSpLenMask; W.node = (S1.sp >> kSpNodeShift) & (kRing - 1);
                W.node0 = S1.node0; W.ov0 = S1.ov0; W.dmin = S1.dmin; W.d0 = W.d1 = S1.dmin;
                W.has0 = S1.node0 != 65535u; W.has1 = false; W.len0 = 0;
                W.veto1 = (S1.sp & kSpVeto1) != 0; W.veto2 = (S1.sp & kSpVeto2) != 0; W.lz1 = S1.lz1; W.lz2 = S1.lz2;
                W.lkey1 = lctx1 << 13 | (hash_of(w4 >> 8 | qtext.b << 24) % kHashSlots);
                W.lkey2 = lctx2 << 13 | (hash_of(w4 >> 16 | qtext.b << 16) % kHashSlots);
                W.ld1 = S1.ld1; W.ld2 = S1.ld2; W.lsrc1 = 0;
            }
            const uint32_t head0 = heads[ctx];
            const bool want1 = cfg.lazy1 > 0, want2 = cfg.lazy2 > 0;
            // speculative token of this lane
            const bool sp_veto = (want1 && W.veto1) || (want2 && W.veto2);
            const bool sp_match = canm && W.len >= (uint32_t)kMatchMin && !(W.len < (uint32_t)kLazyLimit && sp_veto);
            uint32_t ty = sp_match ? kTyMatch : kTyLit;            // token kind / length / match of this lane under the current S
            uint32_t tlen = sp_match ? W.len : 1u;
            uint32_t mlen = W.len, mnode = W.node;                  // mnode: ring slot, or 0x10000 | lane for a start of this round
            int link = -1;                                          // in-slot predecessor among the starts of this round (-1: the snapshot's head)
            uint32_t link_chk = 0;
            uint32_t k_ctx = 0;                                     // accepted earlier starts in my bucket
            if (prof) t1 = __builtin_readcyclecounter();

            // ---------------- tables and per-lane arrays
            const uint32_t kix = canm ? wg_key_ix(key) : (uint32_t)kWgKeyTab, ctx_w = canm ? ctx : 256u, ek_w = live ? ek : 256u;
            atomicOr(&keytab[kix * NW + wv], lane_bit);
            atomicOr(&ctxtab[ctx_w * NW + wv], lane_bit);
            atomicOr(&ektab[ek_w * NW + wv], lane_bit);
            a_key[tid] = key | chk << 21;
            a_ev[tid] = ew | ek << 16;
            a_st[tid] = ty | tlen << 8;
            a_succ[tid] = (uint32_t)NL;

            // closure of the token chain inside this wavefront: after the loop every lane knows the positions of its own
            // wavefront its chain passes (mask) and where the chain leaves the wavefront (nxg, a global lane index)
            auto closure = [&]() {
                uint32_t nxg = (uint32_t)tid + tlen;
                u64 mk = live ? lane_bit : 0ull;
                const uint32_t wend = (uint32_t)(64 * (wv + 1) < nlive ? 64 * (wv + 1) : nlive);
                for (int r = 0; r < 6; r++) {                       // 2^6 hops cover a wavefront of one-byte tokens
                    const bool go = live && nxg < wend;
                    if (!__any(go)) break;
                    const int src = (int)(nxg & 63u);
                    const uint32_t n2 = (uint32_t)__shfl((int)nxg, src);
                    const u64 m2 = (u64)(uint32_t)__shfl((int)(uint32_t)(mk >> 32), src) << 32 | (uint32_t)__shfl((int)(uint32_t)mk, src);
                    if (go) { nxg = n2; mk |= m2; }
                }
                c_exit[tid] = nxg; c_mask[tid] = mk;
            };
            u64 tq1 = 0, tq2 = 0;
            if (prof) tq1 = __builtin_readcyclecounter();
            closure();
            __syncthreads();                                        // (B1) tables, arrays, closure are in LDS
            if (prof) { tq2 = __builtin_readcyclecounter(); c_t1 += tq2 - tq1; }

            // exact masks of this lane
            u64 KM[NW], CM[NW], LK1[NW], LC1[NW], LK2[NW], LC2[NW], EKC[NW], EKS[NW];
            {
                const uint32_t lk1 = canm ? wg_key_ix(W.lkey1) : (uint32_t)kWgKeyTab, lc1 = canm ? lctx1 : 256u;
                const uint32_t lk2 = (canm && want2) ? wg_key_ix(W.lkey2) : (uint32_t)kWgKeyTab, lc2 = (canm && want2) ? lctx2 : 256u;
                const uint32_t ekc = live ? ctx : 256u;
#pragma unroll
                for (int w = 0; w < NW; w++) {
                    KM[w] = keytab[kix * NW + w]; CM[w] = ctxtab[ctx_w * NW + w];
                    LK1[w] = keytab[lk1 * NW + w]; LC1[w] = ctxtab[lc1 * NW + w];
                    LK2[w] = keytab[lk2 * NW + w]; LC2[w] = ctxtab[lc2 * NW + w];
                    EKC[w] = ektab[ekc * NW + w]; EKS[w] = ektab[ek_w * NW + w];
                }
                verify_row<NW>(KM, a_key, key, wv, below, canm);
                verify_row<NW>(LK1, a_key, W.lkey1, wv, beloweq, canm);
                if (want2) verify_row<NW>(LK2, a_key, W.lkey2, wv, beloweq, canm);
                if (!canm) {
#pragma unroll
                    for (int w = 0; w < NW; w++) { CM[w] = 0; LC1[w] = 0; LC2[w] = 0; }
                }
                if (!live) {
#pragma unroll
                    for (int w = 0; w < NW; w++) { EKC[w] = 0; EKS[w] = 0; }
                }
            }
            if (prof) t2 = __builtin_readcyclecounter();

            // ---------------- iterate to the fixed point
            u64 S[NW], EV[NW], EFF[NW], MAT[NW];
            int limit = 0;
            bool limit_hard = false;
            bool has_ev = false, cond = false, eff = false;
            uint32_t s0b = 0;
            for (int it = 0;; it++) {
                u64 ta = 0, tb = 0, tc = 0, td = 0, te = 0, tf = 0, tg = 0, th = 0;
                if (prof) ta = __builtin_readcyclecounter();
                if (it > 0) { closure(); __syncthreads(); }
                // chase: one hop per wavefront (every wavefront walks it; the result is uniform)
#pragma unroll
                for (int w = 0; w < NW; w++) S[w] = 0ull;
                {
                    int e = 0;
                    for (int hop = 0; hop < NW && e < nlive; hop++) {
                        const int ew_ = e >> 6;
                        const u64 mk = uni64(c_mask[e]);
                        const int nxe = (int)ufl(c_exit[e]);
#pragma unroll
                        for (int w = 0; w < NW; w++) if (w == ew_) S[w] = mk;
                        e = nxe;
                    }
                }
                const bool inS = ((S[wv] >> lane) & 1ull) != 0;
                if (prof) tb = __builtin_readcyclecounter();
                // E step A: kind of the previous token, boundary events (src/libzling_lz.cpp:163-166, 181-182, 190-191)
                {
                    u64 ones[NW];
#pragma unroll
                    for (int w = 0; w < NW; w++) ones[w] = ~0ull;
                    const int pl = top_in<NW>(ones, S, wv, below);
                    const uint32_t pty = pl >= 0 ? (a_st[pl] & 0xFF) : prevty;
                    has_ev = live && (pty == kTyMatch || pty == kTyLit || pty == kTyW1);
                    cond = pty == kTyMatch;
                    const u64 evb = __ballot(has_ev && inS);
                    if (lane == 0) u_ev[wv] = evb;
                }
                __syncthreads();                                    // (B2)
                if (prof) tc = __builtin_readcyclecounter();
                if (it == 0) {                                      // every wavefront has read its rows: each lane clears what it set
                    keytab[kix * NW + wv] = 0; ctxtab[ctx_w * NW + wv] = 0; ektab[ek_w * NW + wv] = 0;
                }
#pragma unroll
                for (int w = 0; w < NW; w++) EV[w] = uni64(u_ev[w]);
                // E step B: slot 0 of my event key just before my event; is my push effective?
                {
                    const int e = top_in<NW>(EKS, EV, wv, below);
                    const uint32_t m0e = mru[ek_w & 255u];
                    s0b = e >= 0 ? (a_ev[e] & 0xFFFF) : (m0e & 0xFFFF);
                    eff = has_ev && (!cond || ew != s0b);
                    a_s0b[tid] = s0b;
                    const u64 efb = __ballot(eff && inS);
                    if (lane == 0) u_eff[wv] = efb;
                }
                __syncthreads();                                    // (B3)
                if (prof) td = __builtin_readcyclecounter();
#pragma unroll
                for (int w = 0; w < NW; w++) EFF[w] = uni64(u_eff[w]);
                // E step C: my match given the accepted starts before me
                bool hard = false;
                uint32_t hcls = 0; (void)hcls;
                bool is_match = false;
                uint32_t ml = kMatchMin - 1, mn = 0;
                int lk = -1;
                uint32_t lkchk = 0;
                const uint32_t k = cnt_in<NW>(CM, S, wv, below);
                k_ctx = k;
                if (level0) {
                    int a1 = top_in<NW>(KM, S, wv, below), a2 = -1;
                    if (__any(a1 >= 0)) {
                        u64 K2[NW];
#pragma unroll
                        for (int w = 0; w < NW; w++) K2[w] = (a1 >= 0 && (a1 >> 6) == w) ? (KM[w] & ~(1ull << (a1 & 63))) : KM[w];
                        a2 = a1 >= 0 ? top_in<NW>(K2, S, wv, below) : -1;
                    }
                    const bool ring0 = W.has0 && W.d0 <= k, ring1 = W.has1 && W.d1 <= k;
                    if (canm) {
                        if (a1 >= 0 ? (a2 < 0 && ring0) : ring0) { hard = true; hcls = 2; }
                        else if (a1 < 0 && ring1) {
                            // node 1's slot was rewritten by a start of this round: it holds a later position than node 0's now, so
                            // the reference's chain-end test (src/libzling_lz.cpp:265) stops the walk behind node 0
                            ml = W.len0 > 3u ? W.len0 : 3u; mn = W.node0; is_match = ml >= (uint32_t)kMatchMin;
                        } else if (a1 < 0) { is_match = W.len >= (uint32_t)kMatchMin; ml = W.len; mn = W.node; }
                    }
                    const bool fixl = canm && !hard && a1 >= 0;
                    if (__any(fixl)) {
                        // the chain is [a1, a2 | the snapshot's head] (depth 2): candidate lengths from the window's own text
                        const uint32_t k1 = a_key[fixl ? a1 : tid], k2 = a_key[(fixl && a2 >= 0) ? a2 : tid];
                        const bool c1 = fixl && (k1 >> 21) == chk, c2 = fixl && a2 >= 0 && (k2 >> 21) == chk;
                        const uint32_t l1 = lcp_with(buf, upos, (uint32_t)(P + (c1 ? a1 : 0)), qtext, c1);
                        uint32_t l2 = 0;
                        if (__any(c2)) l2 = lcp_with(buf, upos, (uint32_t)(P + (c2 ? a2 : 0)), qtext, c2);
                        if (fixl) {
                            const bool second = a2 >= 0 || W.has0;
                            const uint32_t ls = a2 >= 0 ? l2 : W.len0, ns = a2 >= 0 ? (0x10000u | (uint32_t)a2) : W.node0;
                            ml = kMatchMin - 1; mn = 0;
                            if (l1 > ml) { ml = l1; mn = 0x10000u | (uint32_t)a1; }
                            if (ml != (uint32_t)kMatchMax && second && ls > ml) { ml = ls; mn = ns; }
                            is_match = ml >= (uint32_t)kMatchMin;
                            lk = a1; lkchk = k1 >> 21;
                        }
                    }
                    if (prof) te = __builtin_readcyclecounter();
                    // the lazy probe under ml (src/libzling_lz.cpp:270-281, 291-316; depth 1: only the chain head is looked at)
                    if (canm && !hard && is_match && ml < (uint32_t)kLazyLimit) {
                        u64 Sx[NW];
#pragma unroll
                        for (int w = 0; w < NW; w++) Sx[w] = S[w] | (w == wv ? lane_bit : 0ull);
                        const int lh = top_in<NW>(LK1, Sx, wv, beloweq);
                        const bool lconf = cnt_in<NW>(LC1, Sx, wv, beloweq) > W.ld1;     // a visited slot at distance d is rewritten by the (d+1)-th insert
                        bool veto;
                        if (lconf) { hard = true; hcls = 3; veto = false; }
                        else if (lh < 0 && ml == W.len) veto = W.veto1;
                        else {
                            const uint32_t mm = ml - 3u;
                            const uint32_t pr = ld32u(buf + (upos + 1u + mm));
                            const uint32_t so = lh >= 0 ? (uint32_t)(P + lh) : (W.lsrc1 & 0xFFFFFF);
                            const uint32_t sr = ld32u(buf + (so + mm));
                            veto = (lh >= 0 || (W.lsrc1 >> 31) != 0) && pr == sr;
                        }
                        if (veto) is_match = false;
                    }
                } else if (canm) {
                    // levels 1-4: a start of this round in my hash slot, a rewritten ring slot or a touched lazy read set -> hard
                    const bool kq = top_in<NW>(KM, S, wv, below) >= 0;
                    if (kq || W.dmin <= k) { hard = true; hcls = kq ? 1 : 2; }
                    else {
                        is_match = W.len >= (uint32_t)kMatchMin; ml = W.len; mn = W.node;
                        if (is_match && ml < (uint32_t)kLazyLimit) {
                            u64 Sx[NW];
#pragma unroll
                            for (int w = 0; w < NW; w++) Sx[w] = S[w] | (w == wv ? lane_bit : 0ull);
                            const bool c1 = want1 && (top_in<NW>(LK1, Sx, wv, beloweq) >= 0 || cnt_in<NW>(LC1, Sx, wv, beloweq) > W.ld1);
                            if (c1) { hard = true; hcls = 3; }
                            else if (want1 && W.veto1) is_match = false;
                            else if (want2) {
                                const bool c2 = top_in<NW>(LK2, Sx, wv, beloweq) >= 0 || cnt_in<NW>(LC2, Sx, wv, beloweq) > W.ld2;
                                if (c2) { hard = true; hcls = 3; }
                                else if (W.veto2) is_match = false;
                            }
                        }
                    }
                }
                if (prof) tf = __builtin_readcyclecounter();
                // word MRU of my context after every boundary event of S up to and including mine (src/libzling_lz.cpp:172-185)
                uint32_t ty2, tlen2;
                if (hard) { ty2 = ty; tlen2 = tlen; }
                else if (is_match) { ty2 = kTyMatch; tlen2 = ml; }
                else {
                    u64 EVx[NW], EFx[NW];
#pragma unroll
                    for (int w = 0; w < NW; w++) { EVx[w] = EV[w] | ((w == wv && has_ev) ? lane_bit : 0ull); EFx[w] = EFF[w] | ((w == wv && eff) ? lane_bit : 0ull); }
                    const int e1 = top_in<NW>(EKC, EVx, wv, beloweq), e2 = top_in<NW>(EKC, EFx, wv, beloweq);
                    const uint32_t m0 = mru[ctx];
                    const uint32_t s0 = e1 >= 0 ? (a_ev[e1] & 0xFFFF) : (m0 & 0xFFFF);
                    const uint32_t s1 = e2 >= 0 ? (e2 == tid ? s0b : a_s0b[e2]) : (m0 >> 16);
                    const bool two = live && pos + 1 < ilen;
                    ty2 = (two && s0 == cw) ? kTyW0 : ((two && s1 == cw) ? kTyW1 : kTyLit);
                    tlen2 = ty2 == kTyLit ? 1u : 2u;
                }
                const bool chg = inS && (ty2 != ty || tlen2 != tlen);
                ty = ty2; tlen = tlen2;
                if (!hard) { mlen = ml; mnode = mn; link = lk; link_chk = lkchk; }
                {
                    const u64 hb = __ballot(hard && inS), cb = __ballot(chg), mb = __ballot(ty == kTyMatch && inS);
                    a_st[tid] = ty | tlen << 8;
                    a_succ[tid] = (uint32_t)NL;
                    if (lane == 0) { u_hard[wv] = hb; u_chg[wv] = cb; u_mat[wv] = mb; }
                }
                if (prof) tg = __builtin_readcyclecounter();
                __syncthreads();                                    // (B4)
                u64 HB[NW], CB[NW];
#pragma unroll
                for (int w = 0; w < NW; w++) { HB[w] = uni64(u_hard[w]); CB[w] = uni64(u_chg[w]); MAT[w] = uni64(u_mat[w]); }
                // cut of the round: the first hard lane of S ...
                limit = nlive; limit_hard = false;
#pragma unroll
                for (int w = NW - 1; w >= 0; w--) if (HB[w]) { limit = 64 * w + (int)__builtin_ctzll(HB[w]); limit_hard = true; }
                // ... or the end of the sub-block (src/libzling_lz.cpp:153): token j may start while opos + 1 < kSubSyms
                {
                    uint32_t tot = 0;
#pragma unroll
                    for (int w = 0; w < NW; w++) {
                        const u64 lm = 64 * w + 64 <= limit ? ~0ull : (64 * w >= limit ? 0ull : ((1ull << (limit - 64 * w)) - 1ull));
                        tot += (uint32_t)__popcll(S[w] & lm) + (uint32_t)__popcll(MAT[w] & lm);
                    }
                    if ((uint32_t)opos + tot + 1u >= (uint32_t)kSubSyms) {
                        u64 ones[NW];
#pragma unroll
                        for (int w = 0; w < NW; w++) ones[w] = ~0ull;
                        const uint32_t before = cnt_in<NW>(ones, S, wv, below) + cnt_in<NW>(ones, MAT, wv, below);
                        const u64 vb = __ballot(inS && !((uint32_t)opos + before + 1u < (uint32_t)kSubSyms));
                        if (lane == 0) u_cut[wv] = vb;
                        __syncthreads();
#pragma unroll
                        for (int w = NW - 1; w >= 0; w--) {
                            const u64 v = uni64(u_cut[w]);
                            if (v) { const int c = 64 * w + (int)__builtin_ctzll(v); if (c <= limit) { limit = c; limit_hard = false; } }
                        }
                        __syncthreads();                            // u_cut is free again
                        if (prof) n_cutr++;
                    }
                }
                bool changed = false;
#pragma unroll
                for (int w = 0; w < NW; w++) {
                    const u64 lm = 64 * w + 64 <= limit ? ~0ull : (64 * w >= limit ? 0ull : ((1ull << (limit - 64 * w)) - 1ull));
                    changed = changed || (CB[w] & lm) != 0ull;
                }
                if (prof) { n_iter++; th = __builtin_readcyclecounter(); c_cl += tb - ta; c_a += tc - tb; c_b += td - tc; c_c1 += te - td; c_c2 += tf - te; c_c3 += tg - tf; c_lim += th - tg; }
                if (!changed) break;
                if (it > 2 * NL) { overflow = true; break; }        // cannot happen: every iteration fixes at least one lane of S
            }
            if (prof) t3 = __builtin_readcyclecounter();

            // ---------------- commit S below the limit
            u64 C[NW];
#pragma unroll
            for (int w = 0; w < NW; w++) {
                const u64 lm = 64 * w + 64 <= limit ? ~0ull : (64 * w >= limit ? 0ull : ((1ull << (limit - 64 * w)) - 1ull));
                C[w] = S[w] & lm;
            }
            const bool mine = ((C[wv] >> lane) & 1ull) != 0;
            // who links to whom in a hash slot: the slot's head must end up being the LAST start of the round in it
            if (mine && canm && link >= 0) atomicMin(&a_succ[link], (uint32_t)tid);
            __syncthreads();                                        // (B5)
            uint32_t ncom = 0, nmat = 0;
#pragma unroll
            for (int w = 0; w < NW; w++) { ncom += (uint32_t)__popcll(C[w]); nmat += (uint32_t)__popcll(C[w] & MAT[w]); }
            if (mine) {
                u64 ones[NW];
#pragma unroll
                for (int w = 0; w < NW; w++) ones[w] = ~0ull;
                const uint32_t rank = cnt_in<NW>(ones, C, wv, below);
                uint32_t word;
                if (canm) {
                    // dictionary insert (src/libzling_lz.cpp:227-230); slots of a bucket are handed out in position order
                    const uint32_t head = (head0 + k_ctx + 1u) & (kRing - 1);
                    BucketT<kWide> B(dict, ctx);
                    uint32_t lslot = W.node0, pword = W.ov0;
                    if (link >= 0) { lslot = (head0 + cnt_below_lane<NW>(CM, C, link) + 1u) & (kRing - 1); pword = (uint32_t)(P + link) | link_chk << 24; }
                    B.suffix[head] = (uint16_t)lslot;
                    if (kWide) B.slot[head] = (u64)((uint32_t)pos | chk << 24) | (u64)pword << 32;
                    else B.offset[head] = (uint32_t)pos | chk << 24;
                    if (a_succ[tid] >= (uint32_t)limit) B.hash[hc] = (uint16_t)head;
                    if (!any_above<NW>(CM, C, wv, above)) heads[ctx] = (uint16_t)head;
                    uint32_t msl = mnode;
                    if (mnode & 0x10000u) msl = (head0 + cnt_below_lane<NW>(CM, C, (int)(mnode & 0xFFFFu)) + 1u) & (kRing - 1);
                    word = (258u + mlen - kMatchMin) | ((head - msl) & (kRing - 1)) << 16;
                } else word = 0;
                if (ty == kTyW0) word = 256; else if (ty == kTyW1) word = 257; else if (ty == kTyLit) word = b_0 | ctx << 16;
                __builtin_nontemporal_store(word, &tok[nt + rank]);
                // MRU slots of my event key after the round: written by the key's last event
                if (has_ev) {
                    u64 EVc[NW], EFc[NW];
#pragma unroll
                    for (int w = 0; w < NW; w++) { EVc[w] = EV[w] & C[w]; EFc[w] = EFF[w] & C[w]; }
                    if (!any_above<NW>(EKS, EVc, wv, above)) {
                        const int e2 = top_in<NW>(EKS, EFc, wv, beloweq);
                        const uint32_t m0e = mru[ek];
                        mru[ek] = ew | (e2 >= 0 ? (e2 == tid ? s0b : a_s0b[e2]) : (m0e >> 16)) << 16;
                    }
                }
            }
            // round summary (uniform): the last committed token leads on
            if (ncom) {
                int lastl = 0;
#pragma unroll
                for (int w = 0; w < NW; w++) if (C[w]) lastl = 64 * w + top_bit(C[w]);
                const uint32_t st = ufl(a_st[lastl]);
                q = P + lastl + (int)(st >> 8);
                prevty = st & 0xFF;
                nt += ncom; opos += (int)(ncom + nmat);
            }
            serial_next = limit_hard;
            if (ncom == 0 && !limit_hard) overflow = true;          // cannot happen: a round commits a token or names a hard lane
            __syncthreads();                                        // (B6) inserts, heads, MRU are visible; per-lane arrays are free again
            if (prof) {
                const u64 t4 = __builtin_readcyclecounter();
                c_p1 += t1 - t0; c_tab += t2 - t1; c_it += t3 - t2; c_com += t4 - t3; n_round++; n_pos += (u64)(q - P);
                if (limit_hard) n_hard[0]++;
            }
        }
        if (nsub < kMaxSub && tid == 0) cuts[nsub] = SubCut{tok_begin, nt, (uint32_t)q, (uint32_t)opos};
        nsub++;
    }
    if (tid == 0) {
        if (overflow) { *a.overflow = 1; nsub = 0; nt = 0; }
        a.nsub[blk] = (uint32_t)nsub; a.ntok[blk] = nt;
    }
    if (prof && tid == 0) {
        u64* d = a.dbg + (size_t)blk * kDbgSlots;
        d[0] = c_p1; d[1] = c_tab; d[2] = c_it; d[3] = n_round; d[4] = nt; d[5] = n_iter; d[6] = n_ser; d[7] = n_pos; d[8] = c_ser; d[9] = c_com;
        d[10] = n_hard[0]; d[11] = n_cutr; d[12] = c_cl; d[13] = c_a; d[14] = c_b; d[15] = c_c1; d[16] = c_c2; d[17] = c_c3; d[18] = c_lim; d[19] = c_t1;
    }
}

void launch_rolz_parse_wg(const ParseArgs& a, uint32_t nblocks_all, hipStream_t s, bool all_level0, int nw) {
    const bool prof = a.dbg != nullptr;
    const uint32_t nblocks = nblocks_all - a.blk0;
#define ZLNG_WG_LAUNCH(NW)                                                                                                   \
    do {                                                                                                                     \
        if (all_level0 && !prof) hipLaunchKernelGGL((k_rolz_parse_wg<NW, true, false>), dim3(nblocks), dim3(64 * NW), 0, s, a);   \
        else if (all_level0) hipLaunchKernelGGL((k_rolz_parse_wg<NW, true, true>), dim3(nblocks), dim3(64 * NW), 0, s, a);        \
        else if (!prof) hipLaunchKernelGGL((k_rolz_parse_wg<NW, false, false>), dim3(nblocks), dim3(64 * NW), 0, s, a);           \
        else hipLaunchKernelGGL((k_rolz_parse_wg<NW, false, true>), dim3(nblocks), dim3(64 * NW), 0, s, a);                       \
    } while (0)
    if (nw <= 2) ZLNG_WG_LAUNCH(2);
    else ZLNG_WG_LAUNCH(4);
#undef ZLNG_WG_LAUNCH
}

}  // namespace zlng
