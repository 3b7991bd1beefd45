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
//   tables    every (ctx, hash13) of the window gets a row of an LDS table of its own (open addressing, exact); buckets and
//             word-MRU keys index their tables directly.
//   iterate   S = the token starts reached from lane 0 under the current per-lane token lengths (pointer doubling inside a
//             wavefront, one hop per wavefront across them), at most 64 per round, numbered by RANK.  The tokens of S put their
//             rank bit into the rows of their hash slot, bucket and MRU key, so that every relation a token needs is ONE 64-bit
//             mask.  Then E(g, S): every token of S re-evaluates itself GIVEN the tokens before it --
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


__device__ __forceinline__ uint32_t wg_key_ix(uint32_t key21) { return ((key21 & 0x1FFFu) ^ ((key21 >> 13) * 0x9E5u) ^ (key21 >> 7)) & 2047u; }
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
    uint32_t pre1, pre2;         // levels 1-4: best (len | node << 9) over the first depth-1 / depth-2 chain nodes
    uint32_t vpos1, vpos2;       // levels 1-4: index of the first node of a probe's chain that vetoes (its depth if none)
    uint32_t d0g, tail0, tail1, tail2, ntail;   // levels 1-4, ring rule: rolz_dev.h Spec
};

// Level 0 (depth 2, one lazy probe of depth 1; src/libzling_lz.cpp:130), straight-line predicated code: three dependent round
// trips after the window's text with the wide slot plane (rolz_dev.h speculate_l0w explains the link copy), then -- only if some
// lane's compare ran to 16 bytes -- the tails (32 bytes per trip, both nodes in one loop) and the lazy probes of those lanes.
// kHot: the bucket of ONE context (`hotctx`, the block's most frequent byte) is mirrored in LDS (`hot`: the wide layout of
// zlng_common.h, 57,344 B, written through by every insert); lanes of that context -- 40 % of a text's positions -- take their hash
// head, ring slot and link from LDS, and their global loads are pointed at one shared line instead of 64 scattered ones.
// round trip 1 of speculate_l0t, both hash heads: asked for by the caller ahead of work that does not depend on them (the row claims)
template <bool kWide, bool kHot>
__device__ __forceinline__ void speculate_l0t_heads(uint8_t* dict, const Quad qa, uint32_t ctx, uint32_t hc, uint32_t hotctx, uint32_t& node0, uint32_t& ln1) {
    const uint32_t w4 = qa.a;
    const uint32_t lctx1 = w4 & 0xFF;
    const uint32_t hh1 = hash_of(w4 >> 8 | qa.b << 24) % kHashSlots;
    BucketT<kWide> B(dict, ctx), B1(dict, lctx1);
    node0 = B.hash[(kHot && ctx == hotctx) ? 0u : hc];
    ln1 = B1.hash[(kHot && lctx1 == hotctx) ? 0u : hh1];
}

template <bool kWide, bool kHot>
__device__ __forceinline__ void speculate_l0t(WSpec& W, Quad& ql_out, uint8_t* dict, const uint8_t* buf, uint32_t head0, uint32_t lhead1, uint32_t pos,
                                              const Quad qa, uint32_t t16, uint32_t ctx, uint32_t hc, uint32_t chk,
                                              const uint8_t* hot, uint32_t hotctx, uint32_t node0, uint32_t ln1) {
    const uint32_t w4 = qa.a;
    const uint32_t lctx1 = w4 & 0xFF;
    const uint32_t hh1 = hash_of(w4 >> 8 | qa.b << 24) % kHashSlots;
    BucketT<kWide> B(dict, ctx), B1(dict, lctx1);
    const bool h0 = kHot && ctx == hotctx, h1 = kHot && lctx1 == hotctx;
    const uint16_t* hot_hash = reinterpret_cast<const uint16_t*>(hot + 8u * kRing + 2u * kRing);
    const uint16_t* hot_sfx = reinterpret_cast<const uint16_t*>(hot + 8u * kRing);
    const u64* hot_slot = reinterpret_cast<const u64*>(hot);
    if (kHot) { const uint32_t a0 = hot_hash[h0 ? hc : 0u], a1 = hot_hash[h1 ? hh1 : 0u]; node0 = h0 ? a0 : node0; ln1 = h1 ? a1 : ln1; }
    const bool has0 = node0 != 65535u, hasl = ln1 != 65535u;
    // round trip 2: node 0's slot (wide: own word + its link's), its link, the probe node's word
    uint32_t ov0, nov = 0;
    if (kWide) { const u64 sl0 = B.slot[h0 ? 0u : (node0 & (kRing - 1))]; ov0 = (uint32_t)sl0; nov = (uint32_t)(sl0 >> 32); }
    else ov0 = B.offset[node0 & (kRing - 1)];
    uint32_t nx = B.suffix[h0 ? 0u : (node0 & (kRing - 1))];
    uint32_t lov1 = B1.offset[h1 ? 0u : (ln1 & (kRing - 1))];
    if (kHot) {
        const u64 s0 = hot_slot[h0 ? (node0 & (kRing - 1)) : 0u], s1 = hot_slot[h1 ? (ln1 & (kRing - 1)) : 0u];
        const uint32_t x0 = hot_sfx[h0 ? (node0 & (kRing - 1)) : 0u];
        if (h0) { ov0 = (uint32_t)s0; nov = (uint32_t)(s0 >> 32); nx = x0; }
        if (h1) lov1 = (uint32_t)s1;
    }
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
    ql_out = ql;                                      // first 16 source bytes of the probe's chain head (a changed length re-probes from them)
}

// The 4 bytes at byte offset `off` (0..13, or ..16 with `next` = the dword behind the quad) of a 16-byte group.
__device__ __forceinline__ uint32_t byte_window(const Quad q, uint32_t next, uint32_t off) {
    const uint32_t dw = off >> 2;
    const uint32_t lo = dw == 0u ? q.a : dw == 1u ? q.b : dw == 2u ? q.c : dw == 3u ? q.d : next;
    const uint32_t hi = dw == 0u ? q.b : dw == 1u ? q.c : dw == 2u ? q.d : next;
    return __builtin_amdgcn_alignbyte(hi, lo, off & 3u);
}

// GetCommonLength (src/libzling_lz.cpp:66-89) of the lane's position (first 16 bytes in qa) with another position of the block.
__device__ __forceinline__ uint32_t lcp_with(const uint8_t* buf, uint32_t pos, uint32_t src, const Quad qa, bool act) {
    const Quad qb = ld128u(buf + (act ? src : pos));
    uint32_t len = act ? lcp16(qa, qb) : 0u;
    const bool lng = act && len == 16u;
    if (__any(lng)) { const uint32_t t = lcp_tail(buf + pos, buf + src, lng); len = lng ? t : len; }
    return len;
}

constexpr int kWgRows = 2048;                       // rows of the exact (ctx, hash13) table (open addressing over the keys of one round)
constexpr uint32_t kWgEmpty = 0xFFFFFFFFu;

// The two MRU slots of a key after the boundary events whose token ranks are in M (newest = highest rank), given the key's slots
// m0 at the start of the round (src/libzling_lz.cpp:163-166, 181-182, 190-191).  Slot 0 is the word of the newest event: a
// conditional push leaves it that word whether it pushes or not.  Slot 1 is what slot 0 was just before the newest EFFECTIVE
// event (unconditional, or conditional with a word that differs from slot 0 at that moment).  t_ev[r] = word of token r's event,
// COND = tokens whose event is conditional (it follows a match).  false: not settled within `bound` events (the lane goes hard).
__device__ __forceinline__ bool ev_state(u64 M, u64 COND, uint32_t m0, const uint32_t* t_ev, uint32_t& s0, uint32_t& s1, int bound) {
    s0 = m0 & 0xFFFF; s1 = m0 >> 16;
    if (M == 0ull) return true;
    int e = top_bit(M);
    uint32_t w = t_ev[e] & 0xFFFF;
    s0 = w;
    u64 cur = M & ~(1ull << e);
    for (int i = 0; i < bound; i++) {
        const uint32_t pw = cur ? (t_ev[top_bit(cur)] & 0xFFFF) : (m0 & 0xFFFF);
        const bool eff = !((COND >> e) & 1ull) || w != pw;
        if (eff) { s1 = pw; return true; }
        if (!cur) return true;                       // no effective event in the round: slot 1 as it was
        e = top_bit(cur); cur &= ~(1ull << e); w = pw;
    }
    return false;
}

// One workgroup per block.  NW wavefronts; lane g = 64 * wavefront + lane stands for position P + g.  The fixed-point iteration
// works on the TOKENS of S, numbered by rank (their order in S; at most 64 per round): every relation between tokens -- same hash
// slot, same bucket, same MRU key -- is one 64-bit rank mask in LDS, rebuilt per iteration by the tokens of S themselves.
// kRingRule: the ring rule of levels 1-4 (ZLNG_RING_FIX=1; see E below) as an instantiation of its own -- built in round 5, exact in the CPU
// model, never run on a GPU.  With kRingRule = false none of its code is in the kernel: the default instantiations are, instruction for
// instruction, the kernels that last ran on one (scripts/isa_diff.py, tests/test_isa_hygiene.py).
template <int NW, bool kAllL0, bool kProf, bool kWide, bool kHot = false, bool kRingRule = false>
__global__ __launch_bounds__(64 * NW) void k_rolz_parse_wg(ParseArgs a) {
    constexpr int NL = 64 * NW;
    static_assert(!kRingRule || !kAllL0, "the ring rule belongs to the generic-level walk");
    static_assert(!kHot || (kWide && kAllL0), "the LDS bucket mirrors the wide layout of a level-0 context");
    __shared__ u64 hot_lds[kHot ? kBktBytes / 8 : 1];     // kHot: write-through mirror of one bucket (57,344 B)
    __shared__ uint32_t hot_hist[kHot ? 256 : 1];
    __shared__ uint16_t heads[256];
    __shared__ uint32_t mru[256];                    // slot0 | slot1 << 16
    __shared__ uint32_t ht_key[kWgRows];             // exact key of a row of keyrow (open addressing; kWgEmpty = free)
    __shared__ u64 keyrow[2][kWgRows + 1];           // tokens of S per (ctx, hash13); two buffers alternate between iterations;
    __shared__ u64 ctxrow[2][257];                   // tokens of S per bucket                  the extra last row is a sink
    __shared__ u64 ekrow[2][257];                    // tokens of S per MRU event key
    __shared__ uint32_t a_st[NL];                    // per lane: token kind | token length << 8 (what the chain was built from)
    __shared__ uint32_t c_exit[NL];                  // closure: first chain position beyond the lane's wavefront (global lane index)
    __shared__ u64 c_mask[NL];                       // closure: chain positions inside the lane's wavefront
    __shared__ uint32_t t_lane[64], t_ev[64], t_key[64];   // per token of S: its lane, event word | key << 16, key21 | chk << 21
    __shared__ Quad t_q[64];                         // per token of S: the 16 bytes at its position
    __shared__ uint32_t t_flag[64];                  // per token of S: has an event | conditional << 1
    __shared__ uint32_t t_res[64];                   // per token of S, after E: hard << 2 | changed << 3 | match << 4
    __shared__ u64 u_cut;
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
    const u64 lbelow = lane_bit - 1ull;

    for (int i = tid; i < 2 * 257; i += NL) { (&ctxrow[0][0])[i] = 0; (&ekrow[0][0])[i] = 0; }
    for (int i = tid; i < 2 * (kWgRows + 1); i += NL) (&keyrow[0][0])[i] = 0;
    for (int i = tid; i < kWgRows; i += NL) ht_key[i] = kWgEmpty;
    for (int i = tid; i < 256; i += NL) heads[i] = 0;
    uint32_t hotctx = 256u;                          // (no context)
    uint8_t* hot = reinterpret_cast<uint8_t*>(hot_lds);
    if (kHot) {
        // the block's most frequent byte over its first 64 KiB is the context whose bucket lives in LDS; the mirror starts as Reset() leaves a bucket
        for (int i = tid; i < 256; i += NL) hot_hist[i] = 0;
        for (int i = tid; i < (int)(kBktBytes / 8); i += NL) hot_lds[i] = i < (int)kRing ? 0ull : ~0ull;
        __syncthreads();
        const int span = ilen < 65536 ? ilen : 65536;
        for (int i = tid; i < span; i += NL) atomicAdd(&hot_hist[buf[i]], 1u);
        __syncthreads();
        uint32_t best = 0, bestc = 0;
        for (int i = 0; i < 256; i++) { const uint32_t v = hot_hist[i]; if (v > best) { best = v; bestc = (uint32_t)i; } }
        hotctx = ufl(bestc);
    }
    __syncthreads();

    uint32_t nt = 0;
    int q = 0, nsub = 0;
    bool overflow = false;
    bool internal = false;                           // a "cannot happen" guard fired: reported as its own flag value (2), not as an exhausted pool
    int pf_P = -1;                                   // start of the window whose text pf_* hold (-1: none)
    uint32_t pf_wraw = 0, pf_t16 = 0;
    Quad pf_q = {0, 0, 0, 0};
    u64 c_p1 = 0, c_tab = 0, c_it = 0, c_com = 0, c_ser = 0, n_round = 0, n_iter = 0, n_ser = 0, n_hardr = 0, n_pos = 0, n_cutr = 0;
    u64 c_dep = 0, c_ev = 0, c_lim = 0, c_x1 = 0, c_x2 = 0, c_x3 = 0;
    const bool prof = kProf && a.dbg != nullptr;
    u64 c_mk[12] = {0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0}, mk_last = 0;
#define ZLNG_MK(i) do { if (prof) { const u64 t_ = __builtin_readcyclecounter(); c_mk[i] += t_ - mk_last; mk_last = t_; } } while (0)

    while (q < ilen && !overflow) {                  // ---- one sub-block (one EncodeImpl call)
        const uint32_t lvl = kAllL0 ? 0u : (uint32_t)a.lvl_sched[blk * kMaxSub + (nsub < kMaxSub ? nsub : kMaxSub - 1)];
        const LevelCfg cfg = level_cfg((int)lvl);
        const bool level0 = kAllL0 || lvl == 0u;
        const bool want1 = cfg.lazy1 > 0, want2 = cfg.lazy2 > 0;
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
            // (a pool of one word per input byte cannot run out: no check, no false alarm on a block of one-byte tokens)
            if (a.tok_cap < kTokCapMax && nt + 64u > a.tok_cap) { overflow = true; break; }

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
                        if (kHot && cq == hotctx && lane == 0) {                     // the same insert into the LDS mirror (MatchAndUpdate, src/libzling_lz.cpp:227-230)
                            const uint32_t hq = hash_of(ld32u(buf + (uint32_t)q));
                            const uint32_t hcq = hq % kHashSlots, own = (uint32_t)q | ((hq / kHashSlots) & 255u) << 24;
                            uint16_t* hh = reinterpret_cast<uint16_t*>(hot + 8u * kRing + 2u * kRing);
                            const uint32_t node = hh[hcq];
                            const uint32_t lw = (node == 65535u || node == head) ? own : (uint32_t)hot_lds[node & (kRing - 1)];
                            reinterpret_cast<uint16_t*>(hot + 8u * kRing)[head] = (uint16_t)node;
                            hot_lds[head] = (u64)own | (u64)(node == head ? own : (node == 65535u ? 0u : lw)) << 32;
                            hh[hcq] = (uint16_t)head;
                        }
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
            // the window's text: asked for at the end of the previous round, in front of that round's last barrier, so that the
            // round trip runs beside that barrier and this round's prologue instead of behind them (measured: asking even earlier,
            // in front of the commit's stores, adds nothing -- the acknowledgements of the stores are not what the load waits for)
            uint32_t wraw, t16;
            Quad qtext;
            if (P == pf_P) { wraw = pf_wraw; qtext = pf_q; t16 = pf_t16; }
            else {
                wraw = ld32u(buf + (upos >= 4u ? upos - 4u : 0u));
                qtext = ld128u(buf + upos);
                t16 = ld32u(buf + (upos + 16u));
            }
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
            const uint32_t lctx1 = w4 & 0xFF, lctx2 = (w4 >> 8) & 0xFF;
            const uint32_t lkey1 = lctx1 << 13 | (hash_of(w4 >> 8 | qtext.b << 24) % kHashSlots);
            const uint32_t lkey2 = lctx2 << 13 | (hash_of(w4 >> 16 | qtext.b << 16) % kHashSlots);
            Quad ql = {0, 0, 0, 0};
            // round trip 1 (the hash heads) is requested first; the exact table rows of the round's keys -- LDS work that needs
            // nothing from the dictionary -- are claimed while it is in flight
            uint32_t hd0 = 65535u, hd1 = 65535u, hd2 = 65535u;
            if (level0) speculate_l0t_heads<kWide, kHot>(dict, qtext, ctx, hc, hotctx, hd0, hd1);
            else if (canm) speculate_heads(dict, cfg, qtext, ctx, hc, hd0, hd1, hd2);
            // ---------------- exact table rows of the round's keys (open addressing: a row belongs to ONE key)
            auto claim_row = [&](uint32_t k21, bool want) -> uint32_t {
                uint32_t slot = wg_key_ix(k21), r = kWgRows;
                bool go = want;
                while (__any(go)) {
                    if (go) {
                        const uint32_t old = atomicCAS(&ht_key[slot], kWgEmpty, k21);
                        if (old == kWgEmpty || old == k21) { r = slot; go = false; }
                        else slot = (slot + 1u) & (kWgRows - 1);
                    }
                }
                return r;
            };
            const uint32_t r_key = claim_row(key, canm);
            // the lazy probes' keys get rows too (one that no position of the window has stays empty)
            const uint32_t r_lk1 = claim_row(lkey1, canm && want1);
            const uint32_t r_lk2 = want2 ? claim_row(lkey2, canm) : (uint32_t)kWgRows;
            if (level0) {
                speculate_l0t<kWide, kHot>(W, ql, dict, buf, heads[ctx], heads[lctx1], upos, qtext, t16, ctx, hc, chk, hot, hotctx, hd0, hd1);
            } else {
                Spec S1;
                S1.sp = kMatchMin - 1; S1.node0 = 65535; S1.head0 = 0; S1.dmin = kRing - 1;
                S1.lkix1 = S1.lkix2 = S1.lctx1 = S1.lctx2 = 0; S1.lz1 = S1.lz2 = false; S1.ld1 = S1.ld2 = kRing - 1; S1.ov0 = 0;
                S1.pre1 = S1.pre2 = kMatchMin - 1; S1.vpos1 = S1.vpos2 = 0;
                if constexpr (kRingRule) { S1.d0g = kRing - 1; S1.tail0 = S1.tail1 = S1.tail2 = S1.ntail = 0; }
                if (canm) speculate_from(S1, dict, buf, heads[ctx], heads[lctx1], heads[lctx2], 0u, pos, cfg, qtext, ctx, hc, chk, hd0, hd1, hd2, kRingRule);
                W.len = S1.sp & kSpLenMask; W.node = (S1.sp >> kSpNodeShift) & (kRing - 1);
                W.node0 = S1.node0; W.ov0 = S1.ov0; W.dmin = S1.dmin; W.d0 = W.d1 = S1.dmin;
                W.has0 = S1.node0 != 65535u; W.has1 = false; W.len0 = 0;
                W.veto1 = (S1.sp & kSpVeto1) != 0; W.veto2 = (S1.sp & kSpVeto2) != 0; W.lz1 = S1.lz1; W.lz2 = S1.lz2;
                W.lkey1 = lkey1;
                W.lkey2 = lkey2;
                W.ld1 = S1.ld1; W.ld2 = S1.ld2; W.lsrc1 = 0;
                W.pre1 = S1.pre1; W.pre2 = S1.pre2; W.vpos1 = S1.vpos1; W.vpos2 = S1.vpos2;
                if constexpr (kRingRule) { W.d0g = S1.d0g; W.tail0 = S1.tail0; W.tail1 = S1.tail1; W.tail2 = S1.tail2; W.ntail = S1.ntail; }
            }
            const uint32_t head0 = heads[ctx];
            const uint32_t m0c = mru[ctx], m0e = mru[ek];           // MRU slots of my check key / my event key at the start of the round
            // speculative token of this lane
            const bool sp_veto = (want1 && W.veto1) || (want2 && W.veto2);
            const bool sp_match = canm && W.len >= (uint32_t)kMatchMin && !(W.len < (uint32_t)kLazyLimit && sp_veto);
            uint32_t ty = sp_match ? kTyMatch : kTyLit;            // token kind / length / match of this lane under the current S
            uint32_t tlen = sp_match ? W.len : 1u;
            uint32_t mlen = W.len, mnode = W.node;                  // mnode: ring slot, or 0x10000 | rank for a token of this round
            int link = -1;                                          // rank of my in-slot predecessor among the tokens of this round (-1: the snapshot's head)
            uint32_t link_chk = 0, link_lane = 0;
            if (prof) { t1 = __builtin_readcyclecounter(); mk_last = t1; }

            a_st[tid] = ty | tlen << 8;
            ZLNG_MK(0);

            // closure of the token chain inside this wavefront: after the loop every lane knows the positions of its own
            // wavefront its chain passes (mask) and where the chain leaves the wavefront (nxg, a global lane index)
            uint32_t cl_nxg = 0;                                    // this lane's closure, kept in registers across the iterations
            u64 cl_mk = 0;
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
                cl_nxg = nxg; cl_mk = mk;
                c_exit[tid] = nxg; c_mask[tid] = mk;
            };
            // The same after the tokens of S in `changed` (a lane mask of this wavefront) took new lengths: a changed lane p only moves
            // the chains that pass THROUGH p -- they keep what they have below p and take p's new continuation -- and the
            // continuation of p is the closure of a lane above it, which is final once the changed lanes above p are done (highest
            // first).  One pass of ~25 instructions per changed lane (one to three per wavefront and iteration) instead of six
            // rounds of three ds_bpermute over all lanes; the result is the full closure's (a -DZLNG_CLOSURE_CHECK build computes both
            // and fails the call on a difference: the encode, fuzz and real-text GPU tests pass under it).  Round 4: parse 581 -> 568 ms.
            auto closure_update = [&](u64 changed) {
                const uint32_t wend = (uint32_t)(64 * (wv + 1) < nlive ? 64 * (wv + 1) : nlive);
                while (changed) {
                    const int p = top_bit(changed);
                    changed &= ~(1ull << p);
                    const uint32_t t = (uint32_t)(64 * wv + p) + ufl((uint32_t)__builtin_amdgcn_readlane((int)tlen, p));
                    u64 nm = 1ull << p;
                    uint32_t nx = t;
                    if (t < wend) {
                        const int lt = (int)(t & 63u);
                        nm |= (u64)(uint32_t)__builtin_amdgcn_readlane((int)(uint32_t)(cl_mk >> 32), lt) << 32 | (uint32_t)__builtin_amdgcn_readlane((int)(uint32_t)cl_mk, lt);
                        nx = (uint32_t)__builtin_amdgcn_readlane((int)cl_nxg, lt);
                    }
                    const bool through = ((cl_mk >> p) & 1ull) != 0;        // (lane p itself included)
                    if (through) { cl_mk = (cl_mk & ((1ull << p) - 1ull)) | nm; cl_nxg = nx; }
                }
                c_exit[tid] = cl_nxg; c_mask[tid] = cl_mk;
            };
            closure();
            ZLNG_MK(1);
            __syncthreads();                                        // (B1) rows claimed, a_st and the closure are in LDS
            ZLNG_MK(2);
            const uint32_t ctx_r = canm ? ctx : 256u, lc1_r = (canm && want1) ? lctx1 : 256u, lc2_r = (canm && want2) ? lctx2 : 256u;
            const uint32_t ek_r = live ? ek : 256u, ekc_r = live ? ctx : 256u;
            ZLNG_MK(3);
            if (prof) t2 = __builtin_readcyclecounter();

            // ---------------- iterate to the fixed point
            u64 KM = 0, CM = 0, EKS = 0, EVu = 0, CONDu = 0, MATu = 0;
            uint32_t rank = 0, ntok = 0, k_ctx = 0;
            bool inS = false, has_ev = false;
            u64 rbit = 0, rbelow = 0;
            int limit = 0;                                          // tokens (ranks) the round commits
            bool limit_hard = false;
            u64 dep_prev = 0;                                       // my bit in the previous iteration's row buffer (cleared one iteration later)
            uint32_t dep_prev_rows = 0;
            int itn = 0;
            for (int it = 0;; it++) {
                itn = it;
                const int bufi = it & 1;
                u64 ta = 0, tb = 0, tc = 0, td = 0;
                if (prof) ta = __builtin_readcyclecounter();
                // chase: one hop per wavefront (every wavefront walks it; the result is uniform).  The dependent chain -- entry lane of
                // the next wavefront the chain reaches -- is NW LDS reads with the address kept in a vector register; the masks of the
                // entries follow in parallel.
                u64 S[NW];
                {
                    uint32_t ev[NW];
                    uint32_t e = 0;
#pragma unroll
                    for (int hop = 0; hop < NW; hop++) { ev[hop] = e; e = c_exit[e < (uint32_t)nlive ? e : 0u] | (e < (uint32_t)nlive ? 0u : 0x8000u); }
                    u64 mv[NW];
#pragma unroll
                    for (int hop = 0; hop < NW; hop++) mv[hop] = c_mask[ev[hop] < (uint32_t)nlive ? ev[hop] : 0u];
#pragma unroll
                    for (int w = 0; w < NW; w++) S[w] = 0ull;
#pragma unroll
                    for (int hop = 0; hop < NW; hop++) {
                        const uint32_t eh = ufl(ev[hop]);
                        const u64 mk = uni64(mv[hop]);
#pragma unroll
                        for (int w = 0; w < NW; w++) if (eh < (uint32_t)nlive && (int)(eh >> 6) == w) S[w] = mk;
                    }
                }
                u64 tx1 = 0, tx2 = 0;
                if (prof) tx1 = __builtin_readcyclecounter();
                // rank of my token in S (tokens beyond the 64th wait for the next round)
                uint32_t pre = 0; ntok = 0;
                u64 Sown = 0, Sprev = 0;
#pragma unroll
                for (int w = 0; w < NW; w++) {
                    if (w < wv) pre += (uint32_t)__popcll(S[w]);
                    if (w == wv) Sown = S[w];
                    if (w + 1 == wv) Sprev = S[w];
                    ntok += (uint32_t)__popcll(S[w]);
                }
                rank = pre + (uint32_t)__popcll(Sown & lbelow);
                inS = ((Sown >> lane) & 1ull) != 0 && rank < 64u;
                if (ntok > 64u) ntok = 64u;
                rbit = inS ? 1ull << (rank & 63u) : 0ull;
                rbelow = rbit - 1ull;                               // (only used when inS)
                // kind of the previous token from my two neighbours (a literal ends one position back, a word two, anything else is a match)
                {
                    const u64 s1 = Sown << 1 | Sprev >> 63, s2 = Sown << 2 | Sprev >> 62;
                    const uint32_t st1 = a_st[tid >= 1 ? tid - 1 : 0], st2 = a_st[tid >= 2 ? tid - 2 : 0];
                    const bool in1 = tid >= 1 && ((s1 >> lane) & 1ull) != 0, in2 = tid >= 2 && ((s2 >> lane) & 1ull) != 0;
                    const uint32_t ty1 = st1 & 0xFF, ty2_ = st2 & 0xFF;
                    uint32_t pty = kTyMatch;
                    if (in2 && (ty2_ == kTyW0 || ty2_ == kTyW1)) pty = ty2_;
                    if (in1 && ty1 == kTyLit) pty = kTyLit;
                    if (tid == 0) pty = prevty;
                    has_ev = inS && (pty == kTyMatch || pty == kTyLit || pty == kTyW1);
                    const bool cond = pty == kTyMatch;
                    if (prof) tx2 = __builtin_readcyclecounter();
                    // the tokens of S put themselves into the rows of this iteration's buffer
                    if (inS) {
                        if (canm) { atomicOr(&keyrow[bufi][r_key], rbit); atomicOr(&ctxrow[bufi][ctx_r], rbit); }   // (the sink rows stay empty)
                        atomicOr(&ekrow[bufi][ek_r], rbit);
                        t_flag[rank] = (has_ev ? 1u : 0u) | (cond ? 2u : 0u);
                        t_lane[rank] = (uint32_t)tid; t_ev[rank] = ew | ek << 16; t_key[rank] = key | chk << 21; t_q[rank] = qtext;
                    }
                }
                if (prof) { tb = __builtin_readcyclecounter(); c_x1 += tx1 - ta; c_x2 += tx2 - tx1; c_x3 += tb - tx2; }
                __syncthreads();                                    // (Bb) rows and token arrays of this iteration are complete
                if (it == 0 && canm) {                              // every row of the round is claimed: the keys can go
                    ht_key[r_key] = kWgEmpty;
                    if (want1) ht_key[r_lk1] = kWgEmpty;
                    if (want2) ht_key[r_lk2] = kWgEmpty;
                }
                // my bits of the PREVIOUS iteration's buffer go now (nobody reads it any more, nobody writes it before the next barrier)
                if (dep_prev) {
                    atomicAnd(&keyrow[bufi ^ 1][dep_prev_rows & 0xFFFu], ~dep_prev);
                    atomicAnd(&ctxrow[bufi ^ 1][(dep_prev_rows >> 12) & 0x1FFu], ~dep_prev);
                    atomicAnd(&ekrow[bufi ^ 1][(dep_prev_rows >> 21) & 0x1FFu], ~dep_prev);
                }
                dep_prev = rbit; dep_prev_rows = r_key | ctx_r << 12 | ek_r << 21;
                {   // uniform masks over the tokens: every wavefront reads the 64 flag words and votes (no same-address atomics)
                    const uint32_t f = (uint32_t)lane < ntok ? t_flag[lane] : 0u;
                    EVu = __ballot((f & 1u) != 0); CONDu = __ballot((f & 3u) == 3u);
                }
                if (prof) { tc = __builtin_readcyclecounter(); mk_last = tc; }

                // ---- E: my token given the tokens of S before me.  Straight-line: every lane computes, the tokens of S keep the result
                // (divergent branches cost a lone wavefront more than the selects do); only rare work sits behind uniform branches.
                KM = keyrow[bufi][r_key]; CM = ctxrow[bufi][ctx_r]; EKS = ekrow[bufi][ek_r];
                const u64 LK1 = keyrow[bufi][r_lk1], LC1 = ctxrow[bufi][lc1_r], EKC = ekrow[bufi][ekc_r];
                const u64 rbeq = rbelow | rbit;
                const uint32_t k = (uint32_t)__popcll(CM & rbelow);
                k_ctx = k;
                bool hard, is_match;
                uint32_t ml = kMatchMin - 1, mn = 0;
                int lk = -1;
                uint32_t lkchk = 0, lklane = 0;
                const bool ecan = inS && canm;
                if (level0) {
                    const u64 kq = KM & rbelow;
                    const bool has_a1 = kq != 0ull;
                    const int a1 = top_bit(kq | 1ull);
                    const u64 kq2 = kq & ~(1ull << a1);
                    const bool has_a2 = has_a1 && kq2 != 0ull;
                    const int a2 = has_a2 ? top_bit(kq2 | 1ull) : a1;
                    const bool ring0 = W.has0 && W.d0 <= k, ring1 = W.has1 && W.d1 <= k;
                    hard = ecan && (has_a1 ? (!has_a2 && ring0) : ring0);
                    // no token of this round in my hash slot: the speculation stands -- unless node 1's slot was rewritten by one (ring1): it
                    // holds a later position than node 0's now, and the reference's chain-end test (src/libzling_lz.cpp:265) stops the walk there
                    ml = ring1 ? (W.len0 > 3u ? W.len0 : 3u) : W.len;
                    mn = ring1 ? W.node0 : W.node;
                    const bool fixl = ecan && !hard && has_a1;
                    if (__any(fixl)) {
                        // the chain is [a1, a2 | the snapshot's head] (depth 2): candidate lengths from the window's own text
                        const uint32_t k1 = t_key[a1], k2 = t_key[a2], la1 = t_lane[a1], la2 = t_lane[a2];
                        const Quad q1 = t_q[a1], q2 = t_q[a2];
                        const bool c1 = fixl && (k1 >> 21) == chk, c2 = fixl && has_a2 && (k2 >> 21) == chk;
                        uint32_t l1 = c1 ? lcp16(qtext, q1) : 0u, l2 = c2 ? lcp16(qtext, q2) : 0u;
                        const bool g1 = c1 && l1 == 16u, g2 = c2 && l2 == 16u;
                        if (__any(g1 || g2)) {
                            uint32_t x1, x2;
                            lcp_tail2(buf + upos, buf + (uint32_t)(P + (int)la1), buf + (uint32_t)(P + (int)la2), g1, g2, x1, x2);
                            l1 = g1 ? x1 : l1; l2 = g2 ? x2 : l2;
                        }
                        const bool second = has_a2 || W.has0;
                        const uint32_t ls = has_a2 ? l2 : W.len0, ns = has_a2 ? (0x10000u | (uint32_t)a2) : W.node0;
                        uint32_t fl = kMatchMin - 1, fn = 0;
                        if (l1 > fl) { fl = l1; fn = 0x10000u | (uint32_t)a1; }
                        if (fl != (uint32_t)kMatchMax && second && ls > fl) { fl = ls; fn = ns; }
                        if (fixl) { ml = fl; mn = fn; lk = a1; lkchk = k1 >> 21; lklane = la1; }
                    }
                    ZLNG_MK(4);
                    is_match = ecan && !hard && ml >= (uint32_t)kMatchMin;
                    // the lazy probe under ml (src/libzling_lz.cpp:270-281, 291-316; depth 1: only the chain head is looked at)
                    const bool lzq = is_match && ml < (uint32_t)kLazyLimit;
                    const u64 lq = LK1 & rbeq;                              // tokens up to and including me with the probe's key
                    const bool has_lh = lq != 0ull;
                    const int lh = top_bit(lq | 1ull);
                    const bool lconf = (uint32_t)__popcll(LC1 & rbeq) > W.ld1;   // a visited slot at distance d is rewritten by the (d+1)-th insert
                    hard = hard || (lzq && lconf);
                    const bool need_re = lzq && !lconf && (has_lh || ml != W.len);
                    bool veto = W.veto1;
                    if (__any(need_re)) {
                        // position bytes mm+1 .. mm+4 against source bytes mm .. mm+3 of the chain head the probe sees now
                        const uint32_t mm = need_re ? ml - 3u : 0u, mc = mm <= 12u ? mm : 0u;
                        const Quad tq = t_q[lh];
                        const Quad sq = has_lh ? tq : ql;
                        uint32_t pr = byte_window(qtext, t16, mc + 1u), sr = byte_window(sq, 0u, mc);
                        const bool far = need_re && mm > 12u;
                        if (__any(far)) {
                            const uint32_t so = has_lh ? (uint32_t)(P + (int)t_lane[lh]) : (W.lsrc1 & 0xFFFFFF);
                            const uint32_t pf = ld32u(buf + (upos + 1u + (far ? mm : 0u))), sf = ld32u(buf + (far ? so + mm : upos));
                            if (far) { pr = pf; sr = sf; }
                        }
                        if (need_re) veto = (has_lh || (W.lsrc1 >> 31) != 0) && pr == sr;
                    }
                    is_match = is_match && !(lzq && !lconf && veto);
                } else {
                    // levels 1-4 (depth 4..16, probes of depth up to 4 and 2).  One or two tokens of this round in my hash slot head my
                    // chain; behind them come the first depth - j nodes of the chain the speculation walked, whose best it recorded
                    // (pre1 / pre2).  Three or more, a rewritten ring slot, a probe whose length changed -> hard.
                    const u64 kq = KM & rbelow;
                    const bool has_a1 = kq != 0ull;
                    const int a1 = top_bit(kq | 1ull);
                    const u64 kq2 = kq & ~(1ull << a1);
                    const bool has_a2 = has_a1 && kq2 != 0ull;
                    const int a2 = has_a2 ? top_bit(kq2 | 1ull) : a1;
                    const bool has_a3 = has_a2 && (kq2 & ~(1ull << a2)) != 0ull;
                    // Ring rule (a.ring_fix; scripts/experiments/wg_parser_model.c ring_fix, exact there at every level): a chain node whose
                    // slot a token of this round has taken over ENDS the walk in front of it -- the reference reads a later position
                    // there and its chain-end test stops (src/libzling_lz.cpp:265) -- so the match is the best over the nodes before
                    // it, which phase 1 recorded for the walk's tail.  Node 0 taken over, a token of the round in my hash slot as well,
                    // or a tail longer than the record: still hard.  (A slot that was only read for the chain-end test changes nothing.)
                    if constexpr (kRingRule) {
                        const bool ringhit = W.dmin <= k;
                        bool cut = false;
                        uint32_t cutpre = W.len | W.node << kSpNodeShift;
                        if (ecan && ringhit && !has_a1 && W.d0g > k) {
                            const bool v0 = W.ntail > 0u && (W.tail0 & 63u) <= k, v1 = W.ntail > 1u && (W.tail1 & 63u) <= k, v2 = W.ntail > 2u && (W.tail2 & 63u) <= k;
                            if (v0) { cut = true; cutpre = W.tail0 >> 6; }
                            else if (v1) { cut = true; cutpre = W.tail1 >> 6; }
                            else if (v2) { cut = true; cutpre = W.tail2 >> 6; }
                            else cut = W.ntail <= 3u;
                        }
                        hard = ecan && ((ringhit && !cut) || has_a3);
                        ml = cutpre & kSpLenMask; mn = (cutpre >> kSpNodeShift) & (kRing - 1);
                    } else {
                        hard = ecan && (W.dmin <= k || has_a3);
                        ml = W.len; mn = W.node;
                    }
                    const bool fixl = ecan && !hard && has_a1;
                    if (__any(fixl)) {
                        const uint32_t k1 = t_key[a1], k2 = t_key[a2], la1 = t_lane[a1], la2 = t_lane[a2];
                        const Quad q1 = t_q[a1], q2 = t_q[a2];
                        const bool c1 = fixl && (k1 >> 21) == chk, c2 = fixl && has_a2 && (k2 >> 21) == chk;
                        uint32_t l1 = c1 ? lcp16(qtext, q1) : 0u, l2 = c2 ? lcp16(qtext, q2) : 0u;
                        const bool g1 = c1 && l1 == 16u, g2 = c2 && l2 == 16u;
                        if (__any(g1 || g2)) {
                            uint32_t x1, x2;
                            lcp_tail2(buf + upos, buf + (uint32_t)(P + (int)la1), buf + (uint32_t)(P + (int)la2), g1, g2, x1, x2);
                            l1 = g1 ? x1 : l1; l2 = g2 ? x2 : l2;
                        }
                        uint32_t fl = kMatchMin - 1, fn = 0;
                        if (l1 > fl) { fl = l1; fn = 0x10000u | (uint32_t)a1; }
                        if (has_a2 && fl != (uint32_t)kMatchMax && l2 > fl) { fl = l2; fn = 0x10000u | (uint32_t)a2; }
                        const uint32_t pre = has_a2 ? W.pre2 : W.pre1;
                        if (fl != (uint32_t)kMatchMax && (pre & kSpLenMask) > fl) { fl = pre & kSpLenMask; fn = (pre >> kSpNodeShift) & (kRing - 1); }
                        if (fixl) { ml = fl; mn = fn; lk = a1; lkchk = k1 >> 21; lklane = la1; }
                    }
                    is_match = ecan && !hard && ml >= (uint32_t)kMatchMin;
                    const bool lzq = is_match && ml < (uint32_t)kLazyLimit;
                    // a probe: 0 no veto, 1 veto, 2 hard.  [the newest one or two tokens up to me with the probe's key] ++ the first
                    // depth - h nodes of the chain the speculation walked (vpos), valid while the length is the speculation's.
                    auto probe = [&](const u64 LK, const u64 LC, uint32_t ld, bool sveto, uint32_t vpos, int Lp, uint32_t pofs) -> uint32_t {
                        const u64 hm = LK & rbeq;
                        const bool lconf = (uint32_t)__popcll(LC & rbeq) > ld;
                        const bool has_h1 = hm != 0ull;
                        const int h1 = top_bit(hm | 1ull);
                        const u64 hm2 = hm & ~(1ull << h1);
                        const bool has_h2 = has_h1 && hm2 != 0ull;
                        const int h2 = has_h2 ? top_bit(hm2 | 1ull) : h1;
                        const bool has_h3 = has_h2 && (hm2 & ~(1ull << h2)) != 0ull;
                        uint32_t st = sveto ? 1u : 0u;
                        const bool ex = lzq && has_h1 && !lconf && !has_h3 && ml == W.len;
                        if (lzq && (lconf || (has_h1 && !ex) || ml != W.len)) st = 2u;      // (a changed length would need the probe's chain walked again)
                        if (__any(ex)) {
                            const uint32_t mm = ex ? ml - 3u : 0u;
                            const uint32_t pr = ld32u(buf + (upos + pofs + mm));
                            const uint32_t s1w = ld32u(buf + ((uint32_t)(P + (int)t_lane[h1]) + mm)), s2w = ld32u(buf + ((uint32_t)(P + (int)t_lane[h2]) + mm));
                            const uint32_t hcnt = has_h2 ? 2u : 1u;
                            bool v = pr == s1w || (has_h2 && Lp >= 2 && pr == s2w);
                            if (!v && hcnt < (uint32_t)Lp) v = vpos < (uint32_t)Lp - hcnt;
                            if (ex) st = v ? 1u : 0u;
                        }
                        return st;
                    };
                    const u64 LC1b = LC1;
                    const uint32_t st1 = want1 ? probe(LK1, LC1b, W.ld1, W.veto1, W.vpos1, cfg.lazy1, 1u) : 0u;
                    uint32_t st2 = 0u;
                    if (want2) {
                        const u64 LK2 = keyrow[bufi][r_lk2], LC2 = ctxrow[bufi][lc2_r];
                        st2 = probe(LK2, LC2, W.ld2, W.veto2, W.vpos2, cfg.lazy2, 2u);
                    }
                    // probe 2 is only looked at when probe 1 does not veto (src/libzling_lz.cpp:276-281)
                    const uint32_t stl = st1 != 0u ? st1 : st2;
                    hard = hard || (lzq && stl == 2u);
                    is_match = is_match && !(lzq && stl == 1u);
                }
                // word MRU of my context after every boundary event of S up to and including mine (src/libzling_lz.cpp:172-185)
                uint32_t s0 = m0c & 0xFFFF, s1 = m0c >> 16;
                {
                    const bool nm = inS && !hard && !is_match;
                    const u64 M = nm ? (EKC & EVu & rbeq) : 0ull;
                    bool act = M != 0ull;
                    int e = top_bit(M | 1ull);
                    uint32_t w = t_ev[e] & 0xFFFF;
                    if (act) s0 = w;
                    u64 cur = M & ~(1ull << e);
                    for (int i = 0; i < 4; i++) {
                        if (!__any(act)) break;
                        const bool more = cur != 0ull;
                        const int e2 = top_bit(cur | 1ull);
                        const uint32_t tw = t_ev[e2] & 0xFFFF;
                        const uint32_t pw = more ? tw : (m0c & 0xFFFF);
                        const bool eff = !((CONDu >> e) & 1ull) || w != pw;
                        if (act && eff) s1 = pw;
                        act = act && !eff && more;
                        e = e2; cur &= ~(1ull << e2); w = pw;
                    }
                    hard = hard || act;                                     // not settled within four events of the key
                }
                ZLNG_MK(6);
                uint32_t ty2, tlen2;
                {
                    const bool two = pos + 1 < ilen;
                    const uint32_t wty = (two && s0 == cw) ? kTyW0 : ((two && s1 == cw) ? kTyW1 : kTyLit);
                    ty2 = is_match ? kTyMatch : wty;
                    tlen2 = is_match ? ml : (wty == kTyLit ? 1u : 2u);
                    const bool keep = !inS || hard;
                    ty2 = keep ? ty : ty2; tlen2 = keep ? tlen : tlen2;
                    if (!keep) { mlen = ml; mnode = mn; link = lk; link_chk = lkchk; link_lane = lklane; }
                }
                hard = hard && inS;
                const bool chg = ty2 != ty || tlen2 != tlen;
                if (inS) t_res[rank] = (hard ? 4u : 0u) | (chg ? 8u : 0u) | (ty2 == kTyMatch ? 16u : 0u);
                ty = ty2; tlen = tlen2;
                if (chg) a_st[tid] = ty | tlen << 8;
                ZLNG_MK(7);
                {
                    const u64 chm = __ballot(chg);
                    if (chm) {
#ifdef ZLNG_CLOSURE_CHECK
                        closure_update(chm);
                        const uint32_t u_n = cl_nxg; const u64 u_m = cl_mk;
                        closure();
                        if (live && (u_n != cl_nxg || u_m != cl_mk)) atomicMax(a.overflow, 2u);      // (no change of control flow: the flag fails the call)
#else
                        if (a.min_restart == -2) closure(); else closure_update(chm);      // ZLNG_MIN_RESTART=-2: the full closure, for A/B timing
#endif
                    }
                }
                ZLNG_MK(8);                          // a wavefront whose lengths did not change keeps its closure
                if (prof) td = __builtin_readcyclecounter();
                __syncthreads();                                    // (Be)
                ZLNG_MK(9);
                u64 HB, CB;
                {
                    const uint32_t f = (uint32_t)lane < ntok ? t_res[lane] : 0u;
                    HB = __ballot((f & 4u) != 0); CB = __ballot((f & 8u) != 0); MATu = __ballot((f & 16u) != 0);
                }
                // cut of the round: the first hard token, the 64th token ...
                limit = (int)ntok; limit_hard = false;
                if (HB) { limit = (int)__builtin_ctzll(HB); limit_hard = true; }
                // ... or the end of the sub-block (src/libzling_lz.cpp:153): token r may start while opos + 1 < kSubSyms
                {
                    const u64 lm = limit >= 64 ? ~0ull : ((1ull << limit) - 1ull);
                    const uint32_t tot = (uint32_t)limit + (uint32_t)__popcll(MATu & lm);
                    if ((uint32_t)opos + tot + 1u >= (uint32_t)kSubSyms) {
                        if (tid == 0) u_cut = 0;
                        __syncthreads();
                        if (inS && !((uint32_t)opos + rank + (uint32_t)__popcll(MATu & rbelow) + 1u < (uint32_t)kSubSyms)) atomicOr(&u_cut, rbit);
                        __syncthreads();
                        const u64 v = uni64(u_cut);
                        if (v) { const int c = (int)__builtin_ctzll(v); if (c <= limit) { limit = c; limit_hard = false; } }
                        if (prof) n_cutr++;
                    }
                }
                const u64 lm = limit >= 64 ? ~0ull : ((1ull << limit) - 1ull);
                const bool changed = (CB & lm) != 0ull;
                if (prof) { n_iter++; c_dep += tb - ta; c_ev += td - tc; c_lim += __builtin_readcyclecounter() - td; }
                if (!changed) break;
                // The tokens in front of the first changed one are final already (each was evaluated under exactly the tokens before it,
                // none of which changed).  When that prefix is most of the round, committing it and starting the next round AT the
                // changed token is cheaper than a second iteration for the few tokens behind it: with the threshold at 70 % the model
                // (scripts/experiments/wg_parser_model.c, prefix_pct) says 1.68 -> 1.52 iterations per round for 3 % more rounds.  MEASURED
                // (ZLNG_PREFIX_PCT, same box): benchmark text 574-578 -> 570-571 ms, real text 671 -> 677, e4 1,444 -> 1,463: a wash; off by default.
                if (it == 0 && a.prefix_pct > 0) {
                    const int cf = (int)__builtin_ctzll(CB & lm);
                    if (cf > 0 && cf * 100 >= a.prefix_pct * limit) { limit = cf; limit_hard = false; break; }
                }
                if (it > 2 * NL) { overflow = true; internal = true; break; }   // cannot happen: every iteration fixes at least one token of S
            }
            if (prof) t3 = __builtin_readcyclecounter();

            // ---------------- commit the first `limit` tokens of S
            const u64 Cm = limit >= 64 ? ~0ull : ((1ull << limit) - 1ull);
            const bool mine = inS && (int)rank < limit;
            if (mine) {
                uint32_t word;
                if (canm) {
                    // dictionary insert (src/libzling_lz.cpp:227-230); slots of a bucket are handed out in position order
                    const uint32_t head = (head0 + k_ctx + 1u) & (kRing - 1);
                    BucketT<kWide> B(dict, ctx);
                    uint32_t lslot = W.node0, pword = W.ov0;
                    if (link >= 0) { lslot = (head0 + (uint32_t)__popcll(CM & ((1ull << link) - 1ull)) + 1u) & (kRing - 1); pword = (uint32_t)(P + (int)link_lane) | link_chk << 24; }
                    B.suffix[head] = (uint16_t)lslot;
                    if (kWide) B.slot[head] = (u64)((uint32_t)pos | chk << 24) | (u64)pword << 32;
                    else B.offset[head] = (uint32_t)pos | chk << 24;
                    // the slot's head must end up being the LAST token of the round in it; the bucket's ring head likewise
                    const bool hwr = (KM & Cm & ~(rbelow | rbit)) == 0ull;
                    if (hwr) B.hash[hc] = (uint16_t)head;
                    if (kHot && ctx == hotctx) {
                        reinterpret_cast<uint16_t*>(hot + 8u * kRing)[head] = (uint16_t)lslot;
                        hot_lds[head] = (u64)((uint32_t)pos | chk << 24) | (u64)pword << 32;
                        if (hwr) reinterpret_cast<uint16_t*>(hot + 8u * kRing + 2u * kRing)[hc] = (uint16_t)head;
                    }
                    if ((CM & Cm & ~(rbelow | rbit)) == 0ull) heads[ctx] = (uint16_t)head;
                    uint32_t msl = mnode;
                    if (mnode & 0x10000u) msl = (head0 + (uint32_t)__popcll(CM & ((1ull << (mnode & 63u)) - 1ull)) + 1u) & (kRing - 1);
                    word = (258u + mlen - kMatchMin) | ((head - msl) & (kRing - 1)) << 16;
                } else word = 0;
                if (ty == kTyW0) word = 256; else if (ty == kTyW1) word = 257; else if (ty == kTyLit) word = b_0 | ctx << 16;
                __builtin_nontemporal_store(word, &tok[nt + rank]);
                // MRU slots of my event key after the round: written by the key's last committed event
                if (has_ev && (EKS & EVu & Cm & ~(rbelow | rbit)) == 0ull) {
                    uint32_t s0, s1;
                    ev_state(EKS & EVu & (rbelow | rbit), CONDu, m0e, t_ev, s0, s1, 64);
                    mru[ek] = s0 | s1 << 16;
                }
            }
            // round summary (uniform): the last committed token leads on
            if (limit > 0) {
                const uint32_t ll = ufl(t_lane[limit - 1]);
                const uint32_t st = ufl(a_st[ll]);
                q = P + (int)ll + (int)(st >> 8);
                prevty = st & 0xFF;
                nt += (uint32_t)limit; opos += limit + __popcll(MATu & Cm);
            }
            // A hard token behind committed ones: at level 0 it is replayed by the serial code (3 K cycles); at levels 1-4 that replay
            // is a walk of up to 16 dependent round trips by one wavefront (9-16 K cycles), and the next round, which starts AT the
            // token, evaluates it as its lane 0 against the dictionary as it then is -- exact unless it conflicts with its own insert.
            serial_next = limit_hard && (level0 || limit == 0 || a.min_restart < 0);
            if (!serial_next && q < ilen) {                          // the next round starts at q: its text
                const int np = q + tid;
                const uint32_t nu = np < ilen ? (uint32_t)np : (uint32_t)q;
                pf_wraw = ld32u(buf + (nu >= 4u ? nu - 4u : 0u));
                pf_q = ld128u(buf + nu);
                pf_t16 = ld32u(buf + (nu + 16u));
                pf_P = q;
            } else pf_P = -1;
            if (limit == 0 && !limit_hard) { overflow = true; internal = true; }   // cannot happen: a round commits a token or names a hard one
            // my bits of the last iteration's buffer
            if (dep_prev) {
                const int bl = itn & 1;
                atomicAnd(&keyrow[bl][dep_prev_rows & 0xFFFu], ~dep_prev);
                atomicAnd(&ctxrow[bl][(dep_prev_rows >> 12) & 0x1FFu], ~dep_prev);
                atomicAnd(&ekrow[bl][(dep_prev_rows >> 21) & 0x1FFu], ~dep_prev);
            }
            __syncthreads();                                        // (B6) inserts, heads, MRU are visible; rows and arrays are free again
            if (prof) {
                const u64 t4 = __builtin_readcyclecounter();
                c_p1 += t1 - t0; c_tab += t2 - t1; c_it += t3 - t2; c_com += t4 - t3; n_round++; n_pos += (u64)(q - P);
                if (limit_hard) n_hardr++;
            }
        }
        if (nsub < kMaxSub && tid == 0) cuts[nsub] = SubCut{tok_begin, nt, (uint32_t)q, (uint32_t)opos};
        nsub++;
    }
    if (tid == 0) {
        if (overflow) { if (internal) atomicMax(a.overflow, 2u); else atomicMax(a.overflow, 1u); nsub = 0; nt = 0; }
        a.nsub[blk] = (uint32_t)nsub; a.ntok[blk] = nt;
    }
    if (prof && tid == 0) {
        u64* d = a.dbg + (size_t)blk * kDbgSlots;
        d[0] = c_p1; d[1] = c_tab; d[2] = c_it; d[3] = n_round; d[4] = nt; d[5] = n_iter; d[6] = n_ser; d[7] = n_pos; d[8] = c_ser; d[9] = c_com;
        d[10] = n_hardr; d[11] = n_cutr; d[12] = c_dep; d[13] = c_ev; d[14] = c_lim; d[15] = c_x1; d[16] = c_x2; d[17] = c_x3;
        for (int i = 0; i < 6; i++) d[18 + i] = c_mk[i] << 32 >> 32 | c_mk[6 + i] << 32;
    }
}

void launch_rolz_parse_wg(const ParseArgs& a, uint32_t nblocks_all, hipStream_t s, bool all_level0, int nw, bool wide, bool hot) {
    const bool prof = a.dbg != nullptr;
    const uint32_t nblocks = nblocks_all - a.blk0;
    if (hot && all_level0 && wide && nw > 2 && nw <= 4) {        // the measured LDS-bucket variant (ZLNG_WG_HOT=1): NW = 4, wide plane, level 0
        if (prof) hipLaunchKernelGGL((k_rolz_parse_wg<4, true, true, true, true>), dim3(nblocks), dim3(256), 0, s, a);
        else hipLaunchKernelGGL((k_rolz_parse_wg<4, true, false, true, true>), dim3(nblocks), dim3(256), 0, s, a);
        return;
    }
    // kWide: slot plane form (zlng_common.h; the caller's k_dict_reset matches).  Only a level-0 context can use the wide form.
#define ZLNG_WG_LAUNCH(NW)                                                                                                          \
    do {                                                                                                                            \
        if (all_level0 && wide && !prof) hipLaunchKernelGGL((k_rolz_parse_wg<NW, true, false, true>), dim3(nblocks), dim3(64 * NW), 0, s, a);   \
        else if (all_level0 && wide) hipLaunchKernelGGL((k_rolz_parse_wg<NW, true, true, true>), dim3(nblocks), dim3(64 * NW), 0, s, a);        \
        else if (all_level0 && !prof) hipLaunchKernelGGL((k_rolz_parse_wg<NW, true, false, false>), dim3(nblocks), dim3(64 * NW), 0, s, a);     \
        else if (all_level0) hipLaunchKernelGGL((k_rolz_parse_wg<NW, true, true, false>), dim3(nblocks), dim3(64 * NW), 0, s, a);               \
        else if (a.ring_fix != 0 && !prof) hipLaunchKernelGGL((k_rolz_parse_wg<NW, false, false, false, false, true>), dim3(nblocks), dim3(64 * NW), 0, s, a);  \
        else if (a.ring_fix != 0) hipLaunchKernelGGL((k_rolz_parse_wg<NW, false, true, false, false, true>), dim3(nblocks), dim3(64 * NW), 0, s, a);  \
        else if (!prof) hipLaunchKernelGGL((k_rolz_parse_wg<NW, false, false, false>), dim3(nblocks), dim3(64 * NW), 0, s, a);                  \
        else hipLaunchKernelGGL((k_rolz_parse_wg<NW, false, true, false>), dim3(nblocks), dim3(64 * NW), 0, s, a);                              \
    } while (0)
    if (nw <= 2) ZLNG_WG_LAUNCH(2);
    else if (nw <= 4) ZLNG_WG_LAUNCH(4);
    else ZLNG_WG_LAUNCH(8);
#undef ZLNG_WG_LAUNCH
}

}  // namespace zlng
