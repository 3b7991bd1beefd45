// Retired in round 4 (not built): the one-wavefront speculative parser of rounds 1-2, cut out of csrc/rolz_parse.hip.
// It needs rolz_dev.h of commit 15b8015 (speculate_l0w / finish_open lived there).
#include "zlng_common.h"
#include "zlng_kernels.h"
#include "rolz_dev.h"
namespace zlng {
// ------------------------------------------------------------------------------ K1 (wavefront form)
// The production parser.  One wavefront owns a block and advances in rounds over a window of 64
// consecutive input positions starting at the next token start P:
//
//  phase 1  (64 lanes, read-only) lane l evaluates position P+l AS IF it were a token start
//           against the dictionary state at the start of the round: hash head, the <= depth chain
//           nodes, the longest-match search and the lazy probes at P+l+1 / P+l+2.  All the
//           dependent HBM round trips of a token are paid once per window instead of once per token,
//           and independent loads (ring offset + suffix of a node, the lazy probes' hash heads,
//           16-byte compare chunks) are issued together to shorten the chain.
//  phase 2  resolves the true token chain inside the window.  A lone wavefront issues about one
//           instruction per 4 cycles, so nothing per-token is done in scalar code except a
//           pointer chase over the per-lane token lengths (P -> P+len -> ...); everything else is
//           evaluated for all 64 lanes at once:
//             * a lane's speculative match is valid unless an EARLIER ACCEPTED start of this round
//               wrote something it read: its hash slot (same (ctx, hash13): `keymask & acc`) or a
//               ring slot it visited (slots are handed out consecutively per context, so the slots
//               written this round are head0+1 .. head0+k: `dmin <= k`); the lazy probes have their
//               own read sets, checked against accepted starts INCLUDING the lane's own insert
//               (a probe may hit the entry just inserted, src/libzling_lz.cpp:271);
//             * a non-match lane is a literal unless the word MRU can possibly hit: the two MRU
//               slots of its context only ever hold their value at the start of the segment or
//               the word of an in-window token boundary with the same key, so lanes whose word
//               matches neither are literals without looking at MRU state;
//           the clean prefix of the chase is then COMMITTED by vector code -- dictionary inserts,
//           token words (coalesced), and the word-MRU events of all its token boundaries (an exact
//           lane-parallel evaluation of the 2-slot push rules) -- and only the first "problem"
//           token (conflict, possible word hit) is replayed by exact scalar code (match_exact and
//           the serial MRU logic of EncodeImpl), after which the chase resumes behind it.
//           The result is the reference's in every case.
//
// MRU bookkeeping convention: the push that EncodeImpl performs after a token (src/libzling_lz.cpp
// :163-166, :181-182, :190-191) is attached to the NEXT token start ("boundary event" with key
// buf[e-3] and word buf[e-2..e-1], conditional after a match, unconditional after a literal or a
// 257 word, absent after a 256 word) and applied when that lane is processed; the type of the
// last token is carried across rounds, and dropped at a sub-block end like the reference's MRU.
// Long matches at level 0: phase 1 compares 16 bytes per chain node and leaves a lane whose compare ran that far "open"
// (rolz_dev.h Spec); open lanes absorb the chase's jump chains, and the one the chase actually reaches as a token start is
// settled there (finish_open: all lanes compare 4 bytes each, one round trip) before it is validated.  Inside a long match
// every lane of the window is long; almost none of them is a token start.
// kAllL0: every sub-block of the batch runs at level 0 (always true for an e0 context, whose schedule cannot
// change): the generic speculation and the level tests drop out of the kernel, and the dictionary's slot plane takes its
// wide form (zlng_common.h; the launcher's k_dict_reset must agree).
// kProf: cycle counters into a.dbg (ZLNG_PROFILE=1); compiled out of the production kernels.
template <bool kAllL0, bool kProf>
__global__ __launch_bounds__(512) void k_rolz_parse_wave(ParseArgs a) {
    __shared__ uint16_t heads[256];
    __shared__ uint32_t mru[256];                    // slot0 | slot1 << 16
    // lane-mask tables; the extra last entry of each is a sink: lanes past the end of the block (not live / no room
    // for a match) deposit and clear there, so none of these accesses sits under an exec-mask branch
    __shared__ unsigned long long keytab[kKeyTab + 1];
    __shared__ unsigned long long ctxtab[256 + 1];
    __shared__ unsigned long long evtab[kEvTab + 1];
    __shared__ unsigned long long ektab[256 + 1];
    __shared__ unsigned long long pred_mask;         // lanes that are the in-slot predecessor of a later lane of the same commit set
    __shared__ int pf_pos, pf_level, pf_done;        // round start / level / end flag published for the prefetch wave
    const uint32_t blk = blockIdx.x + a.blk0;
    const size_t base = (size_t)blk * kBlockIn;
    if (base >= a.in_len) return;
    const uint8_t* buf = a.in + base;
    const int ilen = (int)((a.in_len - base) < (size_t)kBlockIn ? (a.in_len - base) : (size_t)kBlockIn);
    uint8_t* dict = a.dict + (size_t)blk * kDictBytes;
    uint32_t* tok = a.tok + (size_t)blk * a.tok_cap;
    SubCut* cuts = a.cuts + (size_t)blk * kMaxSub;
    const int lane = threadIdx.x & 63;
    const int wave = threadIdx.x >> 6;
    const unsigned long long lane_bit = 1ull << lane;
    const unsigned long long below = lane_bit - 1ull, beloweq = below | lane_bit;
    constexpr bool kWide = kAllL0;                   // slot plane form (zlng_common.h): the launcher's reset matches

    if (wave == 0) {
        for (int i = lane; i < 256 + 1; i += 64) { if (i < 256) heads[i] = 0; ctxtab[i] = 0; ektab[i] = 0; }
        for (int i = lane; i < kKeyTab + 1; i += 64) { keytab[i] = 0; evtab[i] = 0; }
        if (lane == 0) { pf_pos = 0; pf_level = a.lvl_sched[blk * kMaxSub]; pf_done = 0; }
    }
    __syncthreads();                                 // the only workgroup barrier: wave 1 never joins another one

    if (wave >= 1) {
        // wave 1 runs furthest ahead (span pf_ahead), every further wave re-touches a nearer 64-position window
        const int nw = a.pf_waves;
        const int near_lead = 64 * (nw - wave), span = wave == 1 ? a.pf_ahead : 64;
        // ---- prefetch wavefront: runs the same speculative loads for the next window(s), results discarded.
        // It only warms L2/L1 for the dependent chain (hash head -> ring entry -> source bytes) that bounds
        // phase 1; correctness never depends on it (no LDS/global writes, benign race on pf_pos).
        int done_to = 0;
        while (true) {
            const int P = __atomic_load_n(&pf_pos, __ATOMIC_RELAXED);
            if (__atomic_load_n(&pf_done, __ATOMIC_RELAXED)) break;
            int start = P + 64 + near_lead > done_to ? P + 64 + near_lead : done_to;
            if (start >= P + 64 + near_lead + span || start >= ilen) { __builtin_amdgcn_s_sleep(16); continue; }
            const LevelCfg pcfg = kAllL0 ? level_cfg(0) : level_cfg(__atomic_load_n(&pf_level, __ATOMIC_RELAXED));
            const int pos = start + lane;
            if (pos >= 4 && pos + kSentinel < ilen) {
                const uint32_t wpp = ld32u(buf + pos - 4);
                const Quad qap = ld128u(buf + (uint32_t)pos);
                const uint32_t hp = hash_of(qap.a);
                Spec S;
                const uint32_t pctx = wpp >> 24, pl1 = qap.a & 0xFF, pl2 = (qap.a >> 8) & 0xFF;
                if (kAllL0 || (pcfg.depth == 2 && pcfg.lazy1 == 1 && pcfg.lazy2 == 0)) {
                    speculate_l0w<kWide>(S, dict, buf, heads[pctx], heads[pl1], kRiskDist, pos, qap, 0u, pctx, hp % kHashSlots, (hp / kHashSlots) & 255u);
                } else speculate(S, dict, buf, heads[pctx], heads[pl1], heads[pl2], kRiskDist, pos, pcfg, qap, pctx, hp % kHashSlots, (hp / kHashSlots) & 255u);
                asm volatile("" :: "v"(S.sp), "v"(S.dmin), "v"(S.node0));
            }
            done_to = start + 64;
        }
        return;
    }

    uint32_t nt = 0;
    int q = 0, nsub = 0;
    bool overflow = false;
    bool settled_prev = false;                       // the previous round settled an open lane (finish_open)
    unsigned long long c_p1 = 0, c_mask = 0, c_p2 = 0, n_round = 0, n_redo = 0, n_poss = 0, n_seg = 0, c_ser = 0, c_chase = 0;
    unsigned long long n_cA = 0, n_cB = 0, n_cL = 0, n_same = 0, n_replay = 0, c_val = 0, c_com = 0, n_fin = 0, c_fin = 0, c_finw = 0, n_lfix = 0;
    const bool prof = kProf && a.dbg != nullptr;

    while (q < ilen && !overflow) {                  // ---- one sub-block (one EncodeImpl call)
        const LevelCfg cfg = kAllL0 ? level_cfg(0) : level_cfg(a.lvl_sched[blk * kMaxSub + (nsub < kMaxSub ? nsub : kMaxSub - 1)]);
        if (lane == 0) __atomic_store_n(&pf_level, (int)a.lvl_sched[blk * kMaxSub + (nsub < kMaxSub ? nsub : kMaxSub - 1)], __ATOMIC_RELAXED);
        const bool level0 = kAllL0 || (cfg.depth == 2 && cfg.lazy1 == 1 && cfg.lazy2 == 0);
        const uint32_t tok_begin = nt;
        int opos = 0;
        uint32_t prevty = kTyNone;                   // kind of the token that ended at q (none: MRU starts empty)
        for (int i = lane; i < 256; i += 64) mru[i] = 0;
        wsync();
        if (q == 0) {                                // src/libzling_lz.cpp:150-151
            if (lane == 0) tok[nt] = (uint32_t)buf[0] | kTokRawCtx << 16;
            nt++; q = 1; opos = 1;
            if (ilen > 1) { if (lane == 0) tok[nt] = (uint32_t)buf[1] | kTokRawCtx << 16; nt++; q = 2; opos = 2; }
        }

        while (q < ilen && opos + 1 < kSubSyms) {    // ---- one round
            // round state is wave-uniform by construction; pin it to scalar registers
            q = (int)ufl((uint32_t)q); opos = (int)ufl((uint32_t)opos); nt = ufl(nt); prevty = ufl(prevty);
            // a round adds at most one token per window position; out of token words -> the host grows the pool and repeats
            if (a.tok_cap < kTokCapMax && nt + 64u > a.tok_cap) { overflow = true; break; }
            const int P = q;
            if (lane == 0) __atomic_store_n(&pf_pos, P, __ATOMIC_RELAXED);
            unsigned long long t0 = 0, t1 = 0, t2 = 0;
            if (prof) t0 = __builtin_readcyclecounter();
            // ---------------- phase 1: speculative evaluation of position P + lane
            const int pos = P + lane;
            const bool live = pos < ilen;
            const bool canm = pos + kSentinel < ilen;
            // one round trip for all of this position's text: bytes pos-4 .. pos-1 (context and MRU operands) and
            // pos .. pos+15 (hashes of pos and pos+1, first compare block).  No lane guard: the window ends less than
            // 80 bytes behind the block, inside the 512 readable bytes the boundary requires (include/zlng.h).
            const uint32_t upos = (uint32_t)pos;
            const uint32_t wraw = ld32u(buf + (upos >= 4u ? upos - 4u : 0u));
            const Quad qtext = ld128u(buf + upos);
            const uint32_t t16 = ld32u(buf + (upos + 16u));
            const uint32_t wp = upos >= 4u ? wraw : wraw << ((8u * (4u - upos)) & 31u);
            const uint32_t w4 = live ? qtext.a : 0u;
            const uint32_t ctx = wp >> 24;
            const uint32_t h = hash_of(w4);
            const uint32_t hc = h % kHashSlots, chk = (h / kHashSlots) & 255u;
            const uint32_t kix = key_ix(ctx, hc);
            // MRU operands: as a token start (check key / word) and as a token boundary (event key / word)
            const uint32_t b_m3 = (wp >> 8) & 0xFF, b_m2 = (wp >> 16) & 0xFF, b_0 = w4 & 0xFF, b_1 = (w4 >> 8) & 0xFF;
            const uint32_t cw = b_0 << 8 | b_1;                    // check: mru[ctx] vs (b0, b1)
            const uint32_t ek = b_m3, ew = b_m2 << 8 | ctx;        // event at this boundary: mru[b-3] <- (b-2, b-1)
            const uint32_t evix = ev_ix(ek, ew), chix = ev_ix(ctx, cw);
            const uint32_t evix_w = live ? evix : (uint32_t)kEvTab, ek_w = live ? ek : 256u;
            atomicOr(&evtab[evix_w], lane_bit); atomicOr(&ektab[ek_w], lane_bit);

            Spec S;
            S.sp = kMatchMin - 1; S.node0 = 65535; S.head0 = 0; S.dmin = kRing - 1;
            S.lkix1 = S.lkix2 = S.lctx1 = S.lctx2 = 0; S.lz1 = S.lz2 = false;
            S.ld1 = S.ld2 = kRing - 1;
            S.len0 = 0; S.lsrc1 = 0; S.qa = Quad{0, 0, 0, 0};
            S.off0 = S.off1 = S.olen = 0; S.open = false; S.ov0 = 0; S.lkey1 = 0;
            const uint32_t kix_w = canm ? kix : (uint32_t)kKeyTab, ctx_w = canm ? ctx : 256u;
            atomicOr(&keytab[kix_w], lane_bit);
            atomicOr(&ctxtab[ctx_w], lane_bit);
            // Level 0 speculates on every lane, live or not (no exec-mask region, no default values to materialise): a lane
            // without room for a match (the last 275 bytes of the block) reads at most 354 bytes past the block -- the
            // next block's text or the boundary's 512 readable bytes -- and everything derived from its result is
            // gated by canm below.
            if (level0) speculate_l0w<kWide>(S, dict, buf, heads[ctx], heads[w4 & 0xFF], kRiskDist, pos, qtext, t16, ctx, hc, chk);
            else if (canm) speculate(S, dict, buf, heads[ctx], heads[w4 & 0xFF], heads[(w4 >> 8) & 0xFF], kRiskDist, pos, cfg, qtext, ctx, hc, chk);
            uint32_t sp = (level0 && !canm) ? (uint32_t)(kMatchMin - 1) : S.sp;
            const uint32_t node0 = S.node0, head0 = S.head0, dmin = S.dmin;
            const uint32_t lkix1 = S.lkix1, lkix2 = S.lkix2, lctx1 = S.lctx1, lctx2 = S.lctx2;
            const bool lz1 = S.lz1, lz2 = S.lz2;
            if (prof) t1 = __builtin_readcyclecounter();
            // speculative token of this lane: match (if not vetoed by its speculative lazy probes) or literal
            // (level 0: a lane whose compare ran to 16 bytes is "open": length, node and lazy veto are settled by
            //  finish_open below, and only if the lane turns out to be a token start)
            uint32_t spec_len = sp & kSpLenMask;
            const bool spec_veto = ((sp & kSpVeto1) != 0) || (cfg.lazy2 > 0 && (sp & kSpVeto2) != 0);
            bool spec_match = canm && spec_len >= (uint32_t)kMatchMin && !(spec_len < (uint32_t)kLazyLimit && spec_veto);
            uint32_t tlen = spec_match ? spec_len : 1u;
            unsigned long long match_lanes = __ballot(spec_match);
            const bool is_open = level0 && canm && S.open;
            unsigned long long open_mask = __ballot(is_open);
            // The first open lane of a window is where a long match begins: in text with many long matches nearly always
            // a token start.  Its settling loads go out now and land behind the mask phase; finish_open uses them if the
            // chase does stop there.  (a.settle_pf, same box, same run: real text 894 / 867 / 877 ms for never / always /
            // only after a round that settled a lane; the benchmark text 756 / 753 / 756 -- boxes differ by +-2 %.)
            constexpr uint32_t kOK = kOpenAt;                  // bytes already known equal
            auto open_loads = [&](int L, uint32_t& a4, uint32_t& b4, uint32_t& c4, uint32_t& e4, uint32_t& d4) {
                const uint32_t ol = rl(S.olen, L), offA = rl(S.off0, L), offB = rl(S.off1, L), ls = rl(S.lsrc1, L);
                const uint32_t pL = (uint32_t)(P + L), t4 = 4u * (uint32_t)lane;
                a4 = ld32u(buf + (pL + kOK + t4));
                b4 = ld32u(buf + ((((ol >> 16) & 1u) ? offA : pL) + kOK + t4));
                c4 = ld32u(buf + ((((ol >> 17) & 1u) ? offB : pL) + kOK + t4));
                e4 = ld32u(buf + (pL + kOK - 4u + t4));                                   // lazy probe operands,
                d4 = ld32u(buf + (((ls >> 31) ? (ls & 0xFFFFFF) : pL) + kOK - 4u + t4));  // from byte kOpenAt - 4 on
            };
            int pre_L = -1;
            uint32_t pre_a = 0, pre_b = 0, pre_c = 0, pre_e = 0, pre_d = 0;
            if (open_mask && (a.settle_pf == 1 || (a.settle_pf == 2 && settled_prev))) { pre_L = (int)__builtin_ctzll(open_mask); open_loads(pre_L, pre_a, pre_b, pre_c, pre_e, pre_d); }
            settled_prev = false;
            // eight-token jumps for the chase: next start after 8 tokens and the starts passed on the way (three
            // doubling steps over ds_bpermute; a lone wave pays ~100 cycles per scalar hop otherwise).  The first
            // step's mask is known without a shuffle: the token after mine starts at lane n1, if that lane is live.
            uint32_t hop_next;
            unsigned long long hop_mask;
            {
                auto shfl64 = [](unsigned long long v, uint32_t src) {
                    return (unsigned long long)(uint32_t)__shfl((int)(uint32_t)(v >> 32), (int)src) << 32 | (uint32_t)__shfl((int)(uint32_t)v, (int)src);
                };
                // (an open lane absorbs the chains that reach it: its length is not known yet)
                const uint32_t n1 = live ? (is_open ? (uint32_t)lane : min((uint32_t)lane + tlen, 64u)) : 64u;
                const bool v1 = n1 < 64u && P + (int)n1 < ilen;
                const uint32_t n2g = (uint32_t)__shfl((int)n1, (int)(n1 & 63u));
                const uint32_t n2 = v1 ? n2g : 64u;
                const unsigned long long m2 = (live ? lane_bit : 0ull) | (v1 ? 1ull << (n1 & 63u) : 0ull);
                const bool v2 = n2 < 64u;
                const uint32_t n4g = (uint32_t)__shfl((int)n2, (int)(n2 & 63u));
                const unsigned long long m2g = shfl64(m2, n2 & 63u);
                const uint32_t n4 = v2 ? n4g : 64u;
                const unsigned long long m4 = m2 | (v2 ? m2g : 0ull);
                const bool v4 = n4 < 64u;
                const uint32_t n8g = (uint32_t)__shfl((int)n4, (int)(n4 & 63u));
                const unsigned long long m4g = shfl64(m4, n4 & 63u);
                hop_next = v4 ? n8g : 64u;
                hop_mask = m4 | (v4 ? m4g : 0ull);
            }

            wsync();                         // all lane bits are in the tables
            unsigned long long lkey = 0, lk_mask = 0, lc_mask = 0;    // (lk / lc apart: the lazy-only conflict fix, level 0)
            const unsigned long long keymask_r = keytab[kix_w], ctxmask_r = ctxtab[ctx_w];
            const unsigned long long keymask = canm ? keymask_r : 0ull, ctxmask = canm ? ctxmask_r : 0ull;
            // a lazy probe is invalidated by an accepted insert with its key, or -- if it walked near the
            // ring head -- by any accepted insert into its bucket
            // (at level 0 the probe's read set is kept for every lane: the in-register conflict fix can change a length
            //  and then needs a probe the speculation did not evaluate)
            if (level0) {                            // (lkix1 / lctx1 are 0 for lanes without a speculation: harmless reads)
                const unsigned long long lk = keytab[lkix1], lc = ctxtab[lctx1];
                lk_mask = canm ? lk : 0ull; lc_mask = (canm && (sp & kSpRisk1)) ? lc : 0ull;
                lkey = lk_mask | lc_mask;
            } else if (lz1) lkey |= keytab[lkix1] | ((sp & kSpRisk1) ? ctxtab[lctx1] : 0ull);
            if (lz2) lkey |= keytab[lkix2] | ((sp & kSpRisk2) ? ctxtab[lctx2] : 0ull);
            const unsigned long long hit_r = evtab[chix], same_r = ektab[ek_w];
            const unsigned long long hitmask = live ? hit_r : 0ull;         // boundaries whose (key, word) may equal my check
            const unsigned long long samekey = live ? same_r : 0ull;        // boundaries with my event key
            wsync();
            keytab[kix_w] = 0; ctxtab[ctx_w] = 0;
            evtab[evix_w] = 0; ektab[ek_w] = 0;

            // ---------------- phase 2
            if (prof) { t2 = __builtin_readcyclecounter(); c_p1 += t1 - t0; c_mask += t2 - t1; n_round++; }
            unsigned long long acc = 0;              // accepted token starts of this round (committed or replayed)
            const bool near_cut = opos + 2 * 64 + 4 >= kSubSyms;

            // Settle the open lane L (wave-uniform), all lanes helping: lane t compares bytes kOpenAt+4t .. +3 of the
            // position with both chain nodes' sources (one coalesced round trip for up to 272 bytes each), the best-of
            // rule and the lazy probe of speculate_l0 follow, and lane L's speculative registers take the result.
            // Reads only the block's text and lane L's phase-1 registers, so it can run at any point of the round.
            auto finish_open = [&](int L) {
                unsigned long long tf0 = 0;
                if (prof) tf0 = __builtin_readcyclecounter();
                const uint32_t ol = rl(S.olen, L), ls = rl(S.lsrc1, L);
                const bool lg0 = ((ol >> 16) & 1u) != 0, lg1 = ((ol >> 17) & 1u) != 0, h1s = ((ol >> 18) & 1u) != 0;
                constexpr uint32_t K = kOpenAt;
                uint32_t a4 = pre_a, b4 = pre_b, c4 = pre_c, e4 = pre_e, d4 = pre_d;
                if (L != pre_L) open_loads(L, a4, b4, c4, e4, d4);
                uint32_t l0 = ol & 0xFF, l1 = (ol >> 8) & 0xFF;
                if (prof) { asm volatile("s_waitcnt vmcnt(0)" ::: "memory"); c_finw += __builtin_readcyclecounter() - tf0; }
                if (lg0) {
                    const unsigned long long m = __ballot(a4 != b4);
                    uint32_t n = K + 256u;
                    if (m) { const int f = (int)__builtin_ctzll(m); n = K + 4u * (uint32_t)f + ((uint32_t)__ffs((int)rl(a4 ^ b4, f)) - 1u) / 8u; }
                    l0 = n < (uint32_t)kMatchMax ? n : (uint32_t)kMatchMax;
                }
                if (lg1) {
                    const unsigned long long m = __ballot(a4 != c4);
                    uint32_t n = K + 256u;
                    if (m) { const int f = (int)__builtin_ctzll(m); n = K + 4u * (uint32_t)f + ((uint32_t)__ffs((int)rl(a4 ^ c4, f)) - 1u) / 8u; }
                    l1 = n < (uint32_t)kMatchMax ? n : (uint32_t)kMatchMax;
                }
                uint32_t ml = kMatchMin - 1, mn = 0;
                if (l0 > ml) { ml = l0; mn = rl(node0, L); }
                if (h1s && ml != (uint32_t)kMatchMax && l1 > ml) { ml = l1; mn = (ol >> 19) & (kRing - 1); }
                ml = ufl(ml);
                bool veto = false;
                if (ml < (uint32_t)kLazyLimit && (ls >> 31) != 0) {      // src/libzling_lz.cpp:291-316, depth 1
                    const uint32_t i0 = ml + 2u - K, j0 = ml + 1u - K;    // byte ml-2 of the position / ml-3 of the source, counted from byte K-4
                    const unsigned long long pw = (unsigned long long)rl(e4, (int)(i0 >> 2) + 1) << 32 | rl(e4, (int)(i0 >> 2));
                    const unsigned long long sw = (unsigned long long)rl(d4, (int)(j0 >> 2) + 1) << 32 | rl(d4, (int)(j0 >> 2));
                    veto = (uint32_t)(pw >> (8u * (i0 & 3u))) == (uint32_t)(sw >> (8u * (j0 & 3u)));
                }
                const bool nm = !(ml < (uint32_t)kLazyLimit && veto);
                const uint32_t nsp = ml | mn << kSpNodeShift | kSpCanMatch | (veto ? kSpVeto1 : 0u) | (rl(sp, L) & (kSpRisk1 | kSpRisk2));
                if (lane == L) { sp = nsp; S.len0 = l0; spec_len = ml; spec_match = nm; tlen = nm ? ml : 1u; }
                match_lanes = (match_lanes & ~(1ull << L)) | (nm ? 1ull << L : 0ull);
                open_mask &= ~(1ull << L);
                settled_prev = true;
                if (prof) { n_fin++; c_fin += __builtin_readcyclecounter() - tf0; }
            };

            // exact scalar replay of the token at q (wave-uniform): boundary event, match_exact / word MRU / literal
            // per-lane insert link and match node actually used (the in-register conflict fix may override the speculation)
            uint32_t node0w = node0, mnode = (sp >> kSpNodeShift) & (kRing - 1);
            uint32_t pword = S.ov0;                  // word of the slot node0w (stored beside my own word: the link's copy)
            auto serial_token = [&](bool use_spec) {
                const int sl = q - P;
                const uint32_t xk = rl(ek, sl), xw = rl(ew, sl);
                // (LDS values are wave-uniform here; readfirstlane tells the compiler so, which keeps q / opos /
                //  prevty and with them the whole round control flow in scalar registers)
                if (prevty == kTyMatch) { const uint32_t m = ufl(mru[xk]); if ((m & 0xFFFF) != xw) mru[xk] = (m << 16) | xw; }
                else if (prevty == kTyLit || prevty == kTyW1) { mru[xk] = (ufl(mru[xk]) << 16) | xw; }
                bool is_match = false;
                int mlen = 0, midx = 0;
                const uint32_t spq = rl(sp, sl);
                if (spq & kSpCanMatch) {
                    const uint32_t head = (rl(head0, sl) + (uint32_t)__popcll(rl64(ctxmask, sl) & acc) + 1u) & (kRing - 1);
                    if (use_spec) {                  // speculation validated by the caller: insert + speculative result
                        if (lane == sl) {
                            BucketT<kWide> B(dict, ctx);
                            B.suffix[head] = (uint16_t)node0w;
                            if (kWide) B.slot[head] = (unsigned long long)((uint32_t)pos | chk << 24) | (unsigned long long)pword << 32;
                            else B.offset[head] = (uint32_t)pos | chk << 24;
                            B.hash[hc] = (uint16_t)head;
                        }
                        is_match = ((match_lanes >> sl) & 1ull) != 0;
                        mlen = (int)(spq & kSpLenMask);
                        midx = (int)((head - rl(mnode, sl)) & (kRing - 1));
                    } else {
                        int mi = 0, ml = 0;
                        const bool hit = match_exact<kWide, true>(dict, buf, q, cfg, head, lane == 0, mi, ml);
                        is_match = __builtin_amdgcn_readfirstlane((int)hit) != 0;
                        mlen = __builtin_amdgcn_readfirstlane(ml);
                        midx = __builtin_amdgcn_readfirstlane(mi);
                    }
                }
                acc |= 1ull << sl;
                uint32_t word;
                if (is_match) {                      // src/libzling_lz.cpp:160-167
                    word = (uint32_t)(258 + mlen - kMatchMin) | (uint32_t)midx << 16;
                    opos += 2; q += mlen; prevty = kTyMatch;
                } else {
                    const uint32_t cq = rl(ctx, sl), w = rl(cw, sl);
                    const uint32_t m = ufl(mru[cq]);
                    if (q + 1 < ilen && (m & 0xFFFF) == w) { word = 256; opos++; q += 2; prevty = kTyW0; }          // :172-177
                    else if (q + 1 < ilen && (m >> 16) == w) { word = 257; opos++; q += 2; prevty = kTyW1; }         // :178-184
                    else { word = (w >> 8) | cq << 16; opos++; q++; prevty = kTyLit; }                              // :188-191 (raw; K2 ranks)
                }
                if (lane == 0) tok[nt] = word;
                nt++;
            };

            if (near_cut) {
                // the sub-block is about to fill up (src/libzling_lz.cpp:153): replay token by token
                while (q < P + 64 && q < ilen && opos + 1 < kSubSyms) serial_token(false);
            } else {
                unsigned long long seg = 0;
                while (q < P + 64 && q < ilen) {
                    if (prof) n_seg++;
                    // ---- chase: token starts reachable from q under the speculative lengths.  Chains from
                    // different starts merge, so after a replayed token the previous chase is usually still good.
                    q = (int)ufl((uint32_t)q); opos = (int)ufl((uint32_t)opos); nt = ufl(nt); prevty = ufl(prevty);
                    seg = (unsigned long long)ufl((uint32_t)(seg >> 32)) << 32 | ufl((uint32_t)seg);
                    int s = q - P;
                    unsigned long long tc = 0;
                    if (prof) tc = __builtin_readcyclecounter();
                    if ((seg >> s) & 1ull) seg &= ~((1ull << s) - 1ull);
                    else {
                        seg = 0;
                        while (s < 64) {
                            seg |= rl64(hop_mask, s);
                            int s2 = (int)rl(hop_next, s);
                            if (s2 == s) {                   // an open lane: settle it (once), then its real length leads on
                                if ((open_mask >> s) & 1ull) finish_open(s);
                                s2 = s + (int)rl(tlen, s);
                            }
                            s = s2;
                        }
                    }
                    if (prof) c_chase += __builtin_readcyclecounter() - tc;
                    // ---- validate every lane against (acc | seg); only lanes of seg matter
                    unsigned long long tv = 0;
                    if (prof) tv = __builtin_readcyclecounter();
                    const unsigned long long all = acc | seg;
                    const uint32_t k = (uint32_t)__popcll(ctxmask & all & below);
                    const unsigned long long kq = canm ? (keymask & all & below) : 0ull;   // accepted earlier starts in my hash slot
                    const bool ring = canm && dmin <= k;
                    // ---- same-slot conflicts resolved in registers (level 0).  An earlier accepted start p with my
                    // (ctx, hash13) is now the head of my chain and its predecessor in the slot -- an even earlier such
                    // start p2, else the node my speculation started from -- is the second and last node examined
                    // (depth 2): the true result is best-of {p, p2 | node0} under the reference's rules
                    // (src/libzling_lz.cpp:240-267).  It is taken only if it leaves the token's length unchanged, so
                    // the chase stays valid; everything else falls back to the restart / exact-replay path.
                    bool fixed_ok = false;
                    int pfix = 0;
                    node0w = node0; mnode = (sp >> kSpNodeShift) & (kRing - 1); pword = S.ov0;
                    const bool inseg = (seg & lane_bit) != 0;
                    if (level0 && __any(inseg && kq != 0 && !ring)) {
                        const bool cand = inseg && kq != 0 && !ring;
                        const int p = kq ? top_bit(kq) : 0;
                        const unsigned long long kq2 = kq & ~(1ull << p);
                        const bool has2 = kq2 != 0;
                        const int p2 = has2 ? top_bit(kq2) : 0;
                        const uint32_t key = ctx << 13 | hc;
                        const uint32_t key_p = (uint32_t)__shfl((int)key, p), chk_p = (uint32_t)__shfl((int)chk, p);
                        const Quad qp = {(uint32_t)__shfl((int)S.qa.a, p), (uint32_t)__shfl((int)S.qa.b, p),
                                         (uint32_t)__shfl((int)S.qa.c, p), (uint32_t)__shfl((int)S.qa.d, p)};
                        uint32_t key_p2 = key, chk_p2 = 0;
                        Quad qp2 = {0, 0, 0, 0};
                        if (__any(cand && has2)) {
                            key_p2 = (uint32_t)__shfl((int)key, p2); chk_p2 = (uint32_t)__shfl((int)chk, p2);
                            qp2 = Quad{(uint32_t)__shfl((int)S.qa.a, p2), (uint32_t)__shfl((int)S.qa.b, p2),
                                       (uint32_t)__shfl((int)S.qa.c, p2), (uint32_t)__shfl((int)S.qa.d, p2)};
                        }
                        const bool fixable = cand && key_p == key && (!has2 || key_p2 == key);
                        // candidate of p
                        const bool cp = fixable && chk_p == chk;
                        uint32_t rp = cp ? lcp16(S.qa, qp) : 0u;
                        const bool lp = cp && rp == 16u;
                        // candidate of the second node
                        const bool c2 = fixable && has2 && chk_p2 == chk;
                        uint32_t r2 = c2 ? lcp16(S.qa, qp2) : 0u;
                        const bool l2 = c2 && r2 == 16u;
                        if (__any(lp || l2)) {
                            uint32_t t0, t1;
                            lcp_tail2(buf + pos, buf + P + p, buf + P + p2, lp, l2, t0, t1);
                            rp = lp ? t0 : rp; r2 = l2 ? t1 : r2;
                        }
                        const uint32_t slot_p = (head0 + (uint32_t)__popcll(ctxmask & all & ((1ull << p) - 1ull)) + 1u) & (kRing - 1);
                        const uint32_t slot_p2 = (head0 + (uint32_t)__popcll(ctxmask & all & ((1ull << p2) - 1ull)) + 1u) & (kRing - 1);
                        const bool second = has2 || node0 != 65535u;
                        const uint32_t rs = has2 ? r2 : S.len0, ns = has2 ? slot_p2 : node0;
                        uint32_t ml = kMatchMin - 1, mn = 0;
                        if (rp > ml) { ml = rp; mn = slot_p; }
                        if (ml != (uint32_t)kMatchMax && second && rs > ml) { ml = rs; mn = ns; }
                        // lazy probe under the (possibly different) length; its own read set must be clean
                        const bool lzn = ml >= (uint32_t)kMatchMin && ml < (uint32_t)kLazyLimit;
                        const bool lclean = (lkey & all & beloweq) == 0 || !lzn;
                        bool veto = (sp & kSpVeto1) != 0;
                        const bool reprobe = fixable && lzn && lclean && ml != spec_len;
                        if (__any(reprobe)) {
                            const uint32_t mm = reprobe ? ml - 3u : 0u;
                            const uint32_t pr = ld32u(buf + pos + 1 + mm);
                            const uint32_t sr = ld32u(buf + (reprobe ? (S.lsrc1 & 0xFFFFFF) + mm : (uint32_t)pos));
                            if (reprobe) veto = (S.lsrc1 >> 31) != 0 && pr == sr;
                        }
                        const bool nmatch = ml >= (uint32_t)kMatchMin && !(lzn && veto);
                        fixed_ok = fixable && lclean && nmatch == spec_match && (!nmatch || ml == spec_len);
                        if (fixed_ok) { node0w = slot_p; mnode = mn; pfix = p; pword = (uint32_t)(P + p) | chk_p << 24; }
                    }
                    const bool dirty = canm && (kq != 0 || ring) && !fixed_ok;
                    const bool ldirty = !fixed_ok && spec_len >= (uint32_t)kMatchMin && spec_len < (uint32_t)kLazyLimit && (lkey & all & beloweq) != 0;
                    const uint32_t m0 = mru[ctx];
                    const bool poss = !spec_match && pos + 1 < ilen &&
                                      ((m0 & 0xFFFF) == cw || (m0 >> 16) == cw || (hitmask & all & beloweq) != 0);
                    unsigned long long prob = seg & __ballot(dirty || ldirty || poss);
                    // ---- lazy-only conflicts resolved in registers (level 0), one at a time and only when such a lane is the
                    // first problem of the segment.  The lane's own match stands (no accepted start wrote its hash slot or a
                    // ring entry it read) but ONE accepted start p <= lane -- its own insert included, src/libzling_lz.cpp:271
                    // -- carries the lane-mask bit of its lazy probe's key.  If p's key is that key exactly, p's insert is the
                    // chain head the probe sees (depth 1: the only node it looks at) and the veto is 4 bytes of text against 4
                    // bytes of text; if it is another key (the mask table is hashed) nothing the probe read has changed.  The
                    // lane is cleared only if the veto comes out as speculated, so the chase stays valid.
                    if (level0 && a.lazy_fix && prob) {
                        const unsigned long long lhit = lk_mask & all & beloweq;
                        const bool lcand = ldirty && !dirty && !poss && (lhit & (lhit - 1ull)) == 0 && lhit != 0 && (lc_mask & all & beloweq) == 0;
                        const unsigned long long lcm = seg & __ballot(lcand);
                        while (prob) {
                            const int f0 = (int)__builtin_ctzll(prob);
                            if (!((lcm >> f0) & 1ull)) break;
                            const int p = (int)__builtin_ctzll(rl64(lhit, f0));
                            bool ok = true;
                            if (rl(ctx << 13 | hc, p) == rl(S.lkey1, f0)) {
                                const uint32_t mm = rl(spec_len, f0) - 3u;
                                const uint32_t pr = ld32u(buf + ((uint32_t)(P + f0) + 1u + mm));
                                const uint32_t sr = ld32u(buf + ((uint32_t)(P + p) + mm));
                                ok = (ufl(pr) == ufl(sr)) == ((rl(sp, f0) & kSpVeto1) != 0);
                            }
                            if (!ok) break;
                            prob &= ~(1ull << f0);
                            if (prof) n_lfix++;
                        }
                    }
                    const int f = prob ? (int)__builtin_ctzll(prob) : 64;
                    const unsigned long long com = f >= 64 ? seg : (seg & ((1ull << f) - 1ull));
                    unsigned long long tcm = 0;
                    if (prof) { tcm = __builtin_readcyclecounter(); c_val += tcm - tv; }
                    if (com) {
                        const bool mine = (com & lane_bit) != 0;
                        // ---- MRU events of the committed boundaries (lane-parallel 2-slot push rules)
                        const int first = (int)__builtin_ctzll(com);
                        const unsigned long long prev_m = com & below;            // earlier committed lanes
                        const int pj = prev_m ? top_bit(prev_m) : 0;
                        const uint32_t pty = prev_m ? (((match_lanes >> pj) & 1ull) ? kTyMatch : kTyLit) : prevty;
                        const bool is_ev = mine && (pty == kTyMatch || pty == kTyLit || pty == kTyW1);
                        const unsigned long long evs = __ballot(is_ev);
                        const uint32_t m0e = mru[ek];
                        const unsigned long long before_k = samekey & evs & below;
                        const int pe = before_k ? top_bit(before_k) : 0;
                        const uint32_t ew_pe = (uint32_t)__shfl((int)ew, pe);
                        const uint32_t s0b = before_k ? ew_pe : (m0e & 0xFFFF);    // slot 0 just before my event
                        const bool eff = is_ev && (pty != kTyMatch || ew != s0b);
                        const unsigned long long effs = __ballot(eff);
                        const unsigned long long upto = samekey & effs & beloweq;
                        const int es = upto ? top_bit(upto) : 0;
                        const uint32_t s0b_es = (uint32_t)__shfl((int)s0b, es);
                        const bool last_of_key = is_ev && (samekey & evs & ~beloweq) == 0;
                        if (last_of_key) mru[ek] = ew | (upto ? s0b_es : (m0e >> 16)) << 16;
                        (void)first;
                        // ---- dictionary inserts (src/libzling_lz.cpp:227-230) and token words
                        const uint32_t head = (head0 + k + 1u) & (kRing - 1);
                        bool head_writer = true;
                        if (__any(mine && fixed_ok)) {               // a fixed lane's predecessor p has the same exact key
                            if (lane == 0) pred_mask = 0;
                            wsync();
                            if (mine && fixed_ok) atomicOr(&pred_mask, 1ull << pfix);
                            wsync();
                            head_writer = ((pred_mask >> lane) & 1ull) == 0;
                        }
                        if (mine) {
                            uint32_t word;
                            if (canm) {
                                BucketT<kWide> B(dict, ctx);
                                B.suffix[head] = (uint16_t)node0w;
                                if (kWide) B.slot[head] = (unsigned long long)((uint32_t)pos | chk << 24) | (unsigned long long)pword << 32;
                                else B.offset[head] = (uint32_t)pos | chk << 24;
                                // several starts of one hash slot can commit together now; the slot's head must end up
                                // being the last of them, so a lane that is the predecessor of a later one does not write it
                                if (head_writer) B.hash[hc] = (uint16_t)head;
                            }
                            if (spec_match) word = (258u + spec_len - kMatchMin) | ((head - mnode) & (kRing - 1)) << 16;
                            else word = b_0 | ctx << 16;
                            __builtin_nontemporal_store(word, &tok[nt + (uint32_t)__popcll(com & below)]);   // streamed out: keep L2 for the dictionary
                        }
                        const int lastl = top_bit(com);
                        const bool last_match = ((match_lanes >> lastl) & 1ull) != 0;
                        nt += (uint32_t)__popcll(com);
                        opos += __popcll(com) + __popcll(com & match_lanes);
                        acc |= com;
                        q = P + lastl + (int)rl(tlen, lastl);
                        prevty = last_match ? kTyMatch : kTyLit;
                    }
                    if (prof) c_com += __builtin_readcyclecounter() - tcm;
                    if (f < 64) {
                        const bool conflict = rl((dirty || ldirty) ? 1u : 0u, f) != 0;
                        if (prof && conflict) {
                            const uint32_t cls = rl((canm && (keymask & all & below) != 0 ? 1u : 0u) | (canm && dmin <= k ? 2u : 0u) | (ldirty ? 4u : 0u), f);
                            if (cls & 1u) n_cA++; else if (cls & 2u) n_cB++; else n_cL++;
                        }
                        const int q_before = q; const uint32_t nt_before = nt;
                        unsigned long long ts = 0;
                        if (prof) { if (conflict) n_redo++; else n_poss++; ts = __builtin_readcyclecounter(); }
                        // A conflict with an earlier start of this round disappears when the round restarts at
                        // that token (its speculation then sees every committed insert); only a token that opens
                        // the round and still conflicts (with its own insert / ring slot) needs the exact replay.
                        if (conflict && f >= a.min_restart) break;
                        serial_token(!conflict);
                        if (prof) c_ser += __builtin_readcyclecounter() - ts;
                        if (prof && conflict) {      // did the exact replay agree with the speculation?
                            n_replay++;
                            const int sl2 = q_before - P;
                            const uint32_t spec_adv = rl(tlen, sl2);
                            const bool spec_m = ((match_lanes >> sl2) & 1ull) != 0;
                            const bool got_m = prevty == kTyMatch;
                            if (spec_m == got_m && (!got_m || (int)spec_adv == q - q_before)) n_same++;
                            (void)nt_before;
                        }
                    }
                }
            }
            if (prof) c_p2 += __builtin_readcyclecounter() - t2;
            // publish the ring heads advanced by this round (every accepted lane of a context writes the same value)
            if (canm && (acc & lane_bit)) heads[ctx] = (uint16_t)((head0 + (uint32_t)__popcll(ctxmask & acc)) & (kRing - 1));
            wsync();
        }
        if (nsub < kMaxSub && lane == 0) cuts[nsub] = SubCut{tok_begin, nt, (uint32_t)q, (uint32_t)opos};
        nsub++;
    }
    if (lane == 0) {
        __atomic_store_n(&pf_done, 1, __ATOMIC_RELAXED);
        if (overflow) { *a.overflow = 1; nsub = 0; nt = 0; }
        a.nsub[blk] = (uint32_t)nsub; a.ntok[blk] = nt;
    }
    if (prof && lane == 0) {
        unsigned long long* d = a.dbg + (size_t)blk * kDbgSlots;
        d[0] = c_p1; d[1] = c_mask; d[2] = c_p2; d[3] = n_round; d[4] = nt; d[5] = n_seg; d[6] = n_redo; d[7] = n_poss; d[8] = c_ser; d[9] = c_chase; d[10] = n_cA; d[11] = n_cB; d[12] = n_cL; d[13] = n_replay; d[14] = n_same; d[15] = c_val; d[16] = c_com; d[17] = n_fin; d[18] = c_fin; d[19] = c_finw; d[20] = n_lfix;
    }
}

void launch_rolz_parse_wave(const ParseArgs& a, uint32_t nblocks_all, hipStream_t s, bool all_level0) {
    const bool prof = a.dbg != nullptr;
    const uint32_t nblocks = nblocks_all - a.blk0;
    if (all_level0 && !prof) hipLaunchKernelGGL((k_rolz_parse_wave<true, false>), dim3(nblocks), dim3(64 * (1 + a.pf_waves)), 0, s, a);
    else if (all_level0) hipLaunchKernelGGL((k_rolz_parse_wave<true, true>), dim3(nblocks), dim3(64 * (1 + a.pf_waves)), 0, s, a);
    else if (!prof) hipLaunchKernelGGL((k_rolz_parse_wave<false, false>), dim3(nblocks), dim3(64 * (1 + a.pf_waves)), 0, s, a);
    else hipLaunchKernelGGL((k_rolz_parse_wave<false, true>), dim3(nblocks), dim3(64 * (1 + a.pf_waves)), 0, s, a);
}

}  // namespace zlng
