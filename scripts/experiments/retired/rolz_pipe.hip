// rolz_pipe.hip -- K1, pipelined form: the ROLZ block parser with speculation and resolution on different wavefronts.
//
// Same semantics as k_rolz_parse_wave (rolz_parse.hip): ZlingRolzEncoder::Encode / EncodeImpl / MatchAndUpdate /
// MatchLazy, src/libzling_lz.cpp:128-316.  What changes is who does what.  A lone wavefront issues one instruction per
// ~5 cycles, and a window of 64 positions cost it ~2,400 cycles of speculation (five dependent memory round trips),
// ~1,100 cycles of lane-mask / jump tables and ~3,700 cycles of resolution, one after the other.  Here the block's
// workgroup runs three wavefronts:
//
//   wave 1  EVALUATOR   for windows on a fixed 64-position grid, ahead of the resolver: the speculative evaluation of
//                       every position as if it were a token start (speculate / speculate_l0), the lane-mask tables
//                       (same (ctx, hash13) key, same context, MRU keys) and the eight-token jump table -- everything
//                       that depends only on the text and on a READ of the dictionary.  Results go to an LDS slot.
//   wave 0  RESOLVER    takes the slots in order and does what must be serial: chase of the true token chain,
//                       validation, commit (dictionary inserts, token words, MRU), exact replay of problem tokens.
//                       It is the only writer of the dictionary, the ring heads and the MRU.
//   wave 2  PREFETCHER  (optional) runs the evaluator's loads one or two windows further ahead and discards them.
//
// Exactness.  A slot was computed from a dictionary that may lack the inserts of the windows the resolver committed
// after the evaluator started (at most `lead` of them).  The resolver therefore treats a lane's speculation like one
// invalidated inside its own window whenever something it read may have changed since:
//   * the evaluator notes D0 = number of completed windows BEFORE its first dictionary load (acquire; the resolver
//     publishes it with release after the window's stores) -- inserts of windows < D0 are visible to it;
//   * ts_key[key_ix] = last window that inserted a token start with that (ctx, hash13) key bucket: a lane whose own key
//     or lazy-probe key was inserted by a window in [D0, j) is dirty;
//   * ring slots are handed out consecutively per context, so the slots written since the evaluator read the context's
//     head are head0+1 .. head0+kb (+ this window's): a lane that visited one of them (dmin <= kb + k) is dirty, and so
//     is a lazy probe that visited one of the slots its bucket has handed out since (ld < kbl + kl).
// A dirty lane that turns out to be a token start is replayed by the exact scalar code (match_exact) against the
// current dictionary, exactly like an in-window conflict -- so the result is the reference's in every case, whatever
// the evaluator saw.  (The ring heads are published AFTER `done`: heads the evaluator reads are never newer than the
// state D0 covers; older ones only make kb larger.)
#include "zlng_common.h"
#include "zlng_kernels.h"
#include "rolz_dev.h"

namespace zlng {

constexpr int kSlots = 4;                         // windows in flight between evaluator and resolver
constexpr int kSlotQ = 8;                         // 16-byte records per lane and slot
constexpr int kTsTab = 16384;                     // key buckets of the "inserted since" table (a false hit only costs a replay)
__device__ __forceinline__ uint32_t ts_ix(uint32_t ctx, uint32_t hc) { return (hc ^ (ctx * 0x2D1u)) & (kTsTab - 1); }

struct Q4 { uint32_t x, y, z, w; };

__device__ __forceinline__ uint32_t slot_tag(int win, int level) { return ((uint32_t)win + 1u) << 3 | (uint32_t)level; }

#define WG_LOAD(p)      __hip_atomic_load((p), __ATOMIC_ACQUIRE, __HIP_MEMORY_SCOPE_WORKGROUP)
#define WG_LOAD_RLX(p)  __hip_atomic_load((p), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP)
#define WG_STORE(p, v)  __hip_atomic_store((p), (v), __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_WORKGROUP)
#define WG_STORE_RLX(p, v) __hip_atomic_store((p), (v), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP)

template <bool kAllL0>
__global__ __launch_bounds__(512) void k_rolz_parse_pipe(ParseArgs a) {
    __shared__ uint16_t heads[256];
    __shared__ uint32_t mru[256];                    // slot0 | slot1 << 16
    // evaluator's lane-mask tables (extra last entry = sink for lanes past the block end, see k_rolz_parse_wave)
    __shared__ unsigned long long keytab[kKeyTab + 1];
    __shared__ unsigned long long ctxtab[256 + 1];
    __shared__ unsigned long long evtab[kEvTab + 1];
    __shared__ unsigned long long ektab[256 + 1];
    __shared__ unsigned long long pred_mask;
    __shared__ Q4 slotq[kSlots][kSlotQ][64];
    __shared__ uint32_t tag[kSlots];                 // slot_tag(window, level) once the slot is complete
    __shared__ uint32_t slot_d0[kSlots];             // completed windows when the slot's evaluation started
    __shared__ uint16_t ts_key[kTsTab];              // 1 + last window that inserted a start with this key bucket (mod 2^16)
    __shared__ int need, want_level, fin, ev_win;
    __shared__ uint32_t pub;                         // (round counter) << 20 | completed windows; release-stored by the resolver every round
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
    const int lead = a.pf_ahead < 1 ? 1 : (a.pf_ahead > kSlots - 1 ? kSlots - 1 : a.pf_ahead);

    if (wave == 0) {
        for (int i = lane; i < 256 + 1; i += 64) { if (i < 256) heads[i] = 0; ctxtab[i] = 0; ektab[i] = 0; }
        for (int i = lane; i < kKeyTab + 1; i += 64) { keytab[i] = 0; evtab[i] = 0; }
        for (int i = lane; i < kTsTab; i += 64) ts_key[i] = 0;
        if (lane < kSlots) { tag[lane] = 0; slot_d0[lane] = 0; }
        if (lane == 0) { need = 0; pub = 0; want_level = a.lvl_sched[blk * kMaxSub]; fin = 0; ev_win = 0; }
    }
    __syncthreads();                                 // the only workgroup barrier

    // =============================================================================== evaluator / prefetcher
    if (wave >= 1) {
        const bool evaluator = wave == 1;
        while (true) {
            if (WG_LOAD_RLX(&fin)) break;
            int w;
            int lvl = WG_LOAD_RLX(&want_level);
            if (evaluator) {
                const int nd = WG_LOAD_RLX(&need);
                w = nd;
                while (w < nd + lead && WG_LOAD_RLX(&tag[w & (kSlots - 1)]) == slot_tag(w, lvl)) w++;
                if (w >= nd + lead || (long long)w * 64 >= ilen) { __builtin_amdgcn_s_sleep(1); continue; }
                // the resolver publishes `done` for the windows before `need` shortly after it starts on `need`: wait for it,
                // so that a window evaluated one ahead is stale by the resolver's current window only
                if (w > nd && (int)(WG_LOAD_RLX(&pub) & 0xFFFFFu) < nd) { __builtin_amdgcn_s_sleep(1); continue; }
                WG_STORE_RLX(&ev_win, w);
            } else {
                // prefetcher: the window(s) behind the one the evaluator is working on
                static_assert(kSlots >= 2, "");
                const int e = WG_LOAD_RLX(&ev_win);
                w = e + (wave - 1);
                if ((long long)w * 64 >= ilen) { __builtin_amdgcn_s_sleep(8); continue; }
            }
            const LevelCfg cfg = kAllL0 ? level_cfg(0) : level_cfg(lvl);
            const bool level0 = kAllL0 || (cfg.depth == 2 && cfg.lazy1 == 1 && cfg.lazy2 == 0);
            const int P = w << 6;
            const int pos = P + lane;
            const bool live = pos < ilen;
            const bool canm = pos + kSentinel < ilen;
            const uint32_t upos = (uint32_t)pos;
            const uint32_t wraw = ld32u(buf + (upos >= 4u ? upos - 4u : 0u));
            const Quad qtext = ld128u(buf + upos);
            const uint32_t wp = upos >= 4u ? wraw : wraw << ((8u * (4u - upos)) & 31u);
            const uint32_t w4 = live ? qtext.a : 0u;
            const uint32_t ctx = wp >> 24;
            const uint32_t h = hash_of(w4);
            const uint32_t hc = h % kHashSlots, chk = (h / kHashSlots) & 255u;
            const uint32_t lc1 = qtext.a & 0xFF, lc2 = (qtext.a >> 8) & 0xFF;
            if (!evaluator) {
                // same loads as the evaluator's, results discarded (warms L2 / L1 for the dependent chain)
                if (pos >= 4 && canm) {
                    Spec S;
                    if (level0) speculate_l0(S, dict, buf, heads[ctx], heads[lc1], 0u, pos, qtext, ctx, hc, chk);
                    else speculate(S, dict, buf, heads[ctx], heads[lc1], heads[lc2], 0u, pos, cfg, qtext, ctx, hc, chk);
                    asm volatile("" :: "v"(S.sp), "v"(S.dmin), "v"(S.node0));
                }
                // one window per evaluator step is enough: wait until the evaluator moves on
                while (!WG_LOAD_RLX(&fin) && WG_LOAD_RLX(&ev_win) + (wave - 1) <= w) __builtin_amdgcn_s_sleep(4);
                continue;
            }
            // ---- text-only part: MRU operands and lane masks
            const uint32_t kix = key_ix(ctx, hc);
            const uint32_t b_m3 = (wp >> 8) & 0xFF, b_m2 = (wp >> 16) & 0xFF, b_0 = w4 & 0xFF, b_1 = (w4 >> 8) & 0xFF;
            const uint32_t cw = b_0 << 8 | b_1;                    // check: mru[ctx] vs (b0, b1)
            const uint32_t ek = b_m3, ew = b_m2 << 8 | ctx;        // event at this boundary: mru[b-3] <- (b-2, b-1)
            const uint32_t evix = ev_ix(ek, ew), chix = ev_ix(ctx, cw);
            const uint32_t evix_w = live ? evix : (uint32_t)kEvTab, ek_w = live ? ek : 256u;
            atomicOr(&evtab[evix_w], lane_bit); atomicOr(&ektab[ek_w], lane_bit);
            const uint32_t kix_w = canm ? kix : (uint32_t)kKeyTab, ctx_w = canm ? ctx : 256u;
            atomicOr(&keytab[kix_w], lane_bit);
            atomicOr(&ctxtab[ctx_w], lane_bit);
            // ---- dictionary part.  Order matters: ring heads first, then D0 (acquire), then the dictionary loads.
            Spec S;
            S.sp = kMatchMin - 1; S.node0 = 65535; S.head0 = 0; S.dmin = kRing - 1;
            S.lkix1 = S.lkix2 = S.lctx1 = S.lctx2 = 0; S.lz1 = S.lz2 = false;
            S.ld1 = S.ld2 = kRing - 1;
            S.len0 = 0; S.lsrc1 = 0; S.qa = Quad{0, 0, 0, 0};
            // the heads are read BEFORE D0 and handed to the speculation, which measures every ring distance against them
            const uint32_t hd_main = heads[ctx], hd_l1 = heads[lc1], hd_l2 = heads[lc2];
            asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
            const int d0 = (int)(WG_LOAD(&pub) & 0xFFFFFu);
            if (level0) speculate_l0(S, dict, buf, hd_main, hd_l1, 0u, pos, qtext, ctx, hc, chk);
            else if (canm) speculate(S, dict, buf, hd_main, hd_l1, hd_l2, 0u, pos, cfg, qtext, ctx, hc, chk);
            const uint32_t sp = (level0 && !canm) ? (uint32_t)(kMatchMin - 1) : S.sp;
            const uint32_t dmin_adj = S.dmin;
            const uint32_t spec_len = sp & kSpLenMask;
            const bool spec_veto = ((sp & kSpVeto1) != 0) || (cfg.lazy2 > 0 && (sp & kSpVeto2) != 0);
            const bool spec_match = canm && spec_len >= (uint32_t)kMatchMin && !(spec_len < (uint32_t)kLazyLimit && spec_veto);
            const uint32_t tlen = spec_match ? spec_len : 1u;
            uint32_t hop_next;
            unsigned long long hop_mask;
            {
                auto shfl64 = [](unsigned long long v, uint32_t src) {
                    return (unsigned long long)(uint32_t)__shfl((int)(uint32_t)(v >> 32), (int)src) << 32 | (uint32_t)__shfl((int)(uint32_t)v, (int)src);
                };
                const uint32_t n1 = live ? min((uint32_t)lane + tlen, 64u) : 64u;
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
            const unsigned long long keymask_r = keytab[kix_w], ctxmask_r = ctxtab[ctx_w];
            const unsigned long long keymask = canm ? keymask_r : 0ull, ctxmask = canm ? ctxmask_r : 0ull;
            // in-window read sets of the lazy probes: lanes with the probe's key, lanes inserting into the probe's bucket
            // (at level 0 they are kept for every lane: the in-register conflict fix may need a probe the speculation skipped)
            const bool use1 = canm && (level0 || S.lz1), use2 = canm && S.lz2;
            const unsigned long long lk1 = keytab[S.lkix1], lcm1_r = ctxtab[S.lctx1], lk2 = keytab[S.lkix2], lcm2_r = ctxtab[S.lctx2];
            const unsigned long long lkey = (use1 ? lk1 : 0ull) | (use2 ? lk2 : 0ull);
            const unsigned long long lcm1 = use1 ? lcm1_r : 0ull, lcm2 = use2 ? lcm2_r : 0ull;
            const unsigned long long hit_r = evtab[chix], same_r = ektab[ek_w];
            const unsigned long long hitmask = live ? hit_r : 0ull;
            const unsigned long long samekey = live ? same_r : 0ull;
            wsync();
            keytab[kix_w] = 0; ctxtab[ctx_w] = 0;
            evtab[evix_w] = 0; ektab[ek_w] = 0;
            // ---- hand over
            const uint32_t tsl1 = ts_ix(lc1, hash_of(qtext.a >> 8 | qtext.b << 24) % kHashSlots);
            const uint32_t tsl2 = ts_ix(lc2, hash_of(qtext.a >> 16 | qtext.b << 16) % kHashSlots);
            const int sl = w & (kSlots - 1);
            const uint32_t flags = (live ? 1u : 0u) | (canm ? 2u : 0u) | (S.lz1 ? 4u : 0u) | (S.lz2 ? 8u : 0u);
            slotq[sl][0][lane] = Q4{sp, S.node0 | dmin_adj << 16, hd_main | hd_l1 << 12 | S.lctx1 << 24, tsl1 | S.len0 << 14 | hop_next << 23};
            slotq[sl][1][lane] = Q4{S.lsrc1, ctx | chk << 8 | cw << 16, hc | ek << 16 | b_0 << 24, ew | flags << 16};
            slotq[sl][2][lane] = Q4{S.qa.a, S.qa.b, S.qa.c, S.qa.d};
            slotq[sl][3][lane] = Q4{(uint32_t)keymask, (uint32_t)(keymask >> 32), (uint32_t)ctxmask, (uint32_t)(ctxmask >> 32)};
            slotq[sl][4][lane] = Q4{(uint32_t)lkey, (uint32_t)(lkey >> 32), (uint32_t)hitmask, (uint32_t)(hitmask >> 32)};
            slotq[sl][5][lane] = Q4{(uint32_t)samekey, (uint32_t)(samekey >> 32), (uint32_t)hop_mask, (uint32_t)(hop_mask >> 32)};
            slotq[sl][6][lane] = Q4{(uint32_t)lcm1, (uint32_t)(lcm1 >> 32), S.ld1 | S.ld2 << 12, hd_l2 | S.lctx2 << 12};
            if (!kAllL0) slotq[sl][7][lane] = Q4{(uint32_t)lcm2, (uint32_t)(lcm2 >> 32), tsl2, 0};
            if (lane == 0) slot_d0[sl] = (uint32_t)d0;
            wsync();
            if (lane == 0) WG_STORE(&tag[sl], slot_tag(w, lvl));
        }
        return;
    }

    // =============================================================================== resolver
    uint32_t nt = 0;
    int q = 0, nsub = 0;
    bool overflow = false;
    bool pend_upd = false;                           // ring heads advanced by the previous round, not yet in `heads`
    uint32_t pend_ctx = 0, pend_nh = 0, round_no = 0;
    unsigned long long c_wait = 0, c_p2 = 0, n_round = 0, n_redo = 0, n_poss = 0, n_stale = 0, n_seg = 0;
    const bool prof = a.dbg != nullptr;

    while (q < ilen && !overflow) {                  // ---- one sub-block (one EncodeImpl call)
        const int cur_level = kAllL0 ? 0 : (int)a.lvl_sched[blk * kMaxSub + (nsub < kMaxSub ? nsub : kMaxSub - 1)];
        const LevelCfg cfg = level_cfg(cur_level);
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

        while (q < ilen && opos + 1 < kSubSyms) {    // ---- one round = the rest of one grid window
            q = (int)ufl((uint32_t)q); opos = (int)ufl((uint32_t)opos); nt = ufl(nt); prevty = ufl(prevty);
            if (a.tok_cap < kTokCapMax && nt + 64u > a.tok_cap) { overflow = true; break; }
            const int j = q >> 6;
            const int P = j << 6;
            const int pos = P + lane;
            unsigned long long t0 = 0, t1 = 0;
            if (prof) t0 = __builtin_readcyclecounter();
            // ---------------- take the window's slot
            if (lane == 0) { WG_STORE_RLX(&want_level, cur_level); WG_STORE_RLX(&need, j); }
            const int sl = j & (kSlots - 1);
            const uint32_t want = slot_tag(j, cur_level);
            while (WG_LOAD(&tag[sl]) != want) __builtin_amdgcn_s_sleep(1);
            const Q4 r0 = slotq[sl][0][lane], r1 = slotq[sl][1][lane], r2 = slotq[sl][2][lane], r3 = slotq[sl][3][lane],
                     r4 = slotq[sl][4][lane], r5 = slotq[sl][5][lane], r6 = slotq[sl][6][lane];
            Q4 r7 = Q4{0, 0, 0, 0};
            if (!kAllL0) r7 = slotq[sl][7][lane];
            const uint32_t d0 = ufl(slot_d0[sl]);
            // every window before this one is complete; its stores were issued a slot-wait ago, so the release (which has to
            // see them out) rarely stalls here -- at the end of the previous round it would have cost a store round trip
            // and the release happens EVERY round (a round may have ended inside its window), followed by the ring heads the
            // previous round advanced: heads the evaluator reads are never newer than what its acquire of `pub` covers
            round_no++;
            if (lane == 0) WG_STORE(&pub, (round_no & 0xFFFu) << 20 | (uint32_t)j);
            wsync();
            if (pend_upd) heads[pend_ctx] = (uint16_t)pend_nh;
            pend_upd = false;
            wsync();
            if (prof) { t1 = __builtin_readcyclecounter(); c_wait += t1 - t0; n_round++; }
            const uint32_t sp = r0.x, node0 = r0.y & 0xFFFF, dmin = r0.y >> 16;
            const uint32_t head0 = r0.z & 0xFFF, lhead1 = (r0.z >> 12) & 0xFFF, lctx1 = r0.z >> 24;
            const uint32_t tsl1 = r0.w & 0x3FFF, len0 = (r0.w >> 14) & 0x1FF, hop_next = r0.w >> 23;
            const uint32_t lsrc1 = r1.x, ctx = r1.y & 0xFF, chk = (r1.y >> 8) & 0xFF, cw = r1.y >> 16;
            const uint32_t hc = r1.z & 0x1FFF, ek = (r1.z >> 16) & 0xFF, b_0 = r1.z >> 24, ew = r1.w & 0xFFFF;
            const bool live = (r1.w >> 16 & 1u) != 0, canm = (r1.w >> 17 & 1u) != 0, lz1 = (r1.w >> 18 & 1u) != 0, lz2 = (r1.w >> 19 & 1u) != 0;
            const Quad qa = Quad{r2.x, r2.y, r2.z, r2.w};
            const unsigned long long keymask = (unsigned long long)r3.y << 32 | r3.x, ctxmask = (unsigned long long)r3.w << 32 | r3.z;
            const unsigned long long lkey = (unsigned long long)r4.y << 32 | r4.x, hitmask = (unsigned long long)r4.w << 32 | r4.z;
            const unsigned long long samekey = (unsigned long long)r5.y << 32 | r5.x, hop_mask = (unsigned long long)r5.w << 32 | r5.z;
            const unsigned long long lcm1 = (unsigned long long)r6.y << 32 | r6.x, lcm2 = (unsigned long long)r7.y << 32 | r7.x;
            const uint32_t ld1 = r6.z & 0xFFF, ld2 = (r6.z >> 12) & 0xFFF;
            const uint32_t lhead2 = r6.w & 0xFFF, lctx2 = (r6.w >> 12) & 0xFF, tsl2 = r7.z;
            const uint32_t tsk = ts_ix(ctx, hc);
            (void)live;
            // ---------------- what changed since the evaluator looked: windows d0 .. j-1, and earlier rounds of window j
            // itself (a sub-block cut ends a round inside its window; the next round takes the same slot again)
            const uint32_t hbase = heads[ctx];                                   // my context's ring head now
            const uint32_t kb = (hbase - head0) & (kRing - 1);                   // slots handed out in it since
            const uint16_t span = (uint16_t)((uint32_t)j + 1u - d0);
            auto newer = [&](uint32_t ix) { return (uint16_t)((uint32_t)ts_key[ix] - (d0 + 1u)) < span; };
            const bool xkey = canm && newer(tsk);
            const bool use1 = canm && (level0 || lz1), use2 = canm && lz2;
            const bool xl = (use1 && newer(tsl1)) || (use2 && newer(tsl2));     // a lazy probe's key was inserted since
            const uint32_t kbl1 = use1 ? (heads[lctx1] - lhead1) & (kRing - 1) : 0u;   // slots its bucket handed out since
            const uint32_t kbl2 = use2 ? (heads[lctx2] - lhead2) & (kRing - 1) : 0u;

            const uint32_t spec_len = sp & kSpLenMask;
            const bool spec_veto = ((sp & kSpVeto1) != 0) || (cfg.lazy2 > 0 && (sp & kSpVeto2) != 0);
            const bool spec_match = canm && spec_len >= (uint32_t)kMatchMin && !(spec_len < (uint32_t)kLazyLimit && spec_veto);
            const uint32_t tlen = spec_match ? spec_len : 1u;
            const unsigned long long match_lanes = __ballot(spec_match);

            // ---------------- resolution (as k_rolz_parse_wave's phase 2)
            unsigned long long acc = 0;              // accepted token starts of this round (committed or replayed)
            const bool near_cut = opos + 2 * 64 + 4 >= kSubSyms;
            uint32_t node0w = node0, mnode = (sp >> kSpNodeShift) & (kRing - 1);
            auto serial_token = [&](bool use_spec) {
                const int s1 = q - P;
                const uint32_t xk = rl(ek, s1), xw = rl(ew, s1);
                if (prevty == kTyMatch) { const uint32_t m = ufl(mru[xk]); if ((m & 0xFFFF) != xw) mru[xk] = (m << 16) | xw; }
                else if (prevty == kTyLit || prevty == kTyW1) { mru[xk] = (ufl(mru[xk]) << 16) | xw; }
                bool is_match = false;
                int mlen = 0, midx = 0;
                const uint32_t spq = rl(sp, s1);
                if (spq & kSpCanMatch) {
                    const uint32_t head = (rl(hbase, s1) + (uint32_t)__popcll(rl64(ctxmask, s1) & acc) + 1u) & (kRing - 1);
                    if (use_spec) {                  // speculation validated by the caller: insert + speculative result
                        if (lane == s1) {
                            Bucket B(dict, ctx);
                            B.suffix[head] = (uint16_t)node0w;
                            B.offset[head] = (uint32_t)pos | chk << 24;
                            B.hash[hc] = (uint16_t)head;
                        }
                        is_match = ((match_lanes >> s1) & 1ull) != 0;
                        mlen = (int)(spq & kSpLenMask);
                        midx = (int)((head - rl(mnode, s1)) & (kRing - 1));
                    } else {
                        int mi = 0, ml = 0;
                        const bool hit = match_exact(dict, buf, q, cfg, head, lane == 0, mi, ml);
                        is_match = __builtin_amdgcn_readfirstlane((int)hit) != 0;
                        mlen = __builtin_amdgcn_readfirstlane(ml);
                        midx = __builtin_amdgcn_readfirstlane(mi);
                    }
                    if (lane == s1) ts_key[tsk] = (uint16_t)(j + 1);
                }
                acc |= 1ull << s1;
                uint32_t word;
                if (is_match) {                      // src/libzling_lz.cpp:160-167
                    word = (uint32_t)(258 + mlen - kMatchMin) | (uint32_t)midx << 16;
                    opos += 2; q += mlen; prevty = kTyMatch;
                } else {
                    const uint32_t cq = rl(ctx, s1), w = rl(cw, s1);
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
                    q = (int)ufl((uint32_t)q); opos = (int)ufl((uint32_t)opos); nt = ufl(nt); prevty = ufl(prevty);
                    seg = (unsigned long long)ufl((uint32_t)(seg >> 32)) << 32 | ufl((uint32_t)seg);
                    int s = q - P;
                    if ((seg >> s) & 1ull) seg &= ~((1ull << s) - 1ull);
                    else { seg = 0; while (s < 64) { seg |= rl64(hop_mask, s); s = (int)rl(hop_next, s); } }
                    // ---- validate every lane against (acc | seg) and against what changed since the evaluation
                    const unsigned long long all = acc | seg;
                    const uint32_t k = (uint32_t)__popcll(ctxmask & all & below);
                    const unsigned long long kq = canm ? (keymask & all & below) : 0ull;   // accepted earlier starts in my hash slot
                    const bool ring = canm && dmin <= k + kb;
                    // a probe's node is gone if its bucket has handed out more than `ld` slots since the evaluator read the
                    // bucket's head -- before the window, or inside it up to and including my own insert
                    const bool lring = (use1 && ld1 < kbl1 + (uint32_t)__popcll(lcm1 & all & beloweq)) ||
                                       (use2 && ld2 < kbl2 + (uint32_t)__popcll(lcm2 & all & beloweq));
                    bool fixed_ok = false;
                    int pfix = 0;
                    node0w = node0; mnode = (sp >> kSpNodeShift) & (kRing - 1);
                    const bool inseg = (seg & lane_bit) != 0;
                    if (level0 && __any(inseg && kq != 0 && !ring && !xkey)) {
                        // same-slot conflict inside the window, resolved in registers (see k_rolz_parse_wave)
                        const bool cand = inseg && kq != 0 && !ring && !xkey;
                        const int p = kq ? top_bit(kq) : 0;
                        const unsigned long long kq2 = kq & ~(1ull << p);
                        const bool has2 = kq2 != 0;
                        const int p2 = has2 ? top_bit(kq2) : 0;
                        const uint32_t key = ctx << 13 | hc;
                        const uint32_t key_p = (uint32_t)__shfl((int)key, p), chk_p = (uint32_t)__shfl((int)chk, p);
                        const Quad qp = {(uint32_t)__shfl((int)qa.a, p), (uint32_t)__shfl((int)qa.b, p),
                                         (uint32_t)__shfl((int)qa.c, p), (uint32_t)__shfl((int)qa.d, p)};
                        uint32_t key_p2 = key, chk_p2 = 0;
                        Quad qp2 = {0, 0, 0, 0};
                        if (__any(cand && has2)) {
                            key_p2 = (uint32_t)__shfl((int)key, p2); chk_p2 = (uint32_t)__shfl((int)chk, p2);
                            qp2 = Quad{(uint32_t)__shfl((int)qa.a, p2), (uint32_t)__shfl((int)qa.b, p2),
                                       (uint32_t)__shfl((int)qa.c, p2), (uint32_t)__shfl((int)qa.d, p2)};
                        }
                        const bool fixable = cand && key_p == key && (!has2 || key_p2 == key);
                        const bool cp = fixable && chk_p == chk;
                        uint32_t rp = cp ? lcp16(qa, qp) : 0u;
                        const bool lp = cp && rp == 16u;
                        const bool c2 = fixable && has2 && chk_p2 == chk;
                        uint32_t r2l = c2 ? lcp16(qa, qp2) : 0u;
                        const bool l2 = c2 && r2l == 16u;
                        if (__any(lp || l2)) {
                            uint32_t t0, t1;
                            lcp_tail2(buf + pos, buf + P + p, buf + P + p2, lp, l2, t0, t1);
                            rp = lp ? t0 : rp; r2l = l2 ? t1 : r2l;
                        }
                        const uint32_t slot_p = (hbase + (uint32_t)__popcll(ctxmask & all & ((1ull << p) - 1ull)) + 1u) & (kRing - 1);
                        const uint32_t slot_p2 = (hbase + (uint32_t)__popcll(ctxmask & all & ((1ull << p2) - 1ull)) + 1u) & (kRing - 1);
                        const bool second = has2 || node0 != 65535u;
                        const uint32_t rs = has2 ? r2l : len0, ns = has2 ? slot_p2 : node0;
                        uint32_t ml = kMatchMin - 1, mn = 0;
                        if (rp > ml) { ml = rp; mn = slot_p; }
                        if (ml != (uint32_t)kMatchMax && second && rs > ml) { ml = rs; mn = ns; }
                        const bool lzn = ml >= (uint32_t)kMatchMin && ml < (uint32_t)kLazyLimit;
                        const bool lclean = ((lkey & all & beloweq) == 0 && !xl && !lring) || !lzn;
                        bool veto = (sp & kSpVeto1) != 0;
                        const bool reprobe = fixable && lzn && lclean && ml != spec_len;
                        if (__any(reprobe)) {
                            const uint32_t mm = reprobe ? ml - 3u : 0u;
                            const uint32_t pr = ld32u(buf + pos + 1 + mm);
                            const uint32_t sr = ld32u(buf + (reprobe ? (lsrc1 & 0xFFFFFF) + mm : (uint32_t)pos));
                            if (reprobe) veto = (lsrc1 >> 31) != 0 && pr == sr;
                        }
                        const bool nmatch = ml >= (uint32_t)kMatchMin && !(lzn && veto);
                        fixed_ok = fixable && lclean && nmatch == spec_match && (!nmatch || ml == spec_len);
                        if (fixed_ok) { node0w = slot_p; mnode = mn; pfix = p; }
                    }
                    const bool dirty = canm && (xkey || ((kq != 0 || ring) && !fixed_ok));
                    const bool ldirty = !fixed_ok && spec_len >= (uint32_t)kMatchMin && spec_len < (uint32_t)kLazyLimit &&
                                        ((lkey & all & beloweq) != 0 || xl || lring);
                    const uint32_t m0 = mru[ctx];
                    const bool poss = !spec_match && pos + 1 < ilen &&
                                      ((m0 & 0xFFFF) == cw || (m0 >> 16) == cw || (hitmask & all & beloweq) != 0);
                    const unsigned long long prob = seg & __ballot(dirty || ldirty || poss);
                    const int f = prob ? (int)__builtin_ctzll(prob) : 64;
                    const unsigned long long com = f >= 64 ? seg : (seg & ((1ull << f) - 1ull));
                    if (com) {
                        const bool mine = (com & lane_bit) != 0;
                        // ---- MRU events of the committed boundaries (lane-parallel 2-slot push rules)
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
                        // ---- dictionary inserts (src/libzling_lz.cpp:227-230) and token words
                        const uint32_t head = (hbase + k + 1u) & (kRing - 1);
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
                                Bucket B(dict, ctx);
                                B.suffix[head] = (uint16_t)node0w;
                                B.offset[head] = (uint32_t)pos | chk << 24;
                                if (head_writer) B.hash[hc] = (uint16_t)head;
                                ts_key[tsk] = (uint16_t)(j + 1);
                            }
                            if (spec_match) word = (258u + spec_len - kMatchMin) | ((head - mnode) & (kRing - 1)) << 16;
                            else word = b_0 | ctx << 16;
                            __builtin_nontemporal_store(word, &tok[nt + (uint32_t)__popcll(com & below)]);
                        }
                        const int lastl = top_bit(com);
                        const bool last_match = ((match_lanes >> lastl) & 1ull) != 0;
                        nt += (uint32_t)__popcll(com);
                        opos += __popcll(com) + __popcll(com & match_lanes);
                        acc |= com;
                        q = P + lastl + (int)rl(tlen, lastl);
                        prevty = last_match ? kTyMatch : kTyLit;
                    }
                    if (f < 64) {
                        // the first problem token: a conflict (inside the window or with the windows committed since the
                        // evaluation) is replayed exactly against the current dictionary; a possible word hit is decided
                        // by the serial MRU code with the validated speculation
                        const bool conflict = rl((dirty || ldirty) ? 1u : 0u, f) != 0;
                        if (prof) { if (conflict) { n_redo++; if (rl((xkey || xl || (canm && dmin <= k + kb && dmin > k)) ? 1u : 0u, f)) n_stale++; } else n_poss++; }
                        serial_token(!conflict);
                    }
                }
            }
            // ---------------- publish: stores, then `done`, then the ring heads (see the header comment)
            const bool upd = canm && (acc & lane_bit) != 0;
            const uint32_t nh = (hbase + (uint32_t)__popcll(ctxmask & acc)) & (kRing - 1);
            // (published at the start of the next round, after that round's release of `pub`)
            pend_upd = upd; pend_ctx = ctx; pend_nh = nh;
            if (prof) c_p2 += __builtin_readcyclecounter() - t1;
        }
        if (nsub < kMaxSub && lane == 0) cuts[nsub] = SubCut{tok_begin, nt, (uint32_t)q, (uint32_t)opos};
        nsub++;
    }
    if (lane == 0) {
        WG_STORE_RLX(&fin, 1);
        if (overflow) { *a.overflow = 1; nsub = 0; nt = 0; }
        a.nsub[blk] = (uint32_t)nsub; a.ntok[blk] = nt;
    }
    if (prof && lane == 0) {
        unsigned long long* d = a.dbg + (size_t)blk * kDbgSlots;
        d[0] = c_wait; d[1] = 0; d[2] = c_p2; d[3] = n_round; d[4] = nt; d[5] = n_seg; d[6] = n_redo; d[7] = n_poss; d[17] = n_stale;
    }
}

void launch_rolz_parse_pipe(const ParseArgs& a, uint32_t nblocks_all, hipStream_t s, bool all_level0) {
    const uint32_t nblocks = nblocks_all - a.blk0;
    const int waves = 2 + (a.pf_waves > 0 ? 1 : 0);
    if (all_level0) hipLaunchKernelGGL((k_rolz_parse_pipe<true>), dim3(nblocks), dim3(64 * waves), 0, s, a);
    else hipLaunchKernelGGL((k_rolz_parse_pipe<false>), dim3(nblocks), dim3(64 * waves), 0, s, a);
}

}  // namespace zlng
