// zlng_api.hip -- the C-ABI of include/zlng.h: context, HBM pools, stage orchestration.
//
// Host-side driver of the encode pipeline for a RANGE of 16 MiB blocks per call:
//   K0 dict reset -> K1 parse (per block) -> K2 rank (per context, stream order)
//   -> K3 histogram -> K4 lengths/codes (per sub-block) -> K5 layout scan -> K6 pack+frame
// replacing one iteration range of the reference's outer loop (src/libzling.cpp:187-284).
// There is no CPU code path: every stage is a gfx950 kernel and every entry point fails with
// ZLNG_E_DEVICE when no device / kernel image is available.
#include <hip/hip_runtime.h>

#include <cstdio>
#include <cstdlib>
#include <algorithm>
#include <cstring>
#include <new>
#include <thread>
#include <vector>

#include "../../include/zlng.h"
#include "zlng_common.h"
#include "zlng_kernels.h"

using namespace zlng;

namespace {

constexpr int kMaxStages = 12;

struct StageTimer {
    hipEvent_t ev[kMaxStages + 1];
    const char* name[kMaxStages];
    int n = 0;
    bool ok = false;
};

}  // namespace

struct zlng_ctx {
    int device = 0;
    int level = 0;
    int current_level = 0;        // src/libzling.cpp:185, carried across calls (H3)
    bool is_encode = true;
    uint32_t max_blocks = 0;
    hipStream_t stream = nullptr;
    hipError_t last_hip_error = hipSuccess;

    // HBM pools (sized for max_blocks)
    uint8_t*  d_in = nullptr;      size_t in_cap = 0;     // staging for the host-input entry point
    uint8_t*  d_out = nullptr;     size_t out_cap = 0;    // staging for the host-output entry point
    uint8_t*  d_dict = nullptr;
    uint32_t* d_tok = nullptr;
    SubCut*   d_cuts = nullptr;
    uint32_t* d_nsub = nullptr;
    uint32_t* d_ntok = nullptr;
    uint8_t*  d_sched = nullptr;
    uint32_t* d_freq = nullptr;
    uint8_t*  d_lens = nullptr;
    uint16_t* d_codes = nullptr;
    uint32_t* d_olen = nullptr;
    uint64_t* d_sub_off = nullptr;
    uint64_t* d_blk_end = nullptr;
    uint64_t* d_summary = nullptr;
    uint8_t*  d_mtf = nullptr;        // live MTF tables (65,536 B)
    uint8_t*  d_mtf_snap = nullptr;   // tables at the start of every rank group of the current call: [0] = call entry (restored
                                      // when the call fails), [g] = where a level-schedule repair in group g restarts
    uint32_t  tok_cap = kTokCapDefault;   // token words per block the pools are sized for (grows once to kTokCapMax)
    bool      tokens_ranked = false;      // the pending parse's literals were already ranked in place by a failed finish
    uint32_t  last_blocks = 0;            // blocks the last decode call produced
    int       last_passes = 0;            // parse passes the last encode call needed (1 + level-schedule repairs + pool growth)
    unsigned long long* d_dbg = nullptr;  // parser phase counters (ZLNG_PROFILE=1)
    uint32_t* d_tile_base = nullptr;
    uint32_t* d_tile_hist = nullptr;
    uint32_t* d_ctx_total = nullptr;
    uint32_t* d_ctx_off = nullptr;
    uint8_t*  d_lit_byte = nullptr;
    uint8_t*  d_snap = nullptr;       // rank stage: table snapshots per 64-literal tile
    uint8_t*  d_tile_kk = nullptr;
    // Measured ALTERNATIVE, off by default (ZLNG_HOST_RANK_CONTEXTS=k): the k longest rank chains of a
    // call are walked by host threads instead of by k_mtf_dense, overlapped with the device's other chains.  The product
    // path is all-device; bench.py reports this mode as a separate, labelled line and never as `value`.
    int       host_rank_contexts = 0;
    uint8_t*  d_skip = nullptr;       // [256] contexts left to the host
    uint8_t*  h_pinned = nullptr;     // page-locked staging for their literal runs
    size_t    h_pinned_cap = 0;
    hipStream_t stream2 = nullptr;
    hipEvent_t  ev2 = nullptr;
    hipEvent_t  ev_parsed = nullptr;  // recorded behind every parse queued by zlng_encode_parse[_device] (zlng_encode_parse_after waits for another context's)
    // decode pools
    DecSub*   d_subs = nullptr;
    DecBlock* d_blocks = nullptr;
    uint32_t* d_sub_ntok = nullptr;
    uint32_t* d_ring = nullptr;
    std::vector<DecBlock> h_blocks;

    // host mirrors
    std::vector<uint8_t>  h_sched;
    std::vector<uint32_t> h_nsub, h_olen;
    std::vector<SubCut>   h_cuts;
    std::vector<uint64_t> h_blk_end;

    // state of a parse awaiting its finish (split entry points)
    const uint8_t* pending_in = nullptr;
    size_t pending_len = 0;
    uint32_t pending_blocks = 0;
    size_t staged_len = 0;            // bytes zlng_encode_finish_staged left in d_out for zlng_encode_copy_out

    StageTimer timer;
    int parser_kind = 3;              // 3 = workgroup-wide window parser (rolz_wg.hip; default), 1 = serial cross-check form (ZLNG_PARSER=serial)
};

namespace {

const uint8_t k_mtfinit[256] = {   // src/tables/gen.py:33-48
     32, 101, 116,  97, 105, 111, 110, 114, 115, 108, 104, 100,  99, 117,  93,  91,
    109, 112, 103, 102,  10, 121,  98,  39, 119,  46,  44, 118,  59,  38, 124,  47,
     49, 107,  61,  48,  67,  65,  58,  45,  84,  83,  60,  62,  50, 113,  73,  57,
     42, 120,  41,  40,  66,  77,  80,  69,  68,  53,  51,  72,  70,  56,  52,  71,
     82,  54,  76,  55,  78,  87, 122, 125, 123,  79, 106,  85,  74,  75, 208,  95,
    195,  35,  86, 215,  90,  34,  89, 209, 128, 224, 184, 131,  92, 227,  37,  33,
    176, 169, 206, 226, 130,  63,  88,  81, 161, 153,  43, 129, 188, 179, 216, 164,
    181, 189, 148, 190, 173, 187, 186, 229, 225, 167, 217, 177, 178, 168, 149, 185,
    197, 144, 147, 196, 207, 194, 180, 156, 132, 170, 166, 136, 182, 191,   9, 230,
    141, 160, 175,  36, 152, 140, 165, 145,  94, 133, 163, 183, 171, 157, 137, 174,
    134, 135, 236, 151, 231, 155, 201, 158, 138, 143, 150, 162, 159, 139, 172, 154,
    126, 232, 235, 146, 233, 228, 202, 203, 142, 214, 237, 204, 219, 234, 213,  96,
    218, 199,  64, 210, 239, 198, 211, 205, 212, 240, 222, 220, 200,   0,   1,   2,
      3,   4,   5,   6,   7,   8,  11,  12,  13,  14,  15,  16,  17,  18,  19,  20,
     21,  22,  23,  24,  25,  26,  27,  28,  29,  30,  31, 127, 192, 193, 221, 223,
    238, 241, 242, 243, 244, 245, 246, 247, 248, 249, 250, 251, 252, 253, 254, 255,
};

#define CTX_HIP(expr)                                                                      \
    do { hipError_t e_ = (expr); if (e_ != hipSuccess) { c->last_hip_error = e_; return ZLNG_E_DEVICE; } } while (0)

template <typename T>
int dev_alloc(zlng_ctx* c, T** p, size_t count) {
    void* v = nullptr;
    hipError_t e = hipMalloc(&v, count * sizeof(T));
    if (e != hipSuccess) { c->last_hip_error = e; return e == hipErrorOutOfMemory ? ZLNG_E_NOMEM : ZLNG_E_DEVICE; }
    *p = static_cast<T*>(v);
    return ZLNG_OK;
}

void timer_begin(zlng_ctx* c) {
    c->timer.n = 0;
    c->timer.ok = true;
    hipEventRecord(c->timer.ev[0], c->stream);
}
void timer_mark(zlng_ctx* c, const char* name) {
    if (c->timer.n >= kMaxStages) return;
    c->timer.name[c->timer.n] = name;
    c->timer.n++;
    hipEventRecord(c->timer.ev[c->timer.n], c->stream);
}

uint32_t blocks_of(size_t n) { return (uint32_t)((n + kBlockIn - 1) / kBlockIn); }

int ensure_in(zlng_ctx* c, size_t bytes) {
    if (c->in_cap >= bytes) return ZLNG_OK;
    if (c->d_in) hipFree(c->d_in);
    c->d_in = nullptr; c->in_cap = 0;
    int rc = dev_alloc(c, &c->d_in, bytes);
    if (rc == ZLNG_OK) c->in_cap = bytes;
    return rc;
}
int ensure_out(zlng_ctx* c, size_t bytes) {
    if (c->out_cap >= bytes) return ZLNG_OK;
    if (c->d_out) hipFree(c->d_out);
    c->d_out = nullptr; c->out_cap = 0;
    int rc = dev_alloc(c, &c->d_out, bytes);
    if (rc == ZLNG_OK) c->out_cap = bytes;
    return rc;
}

constexpr uint32_t kRankGroup = 8;        // blocks per rank group at levels 1-4 (granularity of a level-schedule repair)

uint32_t rank_group_blocks(const zlng_ctx* c, uint32_t nb) { return c->level == 0 ? nb : kRankGroup; }
uint32_t* overflow_flag(zlng_ctx* c) { return reinterpret_cast<uint32_t*>(c->d_summary + 4); }

// (Re)allocate the pools whose size follows the token capacity per block.
int alloc_token_pools(zlng_ctx* c, uint32_t tok_cap) {
    for (void* p : {(void*)c->d_tok, (void*)c->d_tile_hist, (void*)c->d_lit_byte, (void*)c->d_snap, (void*)c->d_tile_kk}) if (p) hipFree(p);
    c->d_tok = nullptr; c->d_tile_hist = nullptr; c->d_lit_byte = nullptr; c->d_snap = nullptr; c->d_tile_kk = nullptr;
    const size_t nb = c->max_blocks;
    int rc;
    if ((rc = dev_alloc(c, &c->d_tok, nb * tok_cap)) || (rc = dev_alloc(c, &c->d_tile_hist, nb * (tok_cap / 4096) * 256)) ||
        (rc = dev_alloc(c, &c->d_lit_byte, nb * tok_cap + 256 * 64 + 128)) ||    // + run alignment + one tile of read-ahead
        (rc = dev_alloc(c, &c->d_snap, 4 * (nb * tok_cap + 256 * 64) + 256)) ||   // one 256-byte table per tile of 64 literals
        (rc = dev_alloc(c, &c->d_tile_kk, (nb * tok_cap + 256 * 64) / 64 + 8)))
        return rc;
    c->tok_cap = tok_cap;
    return ZLNG_OK;
}

// Reset + parse of blocks [blk0, nb) under the schedule in d_sched.
void run_front(zlng_ctx* c, const uint8_t* d_in, size_t in_len, uint32_t nb, uint32_t blk0) {
    static const int min_restart = getenv("ZLNG_MIN_RESTART") ? atoi(getenv("ZLNG_MIN_RESTART")) : 12;
    static const int prefix_pct = getenv("ZLNG_PREFIX_PCT") ? atoi(getenv("ZLNG_PREFIX_PCT")) : 0;      // measured within +-1 % of "always iterate" on every workload: off
    static const int ring_fix = getenv("ZLNG_RING_FIX") ? atoi(getenv("ZLNG_RING_FIX")) : 0;           // round 5: exact in the CPU model, not yet measured on a GPU: off
    ParseArgs pa{d_in, in_len, c->d_dict, c->d_tok, c->d_cuts, c->d_nsub, c->d_ntok, c->d_sched, c->d_dbg, min_restart, prefix_pct, c->tok_cap, blk0, overflow_flag(c), ring_fix};
    static const int wg_waves = [] {
        const int v = getenv("ZLNG_WG_WAVES") ? atoi(getenv("ZLNG_WG_WAVES")) : 4;
        if (v != 2 && v != 4 && v != 8) fprintf(stderr, "zlng: ZLNG_WG_WAVES=%d is not 2, 4 or 8: using %d\n", v, v <= 2 ? 2 : (v <= 4 ? 4 : 8));
        return v;
    }();
    static const bool wg_wide = !(getenv("ZLNG_WG_COMPACT") && atoi(getenv("ZLNG_WG_COMPACT")) != 0);   // slot plane of the wg parser at level 0 (A/B switch)
    static const bool wg_hot = getenv("ZLNG_WG_HOT") && atoi(getenv("ZLNG_WG_HOT")) != 0;     // LDS mirror of the hottest bucket (A/B switch; north_star's "LDS-staged buckets")
    const bool wide = c->level == 0 && c->parser_kind == 3 && wg_wide;
    launch_dict_reset(c->d_dict + (size_t)blk0 * kDictBytes, nb - blk0, c->stream, wide);
    timer_mark(c, "dict_reset");
    if (c->parser_kind == 1) launch_rolz_parse_serial(pa, nb, c->stream);
    else launch_rolz_parse_wg(pa, nb, c->stream, c->level == 0, wg_waves, wide, wg_hot);
    timer_mark(c, "rolz_parse");
}

HuffArgs huff_args(zlng_ctx* c, uint32_t nb, uint32_t blk0, uint8_t* d_out, size_t out_cap) {
    return HuffArgs{c->d_tok, c->d_cuts, c->d_nsub, nb, c->tok_cap, blk0, c->d_freq, c->d_lens, c->d_codes, c->d_olen,
                    c->d_sub_off, c->d_blk_end, c->d_summary, overflow_flag(c), d_out, (uint64_t)out_cap};
}

// ---- measured alternative: the longest chains on host cores (ZLNG_HOST_RANK_CONTEXTS, off by default) -------------------
// ZlingMTFEncoder::Encode (src/libzling_lz.cpp:112-117) on one context's dense literal run, ranks in place.
void mtf_chain_host(uint8_t table[256], uint8_t* lits, size_t n) {
    uint8_t index[256], nxt[256];
    for (int i = 0; i < 256; i++) { index[table[i]] = (uint8_t)i; nxt[i] = (uint8_t)mtf_next((uint32_t)i); }
    for (size_t j = 0; j < n; j++) {
        const uint8_t ch = lits[j], i = index[ch], nx = nxt[i], d = table[nx];
        table[nx] = ch; table[i] = d; index[ch] = nx; index[d] = i;
        lits[j] = i;
    }
}

int rank_longest_chains_on_host(zlng_ctx* c, MtfArgs& ma, bool single) {
    uint32_t total[256], off[256];
    CTX_HIP(hipMemcpyAsync(total, c->d_ctx_total, sizeof total, hipMemcpyDeviceToHost, c->stream));
    CTX_HIP(hipMemcpyAsync(off, c->d_ctx_off, sizeof off, hipMemcpyDeviceToHost, c->stream));
    CTX_HIP(hipStreamSynchronize(c->stream));
    // the k longest chains, if they are long enough to be worth two PCIe copies
    std::vector<int> pick;
    uint8_t skip[256] = {0};
    for (int k = 0; k < c->host_rank_contexts; k++) {
        int best = -1;
        for (int x = 0; x < 256; x++) if (!skip[x] && total[x] >= (1u << 18) && (best < 0 || total[x] > total[best])) best = x;
        if (best < 0) break;
        skip[best] = 1;
        pick.push_back(best);
    }
    size_t need = 0;
    for (int x : pick) need += ((size_t)total[x] + 63) & ~(size_t)63;
    if (need > c->h_pinned_cap) {
        if (c->h_pinned) hipHostFree(c->h_pinned);
        c->h_pinned = nullptr; c->h_pinned_cap = 0;
        void* p = nullptr;
        if (hipHostMalloc(&p, need + (need >> 2) + 4096, hipHostMallocDefault) != hipSuccess) return ZLNG_E_NOMEM;
        c->h_pinned = static_cast<uint8_t*>(p); c->h_pinned_cap = need + (need >> 2) + 4096;
    }
    CTX_HIP(hipMemcpyAsync(c->d_skip, skip, 256, hipMemcpyHostToDevice, c->stream));
    CTX_HIP(hipEventRecord(c->ev2, c->stream));                       // partition done, skip list in place
    ma.skip = c->d_skip;
    launch_mtf_chain(ma, c->stream);                                  // the device walks every other chain meanwhile
    if (single) timer_mark(c, "mtf_chain");
    // second stream: runs out, host chains, ranks and tables back
    CTX_HIP(hipStreamWaitEvent(c->stream2, c->ev2, 0));
    std::vector<uint8_t> tables(pick.size() * 256);
    size_t pos = 0;
    std::vector<size_t> at(pick.size());
    for (size_t k = 0; k < pick.size(); k++) {
        at[k] = pos;
        CTX_HIP(hipMemcpyAsync(c->h_pinned + pos, c->d_lit_byte + off[pick[k]], total[pick[k]], hipMemcpyDeviceToHost, c->stream2));
        CTX_HIP(hipMemcpyAsync(tables.data() + 256 * k, c->d_mtf + 256 * pick[k], 256, hipMemcpyDeviceToHost, c->stream2));
        pos += ((size_t)total[pick[k]] + 63) & ~(size_t)63;
    }
    CTX_HIP(hipStreamSynchronize(c->stream2));
    std::vector<std::thread> th;
    for (size_t k = 0; k < pick.size(); k++)
        th.emplace_back(mtf_chain_host, tables.data() + 256 * k, c->h_pinned + at[k], (size_t)total[pick[k]]);
    for (auto& t : th) t.join();
    for (size_t k = 0; k < pick.size(); k++) {
        CTX_HIP(hipMemcpyAsync(c->d_lit_byte + off[pick[k]], c->h_pinned + at[k], total[pick[k]], hipMemcpyHostToDevice, c->stream2));
        CTX_HIP(hipMemcpyAsync(c->d_mtf + 256 * pick[k], tables.data() + 256 * k, 256, hipMemcpyHostToDevice, c->stream2));
    }
    CTX_HIP(hipStreamSynchronize(c->stream2));                        // (tables is a local: the copies must have read it)
    return ZLNG_OK;                                                   // the caller's next launch on c->stream follows k_mtf_dense
}

// Rank (group by group from group g0, whose entry tables are in d_mtf), histogram and lengths of blocks [g0 * G, nb).
int run_back(zlng_ctx* c, uint32_t nb, uint32_t g0, uint8_t* d_out, size_t out_cap) {
    const uint32_t G = rank_group_blocks(c, nb);
    static const uint32_t chain_prio = getenv("ZLNG_CHAIN_PRIO") ? (uint32_t)atoi(getenv("ZLNG_CHAIN_PRIO")) : 1u;
    for (uint32_t b = g0 * G, g = g0; b < nb; b += G, g++) {
        const uint32_t n = std::min(G, nb - b);
        MtfArgs ma{c->d_tok + (size_t)b * c->tok_cap, c->d_ntok + b, n, c->tok_cap, c->d_mtf, c->d_tile_base, c->d_tile_hist,
                   c->d_ctx_total, c->d_ctx_off, c->d_lit_byte, c->d_snap, c->d_tile_kk, nullptr, (c->d_dbg && c->max_blocks >= 22) ? c->d_dbg : nullptr, chain_prio};
        const bool single = G >= nb;                 // one group (always at level 0): time the serial chain by itself
        launch_lit_partition(ma, c->stream);
        if (single) timer_mark(c, "lit_partition");
        if (c->host_rank_contexts > 0) {
            const int rc = rank_longest_chains_on_host(c, ma, single);
            if (rc != ZLNG_OK) return rc;
        } else {
            launch_mtf_chain(ma, c->stream);
            if (single) timer_mark(c, "mtf_chain");
        }
        launch_mtf_finish(ma, c->stream);
        if (b + G < nb)        // tables at the start of the next group
            CTX_HIP(hipMemcpyAsync(c->d_mtf_snap + (size_t)(g + 1) * ZLNG_MTF_STATE, c->d_mtf, ZLNG_MTF_STATE, hipMemcpyDeviceToDevice, c->stream));
    }
    timer_mark(c, G >= nb ? "rank_replay" : "mtf_rank");
    const HuffArgs ha = huff_args(c, nb, g0 * G, d_out, out_cap);
    launch_histogram(ha, c->stream);
    timer_mark(c, "histogram");
    launch_lengths(ha, c->stream);
    timer_mark(c, "huff_lengths");
    return ZLNG_OK;
}

// Host check of the level schedule against what the reference would have used (src/libzling.cpp:261-266).  Returns true if
// consistent.  Otherwise *bad_blk names the block of the first disagreement and h_sched is rewritten from there on: the
// offending sub-block gets the level the reference would have used, and every later one a new speculation derived from the
// payload ratios measured in this pass (the decision a sub-block's ratio implies is a property of the data around it, so
// the same pass usually predicts all further flips of the range at once -- the number of passes follows the number of
// mispredictions, not the number of flips).  *final_level gets current_level after the range.
bool verify_schedule(zlng_ctx* c, uint32_t nb, int entry_level, int* final_level, uint32_t* bad_blk) {
    int cur = entry_level;
    for (uint32_t b = 0; b < nb; b++) {
        uint32_t old = 0;
        for (uint32_t s = 0; s < c->h_nsub[b] && s < (uint32_t)kMaxSub; s++) {
            const size_t si = (size_t)b * kMaxSub + s;
            if (c->h_sched[si] != (uint8_t)cur) {
                *bad_blk = b;
                int guess = cur;
                for (uint32_t b2 = b; b2 < nb; b2++) {
                    uint32_t o2 = 0;
                    for (uint32_t s2 = 0; s2 < (uint32_t)kMaxSub; s2++) {
                        const size_t k = (size_t)b2 * kMaxSub + s2;
                        const bool known = s2 < c->h_nsub[b2];
                        if (b2 > b || s2 >= s) {
                            c->h_sched[k] = (uint8_t)guess;
                            if (known) guess = (double)c->h_olen[k] / (double)(c->h_cuts[k].encpos - o2 + 1) > 0.95 ? 0 : c->level;
                        }
                        if (known) o2 = c->h_cuts[k].encpos;
                    }
                }
                return false;
            }
            const uint32_t enc = c->h_cuts[si].encpos;
            const double ratio = 1.0 * (double)c->h_olen[si] / (double)(enc - old + 1);
            cur = ratio > 0.95 ? 0 : c->level;
            old = enc;
        }
    }
    *final_level = cur;
    return true;
}

// A failed call must leave the stream state as it found it (the caller may retry with a larger buffer).
int fail_restore(zlng_ctx* c, int rc) {
    hipMemcpyAsync(c->d_mtf, c->d_mtf_snap, ZLNG_MTF_STATE, hipMemcpyDeviceToDevice, c->stream);
    hipStreamSynchronize(c->stream);
    return rc;
}

int encode_device_impl(zlng_ctx* c, const uint8_t* d_in, size_t in_len, uint8_t* d_out, size_t out_cap,
                       size_t* out_len, size_t* per_block_out_end, bool do_parse) {
    if (!c || !c->is_encode || !out_len) return ZLNG_E_ARG;
    *out_len = 0;
    if (in_len == 0) return ZLNG_OK;
    if (!d_in || !d_out || ((uintptr_t)d_out & 3)) return ZLNG_E_ARG;
    const uint32_t nb = blocks_of(in_len);
    if (nb > c->max_blocks) return ZLNG_E_ARG;
    if (!c->d_tok || !c->d_lit_byte || !c->d_snap) return ZLNG_E_NOMEM;      // an earlier pool growth ran out of HBM
    CTX_HIP(hipSetDevice(c->device));

    const int entry_level = c->current_level;
    const size_t nsubs = (size_t)nb * kMaxSub;
    const uint32_t G = rank_group_blocks(c, nb);
    if (c->tokens_ranked) do_parse = true;            // a failed finish ranked the pending tokens in place: parse again
    CTX_HIP(hipMemcpyAsync(c->d_mtf_snap, c->d_mtf, ZLNG_MTF_STATE, hipMemcpyDeviceToDevice, c->stream));
    // from here on a failed call puts the tables back (zlng.h: a failed call leaves the stream state as it found it)
#define ENC_HIP(expr) do { hipError_t e_ = (expr); if (e_ != hipSuccess) { c->last_hip_error = e_; return fail_restore(c, ZLNG_E_DEVICE); } } while (0)

    c->last_passes = 0;
    // a finish that reuses a parse queued earlier: what lies between the end of that parse and this call (the contexts before it
    // in the range being ranked) is NOT part of this context's rank stage -- it gets a stage of its own
    if (!do_parse) timer_mark(c, "idle_before_finish");
    for (bool grown = false;; grown = true) {         // second turn only after a token-pool overflow
        if (do_parse) {
            timer_begin(c);
            // speculation: the requested level everywhere, except that a range entered at level 0 (the previous range ended
            // incompressible) is assumed to stay there
            for (size_t k = 0; k < nsubs; k++) c->h_sched[k] = (uint8_t)(entry_level != c->level ? entry_level : c->level);
            ENC_HIP(hipMemsetAsync(overflow_flag(c), 0, 8, c->stream));
        }
        int final_level = entry_level;
        uint32_t restart = 0;                          // first block to (re)parse; a multiple of G
        bool parse_now = do_parse, overflow = false;
        for (int attempt = 0;; attempt++) {
            if (parse_now) {
                ENC_HIP(hipMemcpyAsync(c->d_sched, c->h_sched.data(), nsubs, hipMemcpyHostToDevice, c->stream));
                run_front(c, d_in, in_len, nb, restart);
                c->last_passes++;
            }
            if (attempt > 0)
                ENC_HIP(hipMemcpyAsync(c->d_mtf, c->d_mtf_snap + (size_t)(restart / G) * ZLNG_MTF_STATE, ZLNG_MTF_STATE, hipMemcpyDeviceToDevice, c->stream));
            c->tokens_ranked = true;
            int rc = run_back(c, nb, restart / G, d_out, out_cap);
            if (rc != ZLNG_OK) return fail_restore(c, rc);
            if (c->level == 0) { final_level = 0; break; }        // level 0: adaptation is inert
            uint32_t of = 0;
            ENC_HIP(hipMemcpyAsync(&of, overflow_flag(c), 4, hipMemcpyDeviceToHost, c->stream));
            ENC_HIP(hipMemcpyAsync(c->h_nsub.data(), c->d_nsub, nb * sizeof(uint32_t), hipMemcpyDeviceToHost, c->stream));
            ENC_HIP(hipMemcpyAsync(c->h_cuts.data(), c->d_cuts, nsubs * sizeof(SubCut), hipMemcpyDeviceToHost, c->stream));
            ENC_HIP(hipMemcpyAsync(c->h_olen.data(), c->d_olen, nsubs * sizeof(uint32_t), hipMemcpyDeviceToHost, c->stream));
            if (hipStreamSynchronize(c->stream) != hipSuccess) return fail_restore(c, ZLNG_E_DEVICE);
            if (of >= 2) return fail_restore(c, ZLNG_E_DEVICE);      // the parser's own fault flag: growing the pools would not help
            if (of) { overflow = true; break; }
            uint32_t bad = 0;
            if (verify_schedule(c, nb, entry_level, &final_level, &bad)) break;
            if (attempt > 4 * kMaxSub * (int)nb) return fail_restore(c, ZLNG_E_DEVICE);   // cannot happen: each pass fixes >= 1 sub-block
            restart = bad / G * G;
            parse_now = true;
        }

        uint64_t summary[2] = {0, 0};
        if (!overflow) {
            const HuffArgs ha = huff_args(c, nb, 0, d_out, out_cap);
            launch_layout(ha, c->stream);
            timer_mark(c, "layout_scan");
            launch_pack(ha, c->stream);
            timer_mark(c, "huff_pack");
            hipMemcpyAsync(summary, c->d_summary, sizeof summary, hipMemcpyDeviceToHost, c->stream);
            hipMemcpyAsync(c->h_blk_end.data(), c->d_blk_end, nb * sizeof(uint64_t), hipMemcpyDeviceToHost, c->stream);
            if (hipStreamSynchronize(c->stream) != hipSuccess || hipGetLastError() != hipSuccess) return fail_restore(c, ZLNG_E_DEVICE);
            overflow = (summary[1] & 8) != 0;
        }
        if (overflow) {
            // some block ran out of token words: size the pools for the worst case (one token per byte) and repeat the call
            if (grown || c->tok_cap >= kTokCapMax) return fail_restore(c, ZLNG_E_DEVICE);
            hipMemcpyAsync(c->d_mtf, c->d_mtf_snap, ZLNG_MTF_STATE, hipMemcpyDeviceToDevice, c->stream);
            hipStreamSynchronize(c->stream);
            const int rc = alloc_token_pools(c, kTokCapMax);
            if (rc != ZLNG_OK) { c->tokens_ranked = false; c->pending_in = nullptr; return rc; }   // (tables restored above; the pools are gone: nothing stays pending)
            do_parse = true;
            continue;
        }
        if (summary[1] & 2) return fail_restore(c, ZLNG_E_DEVICE);
        if (summary[1] & 1) return fail_restore(c, ZLNG_E_PAYLOAD);
        if (summary[1] & 4) return fail_restore(c, ZLNG_E_CAP);
        *out_len = (size_t)summary[0];
        if (per_block_out_end) for (uint32_t b = 0; b < nb; b++) per_block_out_end[b] = (size_t)c->h_blk_end[b];
        c->current_level = final_level;
        c->tokens_ranked = false;
        return ZLNG_OK;
    }
}
#undef ENC_HIP

}  // namespace

extern "C" {

int zlng_device_count(void) {
    int n = 0;
    if (hipGetDeviceCount(&n) != hipSuccess) return 0;
    int ok = 0;
    for (int i = 0; i < n; i++) {
        hipDeviceProp_t p;
        if (hipGetDeviceProperties(&p, i) == hipSuccess && strncmp(p.gcnArchName, "gfx950", 6) == 0) ok++;
    }
    return ok;
}

zlng_ctx* zlng_create(int device, int level, int is_encode, int max_blocks, int* err) {
    int dummy;
    if (!err) err = &dummy;
    *err = ZLNG_OK;
    // 240 blocks (3.75 GiB) per call keeps every literal / token position inside 32 bits; larger streams are fed in several calls
    if (level < 0 || level > 4 || max_blocks <= 0 || max_blocks > 240) { *err = ZLNG_E_ARG; return nullptr; }
    int n = 0;
    if (hipGetDeviceCount(&n) != hipSuccess || device < 0 || device >= n) { *err = ZLNG_E_DEVICE; return nullptr; }
    hipDeviceProp_t prop;
    if (hipGetDeviceProperties(&prop, device) != hipSuccess || strncmp(prop.gcnArchName, "gfx950", 6) != 0) {
        *err = ZLNG_E_DEVICE;      // kernels are built for gfx950 only; there is no other path
        return nullptr;
    }
    zlng_ctx* c = new (std::nothrow) zlng_ctx();
    if (!c) { *err = ZLNG_E_NOMEM; return nullptr; }
    c->device = device;
    c->level = level;
    c->current_level = level;
    c->is_encode = is_encode != 0;
    c->max_blocks = (uint32_t)max_blocks;
    const char* pk = getenv("ZLNG_PARSER");
    c->parser_kind = (pk && strcmp(pk, "serial") == 0) ? 1 : 3;
    int rc = ZLNG_OK;
    auto fail = [&](int code) { *err = code; zlng_destroy(c); return (zlng_ctx*)nullptr; };
    if (hipSetDevice(device) != hipSuccess) return fail(ZLNG_E_DEVICE);
    if (hipStreamCreateWithFlags(&c->stream, hipStreamNonBlocking) != hipSuccess) return fail(ZLNG_E_DEVICE);
    for (int i = 0; i <= kMaxStages; i++) if (hipEventCreate(&c->timer.ev[i]) != hipSuccess) return fail(ZLNG_E_DEVICE);
    const size_t nb = (size_t)max_blocks, nsubs = nb * kMaxSub;
    if ((rc = dev_alloc(c, &c->d_mtf, ZLNG_MTF_STATE)) ||
        (rc = dev_alloc(c, &c->d_mtf_snap, ((size_t)max_blocks / kRankGroup + 2) * ZLNG_MTF_STATE))) return fail(rc);
    if (!c->is_encode) {
        const size_t max_subs = nb * kDecSubsPerBlock;
        if ((rc = dev_alloc(c, &c->d_subs, max_subs)) || (rc = dev_alloc(c, &c->d_blocks, nb)) ||
            (rc = dev_alloc(c, &c->d_sub_ntok, 2 * max_subs)) || (rc = dev_alloc(c, &c->d_ring, (size_t)256 * kRing)) ||
            (rc = dev_alloc(c, &c->d_tok, nb * ((size_t)kTokCapMax + 64))) || (rc = dev_alloc(c, &c->d_summary, 8)))
            return fail(rc);
        c->h_blocks.resize(nb);
    }
    if (c->is_encode) {
        const char* tc = getenv("ZLNG_TOK_CAP");      // token words per block the pools start with (testing aid; multiple of 4096)
        uint32_t tok_cap = tc ? (uint32_t)strtoul(tc, nullptr, 10) : kTokCapDefault;
        tok_cap = std::min(kTokCapMax, std::max(8192u, tok_cap / 4096u * 4096u));
        if ((rc = dev_alloc(c, &c->d_dict, nb * kDictBytes)) || (rc = alloc_token_pools(c, tok_cap)) ||
            (rc = dev_alloc(c, &c->d_cuts, nsubs)) || (rc = dev_alloc(c, &c->d_nsub, nb)) ||
            (rc = dev_alloc(c, &c->d_ntok, nb)) || (rc = dev_alloc(c, &c->d_sched, nsubs)) ||
            (rc = dev_alloc(c, &c->d_freq, nsubs * kNsymAll)) || (rc = dev_alloc(c, &c->d_lens, nsubs * kNsymAll)) ||
            (rc = dev_alloc(c, &c->d_codes, nsubs * kNsymAll)) || (rc = dev_alloc(c, &c->d_olen, nsubs)) ||
            (rc = dev_alloc(c, &c->d_sub_off, nsubs)) || (rc = dev_alloc(c, &c->d_blk_end, nb)) ||
            (rc = dev_alloc(c, &c->d_summary, 8)) || (rc = dev_alloc(c, &c->d_tile_base, nb + 1)) ||
            (rc = dev_alloc(c, &c->d_ctx_total, 256)) || (rc = dev_alloc(c, &c->d_ctx_off, 256)))
            return fail(rc);
        if (hipMemset(c->d_summary, 0, 8 * sizeof(uint64_t)) != hipSuccess) return fail(ZLNG_E_DEVICE);
        c->h_sched.resize(nsubs);
        c->h_nsub.resize(nb);
        c->h_olen.resize(nsubs);
        c->h_cuts.resize(nsubs);
        c->h_blk_end.resize(nb);
        const char* hr = getenv("ZLNG_HOST_RANK_CONTEXTS");
        if (hr && (rc = zlng_set_host_rank_contexts(c, atoi(hr))) != ZLNG_OK) return fail(rc);
        const char* pf = getenv("ZLNG_PROFILE");
        if (pf && pf[0] == '1') {
            if ((rc = dev_alloc(c, &c->d_dbg, nb * kDbgSlots))) return fail(rc);
            if (hipMemset(c->d_dbg, 0, nb * kDbgSlots * sizeof(unsigned long long)) != hipSuccess) return fail(ZLNG_E_DEVICE);
        }
    }
    uint8_t init[ZLNG_MTF_STATE];
    for (int ctx = 0; ctx < 256; ctx++) memcpy(init + 256 * ctx, k_mtfinit, 256);   // src/libzling_lz.cpp:106-111
    if (hipMemcpy(c->d_mtf, init, ZLNG_MTF_STATE, hipMemcpyHostToDevice) != hipSuccess) return fail(ZLNG_E_DEVICE);
    return c;
}

void zlng_destroy(zlng_ctx* c) {
    if (!c) return;
    hipSetDevice(c->device);
    if (c->stream) hipStreamSynchronize(c->stream);
    void* ptrs[] = {c->d_in, c->d_out, c->d_dict, c->d_tok, c->d_cuts, c->d_nsub, c->d_ntok, c->d_sched, c->d_freq,
                    c->d_lens, c->d_codes, c->d_olen, c->d_sub_off, c->d_blk_end, c->d_summary, c->d_mtf, c->d_mtf_snap, c->d_dbg, c->d_subs, c->d_blocks, c->d_sub_ntok, c->d_ring, c->d_tile_base, c->d_tile_hist,
                    c->d_ctx_total, c->d_ctx_off, c->d_lit_byte, c->d_snap, c->d_tile_kk, c->d_skip};
    for (void* p : ptrs) if (p) hipFree(p);
    if (c->h_pinned) hipHostFree(c->h_pinned);
    if (c->stream2) hipStreamDestroy(c->stream2);
    if (c->ev2) hipEventDestroy(c->ev2);
    if (c->ev_parsed) hipEventDestroy(c->ev_parsed);
    for (int i = 0; i <= kMaxStages; i++) if (c->timer.ev[i]) hipEventDestroy(c->timer.ev[i]);
    if (c->stream) hipStreamDestroy(c->stream);
    delete c;
}

size_t zlng_encode_bound(size_t n) {
    // every u16 entry covers >= 1 input byte, every sub-block but the last of a block has >= 262143 entries, and a
    // payload above 393,216 B is refused (ZLNG_E_PAYLOAD): <= 1.5 output bytes per input byte plus headers
    const size_t nblk = (n + kBlockIn - 1) / kBlockIn;
    const size_t nsub = n / 262143 + nblk + 1;
    return std::min(nsub * (size_t)(kHeaderBytes + kPayloadMax), 2 * n + nsub * (size_t)(kHeaderBytes + kTableBytes + 8)) + nblk + 64;
}

int zlng_encode_blocks_device(zlng_ctx* c, const void* d_in, size_t in_len, void* d_out, size_t out_cap,
                              size_t* out_len, size_t* per_block_out_end) {
    return encode_device_impl(c, static_cast<const uint8_t*>(d_in), in_len, static_cast<uint8_t*>(d_out), out_cap,
                              out_len, per_block_out_end, true);
}

int zlng_encode_blocks(zlng_ctx* c, const uint8_t* in, size_t in_len, uint8_t* out, size_t out_cap, size_t* out_len,
                       size_t* per_block_out_end) {
    if (!c || !c->is_encode || !out_len) return ZLNG_E_ARG;
    *out_len = 0;
    if (in_len == 0) return ZLNG_OK;
    if (!in || !out) return ZLNG_E_ARG;
    if (blocks_of(in_len) > c->max_blocks) return ZLNG_E_ARG;
    CTX_HIP(hipSetDevice(c->device));
    int rc;
    const size_t bound = zlng_encode_bound(in_len);
    if ((rc = ensure_in(c, in_len + 512)) || (rc = ensure_out(c, bound))) return rc;
    CTX_HIP(hipMemcpyAsync(c->d_in, in, in_len, hipMemcpyHostToDevice, c->stream));
    CTX_HIP(hipMemsetAsync(c->d_in + in_len, 0, 512, c->stream));
    size_t produced = 0;
    // the caller's capacity is checked on the device before anything is written or the stream state moves on
    rc = encode_device_impl(c, c->d_in, in_len, c->d_out, std::min(bound, out_cap), &produced, per_block_out_end, true);
    if (rc != ZLNG_OK) return rc;
    CTX_HIP(hipMemcpyAsync(out, c->d_out, produced, hipMemcpyDeviceToHost, c->stream));
    CTX_HIP(hipStreamSynchronize(c->stream));
    *out_len = produced;
    return ZLNG_OK;
}

int zlng_encode_parse_device(zlng_ctx* c, const void* d_in, size_t in_len) {
    if (!c || !c->is_encode || !d_in || in_len == 0) return ZLNG_E_ARG;
    const uint32_t nb = blocks_of(in_len);
    if (nb > c->max_blocks) return ZLNG_E_ARG;
    if (!c->d_tok || !c->d_lit_byte || !c->d_snap) return ZLNG_E_NOMEM;      // an earlier pool growth ran out of HBM
    CTX_HIP(hipSetDevice(c->device));
    const size_t nsubs = (size_t)nb * kMaxSub;
    timer_begin(c);
    // speculate the requested level everywhere; finish() re-parses if the incoming level disagrees
    for (size_t k = 0; k < nsubs; k++) c->h_sched[k] = (uint8_t)c->level;
    CTX_HIP(hipMemcpyAsync(c->d_sched, c->h_sched.data(), nsubs, hipMemcpyHostToDevice, c->stream));
    CTX_HIP(hipMemsetAsync(overflow_flag(c), 0, 8, c->stream));
    run_front(c, static_cast<const uint8_t*>(d_in), in_len, nb, 0);
    if (!c->ev_parsed) CTX_HIP(hipEventCreateWithFlags(&c->ev_parsed, hipEventDisableTiming));
    CTX_HIP(hipEventRecord(c->ev_parsed, c->stream));
    c->tokens_ranked = false;
    c->pending_in = static_cast<const uint8_t*>(d_in);
    c->pending_len = in_len;
    c->pending_blocks = nb;
    return ZLNG_OK;
}

// Everything queued on `c` from now on starts after the parse last queued on `first` has finished (same device).  A range that
// goes through several contexts uses it to keep only two parses in flight -- parses then END in stream order and the rank stage
// of context k runs beside the parse of context k + 1.
int zlng_encode_parse_after(zlng_ctx* c, zlng_ctx* first) {
    if (!c || !first || !c->is_encode || !first->is_encode || c->device != first->device) return ZLNG_E_ARG;
    if (!first->ev_parsed) return ZLNG_OK;          // nothing was ever queued there
    CTX_HIP(hipSetDevice(c->device));
    CTX_HIP(hipStreamWaitEvent(c->stream, first->ev_parsed, 0));
    return ZLNG_OK;
}

int zlng_encode_finish_device(zlng_ctx* c, void* d_out, size_t out_cap, size_t* out_len, size_t* per_block_out_end) {
    if (!c || !c->pending_in) return ZLNG_E_ARG;
    // a parse speculated at `level` is reusable iff the stream enters this range at `level`
    const bool reuse = (c->current_level == c->level);
    const int rc = encode_device_impl(c, c->pending_in, c->pending_len, static_cast<uint8_t*>(d_out), out_cap, out_len, per_block_out_end, !reuse);
    if (rc == ZLNG_OK) c->pending_in = nullptr;       // a failed finish keeps the range pending: it can be repeated (e.g. with more room)
    return rc;
}

int zlng_encode_parse(zlng_ctx* c, const uint8_t* in, size_t in_len) {
    if (!c || !c->is_encode || !in || in_len == 0) return ZLNG_E_ARG;
    if (blocks_of(in_len) > c->max_blocks) return ZLNG_E_ARG;
    CTX_HIP(hipSetDevice(c->device));
    int rc;
    if ((rc = ensure_in(c, in_len + 512)) || (rc = ensure_out(c, zlng_encode_bound(in_len)))) return rc;
    CTX_HIP(hipMemcpyAsync(c->d_in, in, in_len, hipMemcpyHostToDevice, c->stream));
    CTX_HIP(hipMemsetAsync(c->d_in + in_len, 0, 512, c->stream));
    return zlng_encode_parse_device(c, c->d_in, in_len);
}
int zlng_encode_finish(zlng_ctx* c, uint8_t* out, size_t out_cap, size_t* out_len, size_t* per_block_out_end) {
    if (!c || !c->pending_in || !out || !out_len) return ZLNG_E_ARG;
    *out_len = 0;
    size_t produced = 0;
    {   // the pending range may have come from zlng_encode_parse_device: the staging buffer is sized here, not assumed
        const int rc0 = ensure_out(c, zlng_encode_bound(c->pending_len));
        if (rc0 != ZLNG_OK) return rc0;
    }
    const int rc = zlng_encode_finish_device(c, c->d_out, std::min(zlng_encode_bound(c->pending_len), out_cap), &produced, per_block_out_end);
    if (rc != ZLNG_OK) return rc;
    CTX_HIP(hipMemcpyAsync(out, c->d_out, produced, hipMemcpyDeviceToHost, c->stream));
    CTX_HIP(hipStreamSynchronize(c->stream));
    *out_len = produced;
    return ZLNG_OK;
}

// The two halves of zlng_encode_finish, for a driver that finishes several contexts in stream order (zlng_group.hip): the stream
// state a range leaves is final when the first half returns, so the next context's rank stage can start while this one's bytes
// are still crossing PCIe (the second half, a blocking copy that may run on another thread: it touches only this context).
int zlng_encode_finish_staged(zlng_ctx* c, size_t out_cap, size_t* out_len, size_t* per_block_out_end) {
    if (!c || !c->pending_in || !out_len) return ZLNG_E_ARG;
    *out_len = 0;
    c->staged_len = 0;
    const size_t bound = zlng_encode_bound(c->pending_len);
    const int rc0 = ensure_out(c, bound);
    if (rc0 != ZLNG_OK) return rc0;
    size_t produced = 0;
    const int rc = zlng_encode_finish_device(c, c->d_out, std::min(bound, out_cap), &produced, per_block_out_end);
    if (rc != ZLNG_OK) return rc;
    c->staged_len = produced;
    *out_len = produced;
    return ZLNG_OK;
}
int zlng_encode_copy_out(zlng_ctx* c, uint8_t* out, size_t n) {
    if (!c || !out || n != c->staged_len) return ZLNG_E_ARG;
    if (n == 0) return ZLNG_OK;
    CTX_HIP(hipSetDevice(c->device));                   // (the calling thread may be a helper that has never touched this device)
    CTX_HIP(hipMemcpyAsync(out, c->d_out, n, hipMemcpyDeviceToHost, c->stream));
    CTX_HIP(hipStreamSynchronize(c->stream));
    c->staged_len = 0;
    return ZLNG_OK;
}

// The measured alternative of SURVEY 8(e) Option C as a per-context switch (see include/zlng.h): k = 0 switches it off.
int zlng_set_host_rank_contexts(zlng_ctx* c, int k) {
    if (!c || !c->is_encode || k < 0) return ZLNG_E_ARG;
    k = std::min(8, k);
    CTX_HIP(hipSetDevice(c->device));
    if (k > 0 && !(c->d_skip && c->stream2 && c->ev2)) {        // all three or none: a half-made set is taken down again
        int rc = ZLNG_OK;
        if (!c->d_skip) rc = dev_alloc(c, &c->d_skip, 256);
        if (rc == ZLNG_OK && !c->stream2 && hipStreamCreateWithFlags(&c->stream2, hipStreamNonBlocking) != hipSuccess) rc = ZLNG_E_DEVICE;
        if (rc == ZLNG_OK && !c->ev2 && hipEventCreateWithFlags(&c->ev2, hipEventDisableTiming) != hipSuccess) rc = ZLNG_E_DEVICE;
        if (rc != ZLNG_OK) {
            if (c->ev2) { hipEventDestroy(c->ev2); c->ev2 = nullptr; }
            if (c->stream2) { hipStreamDestroy(c->stream2); c->stream2 = nullptr; }
            if (c->d_skip) { hipFree(c->d_skip); c->d_skip = nullptr; }
            c->host_rank_contexts = 0;
            return rc;
        }
    }
    c->host_rank_contexts = k;
    return ZLNG_OK;
}

int zlng_get_state(zlng_ctx* c, uint8_t mtf[ZLNG_MTF_STATE], int* current_level) {
    if (!c || !mtf) return ZLNG_E_ARG;
    CTX_HIP(hipSetDevice(c->device));
    CTX_HIP(hipMemcpyAsync(mtf, c->d_mtf, ZLNG_MTF_STATE, hipMemcpyDeviceToHost, c->stream));
    CTX_HIP(hipStreamSynchronize(c->stream));
    if (current_level) *current_level = c->current_level;
    return ZLNG_OK;
}

int zlng_set_state(zlng_ctx* c, const uint8_t mtf[ZLNG_MTF_STATE], int current_level) {
    // current_level is either 0 or the stream's level (src/libzling.cpp:261-266)
    if (!c || !mtf || (current_level != 0 && current_level != c->level)) return ZLNG_E_ARG;
    for (int ctx = 0; ctx < 256; ctx++) {           // every table must be a permutation of 0..255
        bool seen[256] = {false};
        for (int i = 0; i < 256; i++) { uint8_t v = mtf[256 * ctx + i]; if (seen[v]) return ZLNG_E_ARG; seen[v] = true; }
    }
    CTX_HIP(hipSetDevice(c->device));
    CTX_HIP(hipMemcpyAsync(c->d_mtf, mtf, ZLNG_MTF_STATE, hipMemcpyHostToDevice, c->stream));
    CTX_HIP(hipStreamSynchronize(c->stream));
    c->current_level = current_level;
    return ZLNG_OK;
}

int zlng_get_state_device(zlng_ctx* c, void* d_mtf, int* current_level) {
    if (!c || !d_mtf) return ZLNG_E_ARG;
    CTX_HIP(hipSetDevice(c->device));
    CTX_HIP(hipMemcpyAsync(d_mtf, c->d_mtf, ZLNG_MTF_STATE, hipMemcpyDeviceToDevice, c->stream));
    CTX_HIP(hipStreamSynchronize(c->stream));
    if (current_level) *current_level = c->current_level;
    return ZLNG_OK;
}

int zlng_set_state_device(zlng_ctx* c, const void* d_mtf, int current_level) {
    if (!c || !d_mtf || (current_level != 0 && current_level != c->level)) return ZLNG_E_ARG;
    CTX_HIP(hipSetDevice(c->device));
    CTX_HIP(hipMemcpyAsync(c->d_mtf, d_mtf, ZLNG_MTF_STATE, hipMemcpyDeviceToDevice, c->stream));
    CTX_HIP(hipStreamSynchronize(c->stream));
    c->current_level = current_level;
    return ZLNG_OK;
}

static_assert(ZLNG_DEC_E_FLAG == ZLNG_E_FLAG && ZLNG_DEC_E_BLOCKSIZE == ZLNG_E_BLOCKSIZE && ZLNG_DEC_E_CODE1 == ZLNG_E_CODE1 &&
              ZLNG_DEC_E_CODE2 == ZLNG_E_CODE2 && ZLNG_DEC_E_EXBITS == ZLNG_E_EXBITS && ZLNG_DEC_E_LZ == ZLNG_E_LZ &&
              ZLNG_DEC_E_LIMIT == ZLNG_E_ARG, "device-side error codes must match zlng.h");

int zlng_decode_blocks_device(zlng_ctx* c, const void* d_in, size_t in_len, size_t* in_used, void* d_out, size_t out_cap,
                              size_t* out_len, size_t* per_block_out_end) {
    if (!c || c->is_encode || !in_used || !out_len) return ZLNG_E_ARG;
    *in_used = 0;
    *out_len = 0;
    if (in_len == 0) return ZLNG_OK;
    if (!d_in || !d_out) return ZLNG_E_ARG;
    CTX_HIP(hipSetDevice(c->device));
    timer_begin(c);
    DecodeArgs da{static_cast<const uint8_t*>(d_in), (uint64_t)in_len, c->max_blocks, c->max_blocks * kDecSubsPerBlock,
                  (uint64_t)c->max_blocks * ((uint64_t)kTokCapMax + 64), c->d_subs, c->d_blocks, c->d_sub_ntok, c->d_tok, c->d_ring,
                  c->d_mtf, c->d_mtf_snap, static_cast<uint8_t*>(d_out), (uint64_t)out_cap, c->d_summary};
    // tables at call entry (second snapshot slot): the host entry point restores them when the caller's buffer is too small
    CTX_HIP(hipMemcpyAsync(c->d_mtf_snap + ZLNG_MTF_STATE, c->d_mtf, ZLNG_MTF_STATE, hipMemcpyDeviceToDevice, c->stream));
    launch_frame_walk(da, c->stream);
    timer_mark(c, "frame_walk");
    uint64_t sum[7] = {0, 0, 0, 0, 0, 0, 0};
    CTX_HIP(hipMemcpyAsync(sum, c->d_summary, sizeof sum, hipMemcpyDeviceToHost, c->stream));
    CTX_HIP(hipStreamSynchronize(c->stream));
    uint32_t nblk = (uint32_t)sum[2];
    const uint32_t nsub = (uint32_t)sum[3];
    // Complete blocks in front of a framing error are decoded and reported first, like the reference's loop, which emits
    // every block before it throws (src/libzling.cpp:306-420); the caller meets the error at the head of its next call.
    if (nblk == 0) return sum[1] ? -(int)sum[1] : ZLNG_E_TRUNC;      // not even one complete block in the prefix
    if (nsub) launch_huff_decode(da, nsub, c->stream);               // (a prefix of empty blocks -- bare 0x00 flags -- has no sub-block at all)
    timer_mark(c, "huff_decode");
    launch_rolz_decode(da, c->stream);
    timer_mark(c, "rolz_decode");
    CTX_HIP(hipMemcpyAsync(sum, c->d_summary, sizeof sum, hipMemcpyDeviceToHost, c->stream));
    CTX_HIP(hipMemcpyAsync(c->h_blocks.data(), c->d_blocks, nblk * sizeof(DecBlock), hipMemcpyDeviceToHost, c->stream));
    CTX_HIP(hipStreamSynchronize(c->stream));
    CTX_HIP(hipGetLastError());
    const uint32_t bad = (uint32_t)sum[5];
    if (bad < nblk) {                                                // block `bad` failed in the Huffman or the replay stage
        if (bad == 0) return -(int)sum[6];
        nblk = bad;                                                  // the blocks before it are good; the tables are those at its start
    }
    *in_used = (size_t)c->h_blocks[nblk - 1].z_end;
    *out_len = (size_t)(c->h_blocks[nblk - 1].out_off + c->h_blocks[nblk - 1].size);
    if (per_block_out_end) for (uint32_t b = 0; b < nblk; b++) per_block_out_end[b] = (size_t)(c->h_blocks[b].out_off + c->h_blocks[b].size);
    c->last_blocks = nblk;
    return ZLNG_OK;
}

int zlng_decode_blocks(zlng_ctx* c, const uint8_t* in, size_t in_len, size_t* in_used, uint8_t* out, size_t out_cap,
                       size_t* out_len, size_t* per_block_out_end) {
    if (!c || c->is_encode || !in_used || !out_len) return ZLNG_E_ARG;
    *in_used = 0;
    *out_len = 0;
    if (in_len == 0) return ZLNG_OK;
    if (!in || !out) return ZLNG_E_ARG;
    CTX_HIP(hipSetDevice(c->device));
    int rc;
    const size_t raw_cap = (size_t)c->max_blocks * kBlockIn;
    if ((rc = ensure_in(c, in_len + 512)) || (rc = ensure_out(c, raw_cap + 512))) return rc;
    CTX_HIP(hipMemcpyAsync(c->d_in, in, in_len, hipMemcpyHostToDevice, c->stream));
    size_t produced = 0;
    rc = zlng_decode_blocks_device(c, c->d_in, in_len, in_used, c->d_out, raw_cap, &produced, per_block_out_end);
    if (rc != ZLNG_OK) return rc;
    if (produced > out_cap) {                                        // nothing is reported: put the stream state back
        *in_used = 0;
        CTX_HIP(hipMemcpyAsync(c->d_mtf, c->d_mtf_snap + ZLNG_MTF_STATE, ZLNG_MTF_STATE, hipMemcpyDeviceToDevice, c->stream));
        CTX_HIP(hipStreamSynchronize(c->stream));
        return ZLNG_E_CAP;
    }
    CTX_HIP(hipMemcpyAsync(out, c->d_out, produced, hipMemcpyDeviceToHost, c->stream));
    CTX_HIP(hipStreamSynchronize(c->stream));
    *out_len = produced;
    return ZLNG_OK;
}

int zlng_last_timings(zlng_ctx* c, const char** names, float* ms, int cap) {
    if (!c || !c->timer.ok) return 0;
    hipSetDevice(c->device);
    hipStreamSynchronize(c->stream);
    int n = c->timer.n < cap ? c->timer.n : cap;
    for (int i = 0; i < n; i++) {
        names[i] = c->timer.name[i];
        float t = 0;
        if (hipEventElapsedTime(&t, c->timer.ev[i], c->timer.ev[i + 1]) != hipSuccess) t = -1.f;
        ms[i] = t;
    }
    return n;
}

// Test hook: copy an internal per-block buffer of the last encode call to the host, so the
// stage-level parity tests can compare each kernel with the oracle's stage API.
//   what: 0 tokens (u32 x ntok), 1 cuts (SubCut x nsub), 2 freq (u32 x 546 per sub-block, kMaxSub rows),
//         3 lens (u8 x 546 rows), 4 olen (u32 x kMaxSub), 5 ntok/nsub (2 x u32), 6 codes (u16 x 546 rows),
//         7 sub-block output offsets (u64 x kMaxSub), 8 literals per context of the last rank group (u32 x 256; blk ignored)
int zlng_debug_fetch(zlng_ctx* c, int what, int blk, void* dst, size_t bytes) {
    if (!c || !c->is_encode || blk < 0 || (uint32_t)blk >= c->max_blocks || !dst) return ZLNG_E_ARG;
    CTX_HIP(hipSetDevice(c->device));
    CTX_HIP(hipStreamSynchronize(c->stream));
    const void* src = nullptr;
    size_t cap = 0;
    uint32_t two[2];
    switch (what) {
        case 0: src = c->d_tok + (size_t)blk * c->tok_cap; cap = (size_t)c->tok_cap * 4; break;
        case 1: src = c->d_cuts + (size_t)blk * kMaxSub; cap = sizeof(SubCut) * kMaxSub; break;
        case 2: src = c->d_freq + (size_t)blk * kMaxSub * kNsymAll; cap = (size_t)kMaxSub * kNsymAll * 4; break;
        case 3: src = c->d_lens + (size_t)blk * kMaxSub * kNsymAll; cap = (size_t)kMaxSub * kNsymAll; break;
        case 4: src = c->d_olen + (size_t)blk * kMaxSub; cap = (size_t)kMaxSub * 4; break;
        case 6: src = c->d_codes + (size_t)blk * kMaxSub * kNsymAll; cap = (size_t)kMaxSub * kNsymAll * 2; break;
        case 7: src = c->d_sub_off + (size_t)blk * kMaxSub; cap = (size_t)kMaxSub * 8; break;
        case 8: src = c->d_ctx_total; cap = 256 * 4; break;
        case 5:
            CTX_HIP(hipMemcpy(&two[0], c->d_ntok + blk, 4, hipMemcpyDeviceToHost));
            CTX_HIP(hipMemcpy(&two[1], c->d_nsub + blk, 4, hipMemcpyDeviceToHost));
            if (bytes < 8) return ZLNG_E_ARG;
            memcpy(dst, two, 8);
            return ZLNG_OK;
        default: return ZLNG_E_ARG;
    }
    if (bytes > cap) return ZLNG_E_ARG;
    CTX_HIP(hipMemcpy(dst, src, bytes, hipMemcpyDeviceToHost));
    return ZLNG_OK;
}

// Test hook: run K4 (code lengths + canonical codes) on caller-supplied frequency rows (546 counts each: 514 of alphabet 1,
// 32 of alphabet 2), so the tie-heavy golden tables of the reference reach the device heap directly.
int zlng_debug_lengths(zlng_ctx* c, const uint32_t* freq, int nrows, uint8_t* lens, uint16_t* codes) {
    if (!c || !c->is_encode || !freq || !lens || nrows <= 0 || (size_t)nrows > (size_t)c->max_blocks * kMaxSub) return ZLNG_E_ARG;
    CTX_HIP(hipSetDevice(c->device));
    const uint32_t nb = ((uint32_t)nrows + kMaxSub - 1) / kMaxSub;
    std::vector<uint32_t> nsub(nb);
    for (uint32_t b = 0; b < nb; b++) nsub[b] = std::min<uint32_t>(kMaxSub, (uint32_t)nrows - b * kMaxSub);
    CTX_HIP(hipMemcpyAsync(c->d_nsub, nsub.data(), nb * sizeof(uint32_t), hipMemcpyHostToDevice, c->stream));
    CTX_HIP(hipMemcpyAsync(c->d_freq, freq, (size_t)nrows * kNsymAll * sizeof(uint32_t), hipMemcpyHostToDevice, c->stream));
    launch_lengths(huff_args(c, nb, 0, nullptr, 0), c->stream);
    CTX_HIP(hipMemcpyAsync(lens, c->d_lens, (size_t)nrows * kNsymAll, hipMemcpyDeviceToHost, c->stream));
    if (codes) CTX_HIP(hipMemcpyAsync(codes, c->d_codes, (size_t)nrows * kNsymAll * sizeof(uint16_t), hipMemcpyDeviceToHost, c->stream));
    CTX_HIP(hipStreamSynchronize(c->stream));
    CTX_HIP(hipGetLastError());
    return ZLNG_OK;
}

// Test hook: parse passes of the last encode call (1 = the speculated level schedule was right and the pools sufficed).
int zlng_debug_passes(zlng_ctx* c) { return c ? c->last_passes : -1; }

// Undocumented profiling aid (ZLNG_PROFILE=1): copies 16 counters per block of the last parse.
int zlng_debug_counters(zlng_ctx* c, unsigned long long* out, int nblocks) {
    if (!c || !c->d_dbg || nblocks > (int)c->max_blocks) return ZLNG_E_ARG;
    CTX_HIP(hipMemcpy(out, c->d_dbg, (size_t)nblocks * kDbgSlots * sizeof(unsigned long long), hipMemcpyDeviceToHost));
    return ZLNG_OK;
}

void* zlng_stream(zlng_ctx* c) { return c ? (void*)c->stream : nullptr; }

const char* zlng_strerror(int code) {
    switch (code) {
        case ZLNG_OK: return "ok";
        case ZLNG_E_ARG: return "invalid argument";
        case ZLNG_E_NOMEM: return "out of memory";
        case ZLNG_E_CAP: return "output capacity too small";
        case ZLNG_E_DEVICE: return "HIP device error or no gfx950 device";
        case ZLNG_E_PAYLOAD: return "sub-block payload exceeds 393216 bytes";
        case ZLNG_E_FLAG: return "baidu::zling::Decode(): invalid encflag.";
        case ZLNG_E_BLOCKSIZE: return "baidu::zling::Decode(): invalid block size.";
        case ZLNG_E_CODE1: return "baidu::zling::Decode(): invalid huffman stream. (bad code1)";
        case ZLNG_E_CODE2: return "baidu::zling::Decode(): invalid huffman stream. (bad code2)";
        case ZLNG_E_EXBITS: return "baidu::zling::Decode(): invalid huffman stream. (bad ex-bits)";
        case ZLNG_E_LZ: return "baidu::zling::Decode(): lzdecode failed.";
        case ZLNG_E_TRUNC: return "truncated stream";
        default: return "unknown error";
    }
}

}  // extern "C"
