// mtf_rank.hip -- K2: the stream-serial literal rank stage.
//
// Replaces ZlingMTFEncoder::Encode (src/libzling_lz.cpp:112-117) as called from the literal
// branch of EncodeImpl (src/libzling_lz.cpp:188).  The 256 tables persist across blocks
// (they are members of the long-lived encoder, src/libzling_lz.h:105, and Reset() does not
// touch them, src/libzling_lz.cpp:197-209), so this stage is one serial chain PER CONTEXT
// over the whole stream.  The parse (K1) leaves literals raw and tags each with its context
// byte, which makes the 256 chains independent of each other.
//
// A single wavefront issues roughly one instruction per 4 cycles, so the hottest chain (the
// context ' ' holds ~30 % of all literals of text) is bound by INSTRUCTIONS PER LITERAL.  The
// stage is therefore split so that the serial part touches nothing but the chain itself:
//
//   K2a  k_lit_tiles<hist>     per 4096-token tile: literals per context            (parallel)
//   K2b  k_lit_scan            per context: exclusive scan over tiles in stream order (parallel)
//   K2c  k_lit_tiles<scatter>  stable partition: literal bytes into one dense run per context
//   K2d  k_mtf_dense           one wavefront per context walks its run: table in 4 VGPRs,
//                              lookup = v_cmp + ballot, swap = two v_writelane  (~11 instr/literal)
//   K2e  k_lit_tiles<gather>   ranks back into the token words
#include "zlng_common.h"
#include "zlng_kernels.h"

namespace zlng {

constexpr uint32_t kLitTile = 4096;                  // tokens per partition tile (one wavefront)

// mtfnext without a division: floor(19i/20) for i < 128, floor(11i/20) otherwise (== zlng_common.h
// mtf_next, i.e. src/tables/gen.py:52-56; checked exhaustively below).
__host__ __device__ constexpr uint32_t mtf_next_fast(uint32_t i) { return i < 128 ? (i * 62263u) >> 16 : (i * 36047u) >> 16; }
constexpr bool mtf_next_fast_ok() {
    for (uint32_t i = 0; i < 256; i++) if (mtf_next_fast(i) != (i < 128 ? (i * 95u) / 100u : (i * 55u) / 100u)) return false;
    return true;
}
static_assert(mtf_next_fast_ok(), "mtf_next_fast must equal int(0.95 i) / int(0.55 i)");

__device__ __forceinline__ uint32_t rdl(uint32_t v, uint32_t lane) { return (uint32_t)__builtin_amdgcn_readlane((int)v, (int)lane); }
// v[lane] = val with wave-uniform val/lane (there is no clang builtin for v_writelane).  gfx9 allows one
// SGPR on the constant bus, M0 as lane select is exempt: the lane goes through M0, written in the same
// statement that reads it.  The lane values here are SALU results (s_ff1 / s_mul / s_lshr), so the
// "VALU-written SGPR as lane select" wait states are not owed; SALU -> M0 -> v_writelane needs none.
__device__ __forceinline__ void wrl(uint32_t& v, uint32_t val, uint32_t lane) {
    asm volatile("s_mov_b32 m0, %2\n\tv_writelane_b32 %0, %1, m0" : "+v"(v) : "s"(val), "s"(lane));
}

// per-lane select by a wave-uniform 64-bit lane mask held in an SGPR pair: lane l gets (mask bit l) ? b : a.
// (Plain C++ makes the compiler rebuild the mask test per lane with v_and + v_cmp_u64; this is one instruction.
//  The mask comes from v_cmp / s_lshr_b64; gfx9 owes no wait states for a VALU reading such an SGPR as a constant.)
__device__ __forceinline__ uint32_t sel(uint32_t a, uint32_t b, uint64_t mask) {
    uint32_t r;
    asm volatile("v_cndmask_b32_e64 %0, %1, %2, %3" : "=v"(r) : "v"(a), "v"(b), "s"(mask));
    return r;
}
__device__ __forceinline__ uint32_t cvec(uint32_t uniform) {           // wave-uniform scalar as a VGPR operand
    uint32_t r;
    asm volatile("v_mov_b32 %0, %1" : "=v"(r) : "s"(uniform));
    return r;
}

__device__ __forceinline__ bool is_literal(uint32_t v) { return (v & 0xFF00u) == 0 && (v >> 16) < 256u; }

// Number of tiles before block b, in stream order (tiny; one lane).
__global__ void k_lit_tile_base(MtfArgs a) {
    if (threadIdx.x != 0 || blockIdx.x != 0) return;
    uint32_t t = 0;
    for (uint32_t b = 0; b < a.nblocks; b++) { a.tile_base[b] = t; t += (a.ntok[b] + kLitTile - 1) / kLitTile; }
    a.tile_base[a.nblocks] = t;
}

// ------------------------------------------------------------------------------ K2a / K2c / K2e
// One wavefront per tile walks its 64 chunks of 64 tokens in stream order.  Stability inside a
// chunk comes from a 64-bit lane mask per context in LDS (atomic OR, then popcount of the lanes below).
enum { kModeHist = 0, kModeScatter = 1, kModeGather = 2 };

template <int MODE>
__global__ __launch_bounds__(64) void k_lit_tiles(MtfArgs a) {
    __shared__ uint32_t run[256];
    __shared__ unsigned long long mask[256];
    const uint32_t blk = blockIdx.y, tile = blockIdx.x, lane = threadIdx.x;
    const uint32_t n = a.ntok[blk];
    if (tile * kLitTile >= n) return;
    const uint32_t gt = a.tile_base[blk] + tile;                       // dense tile number, stream order
    uint32_t* hist = a.tile_hist + (size_t)gt * 256;
    for (uint32_t c = lane; c < 256; c += 64) {
        run[c] = MODE == kModeHist ? 0u : a.ctx_off[c] + hist[c];
        mask[c] = 0;
    }
    __syncthreads();
    uint32_t* t = a.tok + (size_t)blk * kTokCap;
    const unsigned long long below = (1ull << lane) - 1ull;
    for (uint32_t ch = 0; ch < kLitTile / 64; ch++) {
        const uint32_t idx = tile * kLitTile + ch * 64 + lane;
        if (tile * kLitTile + ch * 64 >= n) break;
        const uint32_t v = idx < n ? t[idx] : 0xFFFFFFFFu;
        const bool lit = is_literal(v);
        const uint32_t c = (v >> 16) & 255u;
        if (MODE == kModeHist) {
            if (lit) atomicAdd(&run[c], 1u);
        } else {
            if (lit) atomicOr(&mask[c], 1ull << lane);
            __syncthreads();
            if (lit) {
                const unsigned long long m = mask[c];
                const uint32_t pos = run[c] + (uint32_t)__popcll(m & below);
                if (MODE == kModeScatter) a.lit_byte[pos] = (uint8_t)v;
                else t[idx] = (uint32_t)a.lit_byte[pos] | c << 16;
                if ((m >> lane) == 1ull) { run[c] += (uint32_t)__popcll(m); mask[c] = 0; }   // highest lane of this context
            }
            __syncthreads();
        }
    }
    if (MODE == kModeHist) {
        __syncthreads();
        for (uint32_t c = lane; c < 256; c += 64) hist[c] = run[c];
    }
}

// ------------------------------------------------------------------------------ K2b
// Per context: exclusive scan of tile counts in stream order (in place), total into ctx_total.
__global__ __launch_bounds__(256) void k_lit_scan(MtfArgs a) {
    __shared__ uint32_t wsum[4];
    __shared__ uint32_t carry;
    const uint32_t c = blockIdx.x, tid = threadIdx.x;
    const uint32_t ntiles = a.tile_base[a.nblocks];
    if (tid == 0) carry = 0;
    __syncthreads();
    for (uint32_t t0 = 0; t0 < ntiles; t0 += 256) {
        const uint32_t t = t0 + tid;
        const uint32_t x = t < ntiles ? a.tile_hist[(size_t)t * 256 + c] : 0u;
        uint32_t incl = x;
        for (int o = 1; o < 64; o <<= 1) { const uint32_t y = __shfl_up(incl, o); if ((tid & 63) >= (uint32_t)o) incl += y; }
        if ((tid & 63) == 63) wsum[tid >> 6] = incl;
        __syncthreads();
        uint32_t before = carry;
        for (uint32_t w = 0; w < (tid >> 6); w++) before += wsum[w];
        if (t < ntiles) a.tile_hist[(size_t)t * 256 + c] = before + incl - x;
        __syncthreads();
        if (tid == 255) carry = before + incl;
        __syncthreads();
    }
    if (tid == 0) a.ctx_total[c] = carry;
}

// Start of every context's dense run (exclusive scan of the 256 totals; one wavefront).
__global__ __launch_bounds__(64) void k_ctx_offsets(MtfArgs a) {
    const uint32_t lane = threadIdx.x;
    uint32_t v[4], s = 0;
    for (int k = 0; k < 4; k++) { v[k] = a.ctx_total[lane * 4 + k]; s += v[k]; }
    uint32_t incl = s;
    for (int o = 1; o < 64; o <<= 1) { const uint32_t y = __shfl_up(incl, o); if (lane >= (uint32_t)o) incl += y; }
    uint32_t off = incl - s;
    for (int k = 0; k < 4; k++) { a.ctx_off[lane * 4 + k] = off; off += v[k]; }
}

// ------------------------------------------------------------------------------ K2d
// ZlingMTFEncoder::Encode on a dense run.  The table of the context lives in four VGPRs: lane l of
// t[r] holds table[64 r + l].  rank = position of c (compare + ballot, no index[] array);
// swap with the entry at mtfnext[rank] = two v_writelane.  Ranks < 64 -- almost every literal of
// text -- touch t0 only.
// One literal.  Ranks 0..20 (n = rank - 1: swap with the left neighbour; the bulk of text literals)
// never leave the vector unit: with vcc = lanes whose entry differs from c,
//     t0[l] = vcc ? t0[l] : t0[l-1]          one DPP select (wave_shr:1; lane 0 keeps its value, so rank 0 is a no-op)
//     t0[l] = hit[l+1] ? c : t0[l]           one select with the hit mask shifted right by one
// The update is applied speculatively and undone when the hit was not in lanes 0..20, so the table's
// dependency chain never waits for the scalar rank, which is only needed for the output lane.
// Hazards inside the block (gfx9): a DPP source must be >= 2 wait states behind its VALU writer -- the
// previous writer of t0 is the second select of the previous step, followed by s_ff1, v_writelane and this
// step's v_cmp / s_not / s_lshr; SALU reads of the VALU-written vcc are interlocked; the v_writelane lane
// select is an immediate.
#define ZLNG_MTF_FAST(K, M0)                                                                        \
    asm volatile(                                                                                  \
        "v_mov_b32 %[cv], %[c]\n\t"                                                                \
        "v_cmp_ne_u32_e32 vcc, %[cv], %[t0]\n\t"                                                   \
        "s_not_b64 %[m0], vcc\n\t"                                                                 \
        "s_lshr_b64 %[m1], %[m0], 1\n\t"                                                           \
        "v_cndmask_b32_dpp %[t0], %[t0], %[t0], vcc wave_shr:1 row_mask:0xf bank_mask:0xf\n\t"     \
        "v_cndmask_b32_e64 %[t0], %[t0], %[cv], %[m1]\n\t"                                         \
        "s_ff1_i32_b64 %[i], %[m0]\n\t"                                                            \
        : [t0] "+v"(t0), [m0] "=&s"(M0), [m1] "=&s"(m1_), [i] "=&s"(i), [cv] "=&v"(cv)             \
        : [c] "s"(c)                                                                               \
        : "vcc", "scc")

#define ZLNG_MTF_STEP(K)                                                                           \
    {                                                                                              \
        const uint32_t c = rdl(v, (K));                                                            \
        uint64_t m0, m1_;                                                                          \
        uint32_t i, cv;                                                                            \
        ZLNG_MTF_FAST(K, m0);                                                                      \
        if (__builtin_expect(i > 20u, 0)) {         /* s_ff1 gives 0xFFFFFFFF when there is no hit */ \
            if (m0) {                       /* rank 21..63: undo the neighbour swap, do the real one */ \
                const uint32_t up = (uint32_t)__builtin_amdgcn_update_dpp(0, (int)t0, 0x130, 0xf, 0xf, true); /* wave_shl:1 */ \
                t0 = sel(t0, up, m1_);                                                             \
                t0 = sel(t0, cv, m0);                                                              \
                const uint32_t nx = (i * 62263u) >> 16;                                            \
                const uint32_t d = rdl(t0, nx);                                                    \
                wrl(t0, d, i);                                                                     \
                wrl(t0, c, nx);                                                                    \
            } else {                                                                               \
                i = slow_step(c);                                                                  \
            }                                                                                      \
        }                                                                                          \
        RANKSTORE(i, K);                                                                           \
    }

// Full tiles use a tighter form of the same step, shaped by scripts/ubench/mtfstep.hip (one lone wavefront on
// gfx950: the shipped order above costs 32 ns/step, this one 25 ns): the table chain never leaves the vector
// unit -- up[l] = t0[l+1] (DPP wave_shl:1), "c sits one lane up" = v_cmp_eq(c, up) replaces s_not + s_lshr, and
// the value that lane takes is up itself, so c is only ever a scalar operand (no v_mov) -- the slow-path branch
// hangs off ONE s_andn2 of vcc with the lane 0..20 mask (SCC = "hit in the fast lanes"), vector and scalar
// instructions stay grouped (interleaving them measured slower), and eight literals are fetched by eight
// back-to-back v_readlane in front of their steps.  up's lane 63 is never written (no lane 64 to read): it keeps
// 0xFFFFFFFF, which no literal equals.
// DPP hazard (gfx9: VALU write -> DPP read of the same VGPR needs 2 wait states): t0's last VALU writer is the
// previous step's second select, followed by s_andn2 / s_cbranch / s_ff1 / v_writelane; the slow path ends in s_nop 1.
// One statement covers eight literals (an asm goto with outputs crashes this compiler's instruction selection,
// and a C-level branch per step costs one more scalar instruction): a step whose hit is not in lanes 0..20
// jumps to the end of the statement with its (empty) fast-lane mask in m0; the C code after the statement finds
// the step from the first rank lane still holding the tile's 0xFFFFFFFF fill, replays that literal on the slow
// path and the rest of the group with ZLNG_MTF_STEP.
#define ZLNG_MTF_G_STEP(C, K)                                                                                   \
    "v_mov_b32_dpp %[up], %[t0] wave_shl:1 row_mask:0xf bank_mask:0xf\n\t"                                      \
    "v_cmp_ne_u32_e32 vcc, %[" #C "], %[t0]\n\t"                                                                \
    "v_cmp_eq_u32_e64 %[m1], %[" #C "], %[up]\n\t"                                                              \
    "v_cndmask_b32_dpp %[t0], %[t0], %[t0], vcc wave_shr:1 row_mask:0xf bank_mask:0xf\n\t"                      \
    "v_cndmask_b32_e64 %[t0], %[t0], %[up], %[m1]\n\t"                                                          \
    "s_andn2_b64 %[hm], 0x1fffff, vcc\n\t"                                                                      \
    "s_cbranch_scc0 1" #K "f\n\t"                                                                               \
    "s_ff1_i32_b64 %[i], %[hm]\n\t"                                                                             \
    "v_writelane_b32 %[ranks], %[i], " #K "\n\t"                                                                \
    "2" #K ":\n\t"
// Out-of-line part of step K (all 64 sit behind the 64 steps): the hit was not in lanes 0..20.
//   rank 21..63: the step has already swapped c with its left neighbour; from that state the reference's
//     swap(table[rank], table[mtfnext[rank]]) is  t0[rank-1] = t0[rank];  t0[rank] = t0[next];  t0[next] = c
//     (next <= rank - 2 from rank 21 on, so t0[next] is untouched).  mtfnext = (rank * 62263) >> 16 below 128.
//   rank >= 64 (c not in t0, nothing was changed): leave the statement with hm = 0.
// v_readlane / v_writelane lane selects come from SALU results or M0 (no wait states owed); the closing s_nop
// covers the v_writelane -> DPP read of t0 in the next step.
#define ZLNG_MTF_G_SLOW(C, K)                                                                                   \
    "1" #K ":\n\t"                                                                                              \
    "s_not_b64 %[hm], vcc\n\t"                                                                                  \
    "s_cbranch_scc0 9f\n\t"                                                                                     \
    "s_ff1_i32_b64 %[i], %[hm]\n\t"                                                                             \
    "s_mul_i32 %[nx], %[i], 0xf337\n\t"                                                                         \
    "s_lshr_b32 %[nx], %[nx], 16\n\t"                                                                           \
    "v_readlane_b32 %[d], %[t0], %[i]\n\t"                                                                      \
    "s_sub_u32 m0, %[i], 1\n\t"                                                                                 \
    "v_writelane_b32 %[t0], %[d], m0\n\t"                                                                       \
    "v_readlane_b32 %[d], %[t0], %[nx]\n\t"                                                                     \
    "s_mov_b32 m0, %[i]\n\t"                                                                                    \
    "v_writelane_b32 %[t0], %[d], m0\n\t"                                                                       \
    "s_mov_b32 m0, %[nx]\n\t"                                                                                   \
    "v_writelane_b32 %[t0], %[" #C "], m0\n\t"                                                                  \
    "v_writelane_b32 %[ranks], %[i], " #K "\n\t"                                                                \
    "s_nop 1\n\t"                                                                                               \
    "s_branch 2" #K "b\n\t"
#define ZLNG_MTF_G8(M, A, B, C, D, E, F, G, H) M(c0, A) M(c1, B) M(c2, C) M(c3, D) M(c4, E) M(c5, F) M(c6, G) M(c7, H)
#define ZLNG_MTF_G_FETCH(A, B, C, D, E, F, G, H)                                                                \
    "v_readlane_b32 %[c0], %[v], " #A "\n\tv_readlane_b32 %[c1], %[v], " #B "\n\t"                              \
    "v_readlane_b32 %[c2], %[v], " #C "\n\tv_readlane_b32 %[c3], %[v], " #D "\n\t"                              \
    "v_readlane_b32 %[c4], %[v], " #E "\n\tv_readlane_b32 %[c5], %[v], " #F "\n\t"                              \
    "v_readlane_b32 %[c6], %[v], " #G "\n\tv_readlane_b32 %[c7], %[v], " #H "\n\t"
#define ZLNG_MTF_G_FAST(A, B, C, D, E, F, G, H) ZLNG_MTF_G_FETCH(A, B, C, D, E, F, G, H) ZLNG_MTF_G8(ZLNG_MTF_G_STEP, A, B, C, D, E, F, G, H)
// (A slow part runs before its group's successor fetches c0..c7 again, so its literal is still in its register.)

// A whole 64-literal tile in ONE statement.  hm == 0 afterwards: a literal of rank >= 64 stopped it; the first
// rank lane still holding the tile's 0xFFFFFFFF fill is that literal.
#define ZLNG_MTF_TILE()                                                                                         \
    asm volatile(                                                                                               \
        ZLNG_MTF_G_FAST(0, 1, 2, 3, 4, 5, 6, 7)         ZLNG_MTF_G_FAST(8, 9, 10, 11, 12, 13, 14, 15)           \
        ZLNG_MTF_G_FAST(16, 17, 18, 19, 20, 21, 22, 23) ZLNG_MTF_G_FAST(24, 25, 26, 27, 28, 29, 30, 31)         \
        ZLNG_MTF_G_FAST(32, 33, 34, 35, 36, 37, 38, 39) ZLNG_MTF_G_FAST(40, 41, 42, 43, 44, 45, 46, 47)         \
        ZLNG_MTF_G_FAST(48, 49, 50, 51, 52, 53, 54, 55) ZLNG_MTF_G_FAST(56, 57, 58, 59, 60, 61, 62, 63)         \
        "s_branch 9f\n\t"                                                                                       \
        ZLNG_MTF_G8(ZLNG_MTF_G_SLOW, 0, 1, 2, 3, 4, 5, 6, 7)         ZLNG_MTF_G8(ZLNG_MTF_G_SLOW, 8, 9, 10, 11, 12, 13, 14, 15)   \
        ZLNG_MTF_G8(ZLNG_MTF_G_SLOW, 16, 17, 18, 19, 20, 21, 22, 23) ZLNG_MTF_G8(ZLNG_MTF_G_SLOW, 24, 25, 26, 27, 28, 29, 30, 31) \
        ZLNG_MTF_G8(ZLNG_MTF_G_SLOW, 32, 33, 34, 35, 36, 37, 38, 39) ZLNG_MTF_G8(ZLNG_MTF_G_SLOW, 40, 41, 42, 43, 44, 45, 46, 47) \
        ZLNG_MTF_G8(ZLNG_MTF_G_SLOW, 48, 49, 50, 51, 52, 53, 54, 55) ZLNG_MTF_G8(ZLNG_MTF_G_SLOW, 56, 57, 58, 59, 60, 61, 62, 63) \
        "9:"                                                                                                    \
        : [t0] "+v"(t0), [up] "+v"(up), [ranks] "+v"(ranks), [hm] "=&s"(hm_), [m1] "=&s"(m1_), [i] "=&s"(i_),   \
          [nx] "=&s"(nx_), [d] "=&s"(d_), [c0] "=&s"(c0), [c1] "=&s"(c1), [c2] "=&s"(c2), [c3] "=&s"(c3),       \
          [c4] "=&s"(c4), [c5] "=&s"(c5), [c6] "=&s"(c6), [c7] "=&s"(c7)                                        \
        : [v] "v"(v)                                                                                            \
        : "vcc", "scc")

__global__ __launch_bounds__(64) void k_mtf_dense(MtfArgs a) {
    const uint32_t ctx = blockIdx.x;
    const uint32_t lane = threadIdx.x;
    uint8_t* st = a.state + ctx * 256;
    uint32_t t0 = st[lane], t1 = st[64 + lane], t2 = st[128 + lane], t3 = st[192 + lane];

    auto slow_step = [&](uint32_t c) __attribute__((always_inline)) -> uint32_t {                     // rank >= 64
        const uint64_t m1 = __ballot(t1 == c), m2 = __ballot(t2 == c), m3 = __ballot(t3 == c);
        const uint32_t i = m1 ? 64 + (uint32_t)__builtin_ctzll(m1)
                              : (m2 ? 128 + (uint32_t)__builtin_ctzll(m2) : 192 + (uint32_t)__builtin_ctzll(m3));
        const uint32_t nx = mtf_next_fast(i);
        uint32_t d;
        switch (nx >> 6) { case 0: d = rdl(t0, nx & 63); break; case 1: d = rdl(t1, nx & 63); break;
                           case 2: d = rdl(t2, nx & 63); break; default: d = rdl(t3, nx & 63); break; }
        switch (i >> 6) { case 1: wrl(t1, d, i & 63); break; case 2: wrl(t2, d, i & 63); break; default: wrl(t3, d, i & 63); break; }
        switch (nx >> 6) { case 0: wrl(t0, c, nx & 63); break; case 1: wrl(t1, c, nx & 63); break;
                           case 2: wrl(t2, c, nx & 63); break; default: wrl(t3, c, nx & 63); break; }
        return i;
    };

    uint32_t up = 0xFFFFFFFFu;                                         // ZLNG_MTF_TILE: t0 shifted down one lane
    uint8_t* run = a.lit_byte + a.ctx_off[ctx];
    const uint32_t n = a.ctx_total[ctx];
    uint32_t vnext = lane < n ? (uint32_t)run[lane] : 0u;
    for (uint32_t base = 0; base < n; base += 64) {
        const uint32_t v = vnext;
        const uint32_t nidx = base + 64 + lane;
        vnext = nidx < n ? (uint32_t)run[nidx] : 0u;                   // next tile in flight while this one is replayed
        uint32_t ranks = 0xFFFFFFFFu;
        if (base + 64 <= n) {
            uint32_t c0, c1, c2, c3, c4, c5, c6, c7, i_, nx_, d_;
            uint64_t hm_, m1_;
            ZLNG_MTF_TILE();
            if (__builtin_expect(hm_ == 0, 0)) {
#define RANKSTORE(I, K) wrl(ranks, I, K)
                const uint32_t kk = (uint32_t)__builtin_ctzll(__ballot(ranks == 0xFFFFFFFFu));
                const uint32_t r = slow_step(rdl(v, kk));
                wrl(ranks, r, kk);
                for (uint32_t k = kk + 1; k < 64u; k++) ZLNG_MTF_STEP(k)
#undef RANKSTORE
            }
        } else {
#define RANKSTORE(I, K) wrl(ranks, I, K)
            const uint32_t cnt = n - base;
            for (uint32_t k = 0; k < cnt; k++) ZLNG_MTF_STEP(k)
#undef RANKSTORE
        }
        // take delivery of the next tile BEFORE issuing the rank store: vmcnt retires in order, so a wait for
        // the (long finished) load placed after the store would also wait for the store's round trip
        asm volatile("" : "+v"(vnext));
        if (base + lane < n) run[base + lane] = (uint8_t)ranks;
    }
    st[lane] = (uint8_t)t0; st[64 + lane] = (uint8_t)t1; st[128 + lane] = (uint8_t)t2; st[192 + lane] = (uint8_t)t3;
}

void launch_mtf_rank(const MtfArgs& a, hipStream_t s) {
    const dim3 tiles((unsigned)(kTokCap / kLitTile), a.nblocks);
    hipLaunchKernelGGL(k_lit_tile_base, dim3(1), dim3(64), 0, s, a);
    hipLaunchKernelGGL(k_lit_tiles<kModeHist>, tiles, dim3(64), 0, s, a);
    hipLaunchKernelGGL(k_lit_scan, dim3(256), dim3(256), 0, s, a);
    hipLaunchKernelGGL(k_ctx_offsets, dim3(1), dim3(64), 0, s, a);
    hipLaunchKernelGGL(k_lit_tiles<kModeScatter>, tiles, dim3(64), 0, s, a);
    hipLaunchKernelGGL(k_mtf_dense, dim3(256), dim3(64), 0, s, a);
    hipLaunchKernelGGL(k_lit_tiles<kModeGather>, tiles, dim3(64), 0, s, a);
}

}  // namespace zlng
