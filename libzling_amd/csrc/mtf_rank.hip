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
        : "vcc")

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

__global__ __launch_bounds__(64) void k_mtf_dense(MtfArgs a) {
    const uint32_t ctx = blockIdx.x;
    const uint32_t lane = threadIdx.x;
    uint8_t* st = a.state + ctx * 256;
    uint32_t t0 = st[lane], t1 = st[64 + lane], t2 = st[128 + lane], t3 = st[192 + lane];

    auto slow_step = [&](uint32_t c) -> uint32_t {                     // rank >= 64
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

    uint8_t* run = a.lit_byte + a.ctx_off[ctx];
    const uint32_t n = a.ctx_total[ctx];
    uint32_t vnext = lane < n ? (uint32_t)run[lane] : 0u;
    for (uint32_t base = 0; base < n; base += 64) {
        const uint32_t v = vnext;
        const uint32_t nidx = base + 64 + lane;
        vnext = nidx < n ? (uint32_t)run[nidx] : 0u;                   // next tile in flight while this one is replayed
        uint32_t ranks = 0;
        if (base + 64 <= n) {
#define RANKSTORE(I, K) asm volatile("v_writelane_b32 %0, %1, %2" : "+v"(ranks) : "s"(I), "i"(K))
            ZLNG_MTF_STEP(0)  ZLNG_MTF_STEP(1)  ZLNG_MTF_STEP(2)  ZLNG_MTF_STEP(3)  ZLNG_MTF_STEP(4)  ZLNG_MTF_STEP(5)  ZLNG_MTF_STEP(6)  ZLNG_MTF_STEP(7)
            ZLNG_MTF_STEP(8)  ZLNG_MTF_STEP(9)  ZLNG_MTF_STEP(10) ZLNG_MTF_STEP(11) ZLNG_MTF_STEP(12) ZLNG_MTF_STEP(13) ZLNG_MTF_STEP(14) ZLNG_MTF_STEP(15)
            ZLNG_MTF_STEP(16) ZLNG_MTF_STEP(17) ZLNG_MTF_STEP(18) ZLNG_MTF_STEP(19) ZLNG_MTF_STEP(20) ZLNG_MTF_STEP(21) ZLNG_MTF_STEP(22) ZLNG_MTF_STEP(23)
            ZLNG_MTF_STEP(24) ZLNG_MTF_STEP(25) ZLNG_MTF_STEP(26) ZLNG_MTF_STEP(27) ZLNG_MTF_STEP(28) ZLNG_MTF_STEP(29) ZLNG_MTF_STEP(30) ZLNG_MTF_STEP(31)
            ZLNG_MTF_STEP(32) ZLNG_MTF_STEP(33) ZLNG_MTF_STEP(34) ZLNG_MTF_STEP(35) ZLNG_MTF_STEP(36) ZLNG_MTF_STEP(37) ZLNG_MTF_STEP(38) ZLNG_MTF_STEP(39)
            ZLNG_MTF_STEP(40) ZLNG_MTF_STEP(41) ZLNG_MTF_STEP(42) ZLNG_MTF_STEP(43) ZLNG_MTF_STEP(44) ZLNG_MTF_STEP(45) ZLNG_MTF_STEP(46) ZLNG_MTF_STEP(47)
            ZLNG_MTF_STEP(48) ZLNG_MTF_STEP(49) ZLNG_MTF_STEP(50) ZLNG_MTF_STEP(51) ZLNG_MTF_STEP(52) ZLNG_MTF_STEP(53) ZLNG_MTF_STEP(54) ZLNG_MTF_STEP(55)
            ZLNG_MTF_STEP(56) ZLNG_MTF_STEP(57) ZLNG_MTF_STEP(58) ZLNG_MTF_STEP(59) ZLNG_MTF_STEP(60) ZLNG_MTF_STEP(61) ZLNG_MTF_STEP(62) ZLNG_MTF_STEP(63)
#undef RANKSTORE
            run[base + lane] = (uint8_t)ranks;
        } else {
#define RANKSTORE(I, K) wrl(ranks, I, K)
            const uint32_t cnt = n - base;
            for (uint32_t k = 0; k < cnt; k++) ZLNG_MTF_STEP(k)
#undef RANKSTORE
            if (lane < cnt) run[base + lane] = (uint8_t)ranks;
        }
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
