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
//   K2d  k_mtf_chain           one wavefront per context carries the TABLE through its run (six issue slots per
//                              literal, table front in chain layout in one VGPR) and leaves a snapshot per 64 literals
//   K2d' k_mtf_replay          every tile's ranks from its snapshot, one lane per tile                  (parallel)
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
// per-lane select by a wave-uniform 64-bit lane mask held in an SGPR pair: lane l gets (mask bit l) ? b : a.
// (Plain C++ makes the compiler rebuild the mask test per lane with v_and + v_cmp_u64; this is one instruction.)
// gfx940 / gfx950 owe TWO wait states between a VALU that writes an SGPR or VCC (v_cmp, v_readlane, v_readfirstlane) and a VALU
// that reads it (LLVM's hazard recognizer pads a plain v_cmp; v_cndmask with s_nop on gfx950 and not on gfx90a; it does not look
// inside an asm statement, and these operands often come straight from a v_readlane / a ballot): the pad is part of the statement.
// sel_k is the same select for a mask that is a compile-time constant (materialised by the scalar unit: nothing owed).
__device__ __forceinline__ uint32_t sel(uint32_t a, uint32_t b, uint64_t mask) {
    uint32_t r;
    asm volatile("s_nop 1\n\tv_cndmask_b32_e64 %0, %1, %2, %3" : "=v"(r) : "v"(a), "v"(b), "s"(mask));
    return r;
}
__device__ __forceinline__ uint32_t sel_k(uint32_t a, uint32_t b, uint64_t const_mask) {
    uint32_t r;
    asm volatile("v_cndmask_b32_e64 %0, %1, %2, %3" : "=v"(r) : "v"(a), "v"(b), "s"(const_mask));
    return r;
}
__device__ __forceinline__ uint32_t cvec(uint32_t uniform) {           // wave-uniform scalar as a VGPR operand
    uint32_t r;
    asm volatile("s_nop 1\n\tv_mov_b32 %0, %1" : "=v"(r) : "s"(uniform));
    return r;
}
// v[lane] = val with wave-uniform val / lane, on the slow paths only (the tile statements write lanes with constant selects).  There
// is no clang builtin for v_writelane, and its scalar lane select would have to go through M0 beside a scalar value (one
// constant-bus read on gfx9) -- M0 is a reserved register an inline asm cannot clobber safely (round 3 did).  A select under a
// one-hot lane mask costs two more instructions and touches nothing reserved.
__device__ __forceinline__ void wrl(uint32_t& v, uint32_t val, uint32_t lane) { v = sel(v, cvec(val), 1ull << lane); }

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
    uint32_t* t = a.tok + (size_t)blk * a.tok_cap;
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
    // every run starts on a 64-byte line: k_mtf_dense fetches whole tiles of 64 literals with one scalar load
    for (int k = 0; k < 4; k++) { v[k] = (a.ctx_total[lane * 4 + k] + 63u) & ~63u; s += v[k]; }
    uint32_t incl = s;
    for (int o = 1; o < 64; o <<= 1) { const uint32_t y = __shfl_up(incl, o); if (lane >= (uint32_t)o) incl += y; }
    uint32_t off = incl - s;
    for (int k = 0; k < 4; k++) { a.ctx_off[lane * 4 + k] = off; off += v[k]; }
}

// ------------------------------------------------------------------------------ K2d
// ZlingMTFEncoder::Encode (src/libzling_lz.cpp:112-117) on a dense run: one wavefront per context carries the TABLE through its
// literals and leaves one snapshot of it (256 B) per tile of 64; k_mtf_replay recomputes every tile's ranks from its snapshot, all
// tiles at once.  The wavefront is bound by issue slots per literal (a lone wavefront issues one instruction per ~5.8 cycles and a
// dependent one per ~9.7), so the step is shaped instruction by instruction (scripts/ubench/gstep.hip, round 4):
//
// * CHAIN LAYOUT.  mtfnext (src/tables/gen.py:52-56) is a tree on the table positions: 0 <- 1 <- ... <- 19, position 19 has the
//   children 20 and 21, then two interleaved chains (20 <- 22 <- ..., 21 <- 23 <- ...), position 38 has the children 40 and 41,
//   three chains from there.  The first 60 positions are laid into one register as THREE PATHS in which every position's swap
//   partner mtfnext[p] is its LEFT NEIGHBOUR LANE:
//       lane 0        off (EXEC)                                       lanes 39..54  21 23 .. 39 42 45 .. 57   (head 21 -> 19)
//       lanes 1..37   0 1 .. 19 20 22 .. 40 43 46 .. 58                lane  55      pad
//       lane 38       pad                                              lanes 56..62  41 44 .. 59              (head 41 -> 38)
//                                                                      lane  63      pad
//   so ONE neighbour-swap step serves ranks 0..59 except the two heads (round 3 kept positions in lane order: ranks 21..63 -- 2.5 %
//   of the blank's literals on the benchmark text, a third of them on source text -- left the step for a ~75 ns repair).
// * FOUR table instructions instead of five: with vcc = lanes whose entry differs from the literal c,
//       tf[l] = vcc[l] ? tf[l] : tf[l-1]        v_cndmask_b32_dpp wave_shr:1   (the hit lane takes its left neighbour)
//       vcc   = vcc >> 1 (arithmetic)           s_ashr_i64                     (the hit mask one lane down, inverted; lane 63 never lands)
//       tf[l] = vcc[l] ? tf[l] : c              v_cndmask_b32_sdwa             (the lane below takes c)
//   c is an SDWA byte of a VECTOR register that holds the same four literals in every lane (global_load_dwordx4 with a wave-uniform
//   address, sixteen literals each): as a scalar it would be a second constant-bus read beside VCC.  Round 3 built the shifted mask
//   with v_mov_dpp + v_cmp on the vector unit (two instructions) and took c from there.
// * The table starts in lane 1 and lane 0 is switched off: rank 0 is a no-op by itself (a DPP read of a lane that EXEC disables
//   suppresses the write -- measured, gstep DPPX -- and the landing lane 0 is off), and "hit in a lane the step serves" is ONE
//   s_andn2 of the SHIFTED mask against a constant (SCC), branch one step late (behind the next compare, which only reads).
//   Six issue slots: 14.5 ns per literal against 17.1 for round 3's seven.
// * Out of line (entered from inside step K + 1, tf as step K left it): the literal was at a HEAD (the step has swapped it with
//   the pad on its left: two v_readlane + three v_writelane with constant lanes put it to its partner, 19 or 38), or it is not in the
//   front at all: the statement is left (lv = K + 1), slow_step() walks positions 60..255 (tx: positions 60..63 in the four lanes
//   the front leaves free; t1..t3: lane = position & 63) and the statement is re-entered at step K + 1 through a table of branches.
// scripts/experiments/mtf_chain_model.c is this step in plain C, lane by lane (tests/test_chain_model.py runs it against the
// reference's ranks on the CPU).
#define ZLNG_FRONT 60
__host__ __device__ constexpr int chain_lane(int p) {                   // lane of front position p (0 <= p < ZLNG_FRONT)
    return p <= 19 ? p + 1
         : (p <= 40 && (p & 1) == 0) ? 21 + (p - 20) / 2
         : (p >= 43 && (p - 43) % 3 == 0) ? 32 + (p - 43) / 3
         : (p <= 39 && (p & 1) != 0) ? 39 + (p - 21) / 2
         : (p >= 42 && (p - 42) % 3 == 0) ? 49 + (p - 42) / 3
         : 56 + (p - 41) / 3;
}
__host__ __device__ constexpr int chain_pos(int lane) {                 // front position held by a lane, -1 for lane 0 and the pads
    for (int p = 0; p < ZLNG_FRONT; p++) if (chain_lane(p) == lane) return p;
    return -1;
}
constexpr bool chain_layout_ok() {
    int used = 0;
    for (int l = 0; l < 64; l++) used += chain_pos(l) >= 0;
    if (used != ZLNG_FRONT || chain_pos(0) >= 0 || chain_pos(38) >= 0 || chain_pos(55) >= 0 || chain_pos(63) >= 0) return false;
    for (int p = 1; p < ZLNG_FRONT; p++) {
        if (p == 21 || p == 41) continue;
        if (chain_lane((int)mtf_next_fast((uint32_t)p)) != chain_lane(p) - 1) return false;          // partner = left neighbour
    }
    // the heads and their partners, the coupling positions (mtfnext of 60..63), and nothing below 60 is reached from beyond 63
    return mtf_next_fast(21) == 19 && mtf_next_fast(41) == 38 && chain_lane(21) == 39 && chain_lane(41) == 56 && chain_lane(19) == 20 &&
           chain_lane(38) == 30 && mtf_next_fast(60) == 57 && mtf_next_fast(61) == 57 && mtf_next_fast(62) == 58 && mtf_next_fast(63) == 59 &&
           mtf_next_fast(64) == 60 && chain_lane(57) == 54 && chain_lane(58) == 37 && chain_lane(59) == 62;
}
static_assert(chain_layout_ok(), "chain layout of the table front");
constexpr uint64_t chain_allow_x() {            // bit l - 1 for every lane l whose hit the plain step serves (all but the heads and pads)
    uint64_t m = 0;
    for (int l = 1; l < 64; l++) { const int p = chain_pos(l); if (p >= 0 && p != 21 && p != 41) m |= 1ull << (l - 1); }
    return m;
}
static_assert(chain_allow_x() == 0x3f3fff9fffffffffull, "scripts/experiments/mtf_chain_model.c prints the same mask");

// ZLNG_STEP_PAD: the second of the two wait states gfx950 owes between the compare's write of VCC and the DPP select's read of it
// (the late branch is the first).  Round 4 shipped the step without it -- a lone wavefront issues an instruction every ~5.8 cycles and
// every soak was bit-exact -- but that rests on issue timing nobody documents; scripts/ubench/gstep.hip G7N: +0.8 ns per literal.
// (-DZLNG_STEP_PAD_OFF builds the unpadded step for an A/B measurement; tests/test_isa_hygiene.py fails on such a build.)
#ifdef ZLNG_STEP_PAD_OFF
#define ZLNG_STEP_PAD
#else
#define ZLNG_STEP_PAD "s_nop 0\n\t"
#endif
#define ZLNG_C_STEP(PK, B, K, KP)                                                                               \
    "2" #K ":\n\t"                                                                                              \
    "v_cmp_ne_u32_sdwa vcc, %[" #PK "], %[tf] src0_sel:BYTE_" #B " src1_sel:DWORD\n\t"                          \
    "s_cbranch_scc0 1" #KP "f\n\t"                                                                              \
    ZLNG_STEP_PAD                                                                                               \
    "v_cndmask_b32_dpp %[tf], %[tf], %[tf], vcc wave_shr:1 row_mask:0xf bank_mask:0xf\n\t"                      \
    "s_ashr_i64 vcc, vcc, 1\n\t"                                                                                \
    "v_cndmask_b32_sdwa %[tf], %[" #PK "], %[tf], vcc dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:BYTE_" #B " src1_sel:DWORD\n\t" \
    "s_andn2_b64 s[98:99], %[allow], vcc\n\t"
// Out-of-line part of step K, entered from step KN = K + 1 behind its compare (vcc is recomputed there; SCC is set again before
// the way back, whose late branch is then evaluated a second time).  d = the literal (every active lane of PK holds it).
//   c on a pad lane (38 / 55): it was at the head to the right (39 / 56) and the step swapped it with the pad:
//       head = partner's symbol, partner (lane 20 = position 19 / lane 30 = position 38) = c (read back from the pad lane), pad back.
//   c nowhere in tf (the pads and lane 0 hold values no byte equals): leave with lv = K + 1 and d = c.
// v_readlane / v_writelane ignore EXEC; their lane selects are constants; the SGPR a v_readlane wrote is read as DATA two
// instructions later (no wait states owed for that use); the writes of tf are four instructions ahead of step KN's DPP read.
#define ZLNG_C_SLOW(PK, B, K, KN)                                                                               \
    "1" #K ":\n\t"                                                                                              \
    "v_cmp_eq_u32_sdwa vcc, %[" #PK "], %[tf] src0_sel:BYTE_" #B " src1_sel:DWORD\n\t"                          \
    "s_bitcmp1_b64 vcc, 38\n\t"                                                                                 \
    "s_cbranch_scc0 3" #K "f\n\t"                                                                               \
    "v_readlane_b32 %[da], %[tf], 20\n\t"                                                                       \
    "v_readlane_b32 %[d], %[tf], 38\n\t"                                                                        \
    "v_writelane_b32 %[tf], %[pad1], 38\n\t"                                                                    \
    "v_writelane_b32 %[tf], %[da], 39\n\t"                                                                      \
    "v_writelane_b32 %[tf], %[d], 20\n\t"                                                                       \
    "s_cmp_eq_u32 0, 0\n\t"                                                                                     \
    "s_branch 2" #KN "b\n\t"                                                                                    \
    "3" #K ":\n\t"                                                                                              \
    "s_bitcmp1_b64 vcc, 55\n\t"                                                                                 \
    "s_cbranch_scc0 4" #K "f\n\t"                                                                               \
    "v_readlane_b32 %[da], %[tf], 30\n\t"                                                                       \
    "v_readlane_b32 %[d], %[tf], 55\n\t"                                                                        \
    "v_writelane_b32 %[tf], %[pad2], 55\n\t"                                                                    \
    "v_writelane_b32 %[tf], %[da], 56\n\t"                                                                      \
    "v_writelane_b32 %[tf], %[d], 30\n\t"                                                                       \
    "s_cmp_eq_u32 0, 0\n\t"                                                                                     \
    "s_branch 2" #KN "b\n\t"                                                                                    \
    "4" #K ":\n\t"                                                                                              \
    "v_readfirstlane_b32 %[d], %[" #PK "]\n\t"                                                                  \
    "s_mov_b32 %[lv], " #K "+1\n\t"                                                                             \
    "s_bfe_u32 %[d], %[d], (8 * " #B ") | (8 << 16)\n\t"                                                        \
    "s_branch 9f\n\t"
#define ZLNG_C_FAST(PK, Z, A, B, C, D)                                                                          \
    ZLNG_C_STEP(PK, 0, A, Z) ZLNG_C_STEP(PK, 1, B, A) ZLNG_C_STEP(PK, 2, C, B) ZLNG_C_STEP(PK, 3, D, C)
#define ZLNG_C_COLD(PK, A, B, C, D, N)                                                                          \
    ZLNG_C_SLOW(PK, 0, A, B) ZLNG_C_SLOW(PK, 1, B, C) ZLNG_C_SLOW(PK, 2, C, D) ZLNG_C_SLOW(PK, 3, D, N)
#define ZLNG_C_BODY                                                                                             \
        ZLNG_C_FAST(p0, 19, 0, 1, 2, 3)      ZLNG_C_FAST(p1, 3, 4, 5, 6, 7)                                     \
        ZLNG_C_FAST(p2, 7, 8, 9, 10, 11)     ZLNG_C_FAST(p3, 11, 12, 13, 14, 15)                                \
        ZLNG_C_FAST(p4, 15, 16, 17, 18, 19)  ZLNG_C_FAST(p5, 19, 20, 21, 22, 23)                                \
        ZLNG_C_FAST(p6, 23, 24, 25, 26, 27)  ZLNG_C_FAST(p7, 27, 28, 29, 30, 31)                                \
        ZLNG_C_FAST(p8, 31, 32, 33, 34, 35)  ZLNG_C_FAST(p9, 35, 36, 37, 38, 39)                                \
        ZLNG_C_FAST(p10, 39, 40, 41, 42, 43) ZLNG_C_FAST(p11, 43, 44, 45, 46, 47)                               \
        ZLNG_C_FAST(p12, 47, 48, 49, 50, 51) ZLNG_C_FAST(p13, 51, 52, 53, 54, 55)                               \
        ZLNG_C_FAST(p14, 55, 56, 57, 58, 59) ZLNG_C_FAST(p15, 59, 60, 61, 62, 63)                               \
        "264:\n\t"                                                                                              \
        "s_cbranch_scc0 163f\n\t"                                                                               \
        "s_mov_b32 %[lv], 0\n\t"                                                                                \
        "s_branch 9f\n\t"                                                                                       \
        ZLNG_C_COLD(p0, 0, 1, 2, 3, 4)       ZLNG_C_COLD(p1, 4, 5, 6, 7, 8)                                     \
        ZLNG_C_COLD(p2, 8, 9, 10, 11, 12)    ZLNG_C_COLD(p3, 12, 13, 14, 15, 16)                                \
        ZLNG_C_COLD(p4, 16, 17, 18, 19, 20)  ZLNG_C_COLD(p5, 20, 21, 22, 23, 24)                                \
        ZLNG_C_COLD(p6, 24, 25, 26, 27, 28)  ZLNG_C_COLD(p7, 28, 29, 30, 31, 32)                                \
        ZLNG_C_COLD(p8, 32, 33, 34, 35, 36)  ZLNG_C_COLD(p9, 36, 37, 38, 39, 40)                                \
        ZLNG_C_COLD(p10, 40, 41, 42, 43, 44) ZLNG_C_COLD(p11, 44, 45, 46, 47, 48)                               \
        ZLNG_C_COLD(p12, 48, 49, 50, 51, 52) ZLNG_C_COLD(p13, 52, 53, 54, 55, 56)                               \
        ZLNG_C_COLD(p14, 56, 57, 58, 59, 60) ZLNG_C_COLD(p15, 60, 61, 62, 63, 64)                               \
        "9:\n\t"                                                                                                \
        "s_mov_b64 exec, -1\n\t"
#define ZLNG_C_LITS(A0, A1, A2, A3)                                                                             \
          [p0] "v"(A0[0]), [p1] "v"(A0[1]), [p2] "v"(A0[2]), [p3] "v"(A0[3]), [p4] "v"(A1[0]), [p5] "v"(A1[1]), [p6] "v"(A1[2]),   \
          [p7] "v"(A1[3]), [p8] "v"(A2[0]), [p9] "v"(A2[1]), [p10] "v"(A2[2]), [p11] "v"(A2[3]), [p12] "v"(A3[0]), [p13] "v"(A3[1]), \
          [p14] "v"(A3[2]), [p15] "v"(A3[3])
// One full tile from its first literal; the NEXT tile's literals are requested at the top (four wave-uniform 16-byte loads: every
// lane gets the same bytes) and awaited at the end.
#define ZLNG_C_TILE(A0, A1, A2, A3, N0, N1, N2, N3)                                                             \
    asm volatile(                                                                                               \
        "global_load_dwordx4 %[n0], %[vz], %[ptr] offset:64\n\t"                                                \
        "global_load_dwordx4 %[n1], %[vz], %[ptr] offset:80\n\t"                                                \
        "global_load_dwordx4 %[n2], %[vz], %[ptr] offset:96\n\t"                                                \
        "global_load_dwordx4 %[n3], %[vz], %[ptr] offset:112\n\t"                                               \
        "s_mov_b64 exec, -2\n\t"                                                                                \
        "s_cmp_eq_u32 0, 0\n\t"                                                                                 \
        ZLNG_C_BODY                                                                                             \
        "s_waitcnt vmcnt(0)"                                                                                    \
        : [tf] "+v"(tf), [d] "=&s"(d_), [da] "=&s"(da_), [lv] "=&s"(lv_), [n0] "=&v"(N0), [n1] "=&v"(N1), [n2] "=&v"(N2), [n3] "=&v"(N3) \
        : [ptr] "s"(tile_ptr), [vz] "v"(vzero), [allow] "s"(allow_x), [pad1] "s"(pad1), [pad2] "s"(pad2), ZLNG_C_LITS(A0, A1, A2, A3) \
        : "vcc", "scc", "s98", "s99")
// The same steps entered at step `ent` (1..63) through s_getpc_b64 + a table of branches (what follows a slow_step); no loads.
#define ZLNG_C_TILE_RE(A0, A1, A2, A3, ENT)                                                                     \
    asm volatile(                                                                                               \
        "s_mov_b64 exec, -2\n\t"                                                                                \
        "s_getpc_b64 s[98:99]\n\t"                                                                              \
        "s_lshl_b32 %[d], %[ent], 2\n\t"                                                                        \
        "s_add_u32 %[d], %[d], 24\n\t"            /* the six 4-byte instructions from here to the table */      \
        "s_add_u32 s98, s98, %[d]\n\t"                                                                          \
        "s_addc_u32 s99, s99, 0\n\t"                                                                            \
        "s_cmp_eq_u32 0, 0\n\t"                                                                                 \
        "s_setpc_b64 s[98:99]\n\t"                                                                              \
        "s_branch 20f\n\t" "s_branch 21f\n\t" "s_branch 22f\n\t" "s_branch 23f\n\t" "s_branch 24f\n\t" "s_branch 25f\n\t" "s_branch 26f\n\t" "s_branch 27f\n\t" "s_branch 28f\n\t" "s_branch 29f\n\t" "s_branch 210f\n\t" "s_branch 211f\n\t" "s_branch 212f\n\t" "s_branch 213f\n\t" "s_branch 214f\n\t" "s_branch 215f\n\t" "s_branch 216f\n\t" "s_branch 217f\n\t" "s_branch 218f\n\t" "s_branch 219f\n\t" "s_branch 220f\n\t" "s_branch 221f\n\t" "s_branch 222f\n\t" "s_branch 223f\n\t" "s_branch 224f\n\t" "s_branch 225f\n\t" "s_branch 226f\n\t" "s_branch 227f\n\t" "s_branch 228f\n\t" "s_branch 229f\n\t" "s_branch 230f\n\t" "s_branch 231f\n\t" "s_branch 232f\n\t" "s_branch 233f\n\t" "s_branch 234f\n\t" "s_branch 235f\n\t" "s_branch 236f\n\t" "s_branch 237f\n\t" "s_branch 238f\n\t" "s_branch 239f\n\t" "s_branch 240f\n\t" "s_branch 241f\n\t" "s_branch 242f\n\t" "s_branch 243f\n\t" "s_branch 244f\n\t" "s_branch 245f\n\t" "s_branch 246f\n\t" "s_branch 247f\n\t" "s_branch 248f\n\t" "s_branch 249f\n\t" "s_branch 250f\n\t" "s_branch 251f\n\t" "s_branch 252f\n\t" "s_branch 253f\n\t" "s_branch 254f\n\t" "s_branch 255f\n\t" "s_branch 256f\n\t" "s_branch 257f\n\t" "s_branch 258f\n\t" "s_branch 259f\n\t" "s_branch 260f\n\t" "s_branch 261f\n\t" "s_branch 262f\n\t" "s_branch 263f\n\t" \
        ZLNG_C_BODY                                                                                             \
        : [tf] "+v"(tf), [d] "=&s"(d_), [da] "=&s"(da_), [lv] "=&s"(lv_)                                        \
        : [ent] "s"(ENT), [allow] "s"(allow_x), [pad1] "s"(pad1), [pad2] "s"(pad2), ZLNG_C_LITS(A0, A1, A2, A3)  \
        : "vcc", "scc", "s98", "s99")

typedef uint32_t LitQuad __attribute__((ext_vector_type(4)));

__global__ __launch_bounds__(64) void k_mtf_chain(MtfArgs a) {
    const uint32_t ctx = blockIdx.x;
    const uint32_t lane = threadIdx.x;
    const uint32_t n = a.ctx_total[ctx];
    uint8_t* tile_kk = a.tile_kk + (a.ctx_off[ctx] >> 6);
    if (a.skip && a.skip[ctx]) {                                       // ranked elsewhere: only tell the replay to keep off
        for (uint32_t t = lane; t < ((n + 63u) >> 6); t += 64) tile_kk[t] = 0;
        return;
    }
    if (n == 0) {
        if (a.dbg && lane == 0) { a.dbg[ctx] = 0; a.dbg[256 + ctx] = 0; }
        return;
    }
    // The stage is one dependent chain per wavefront; whatever else is resident on its SIMD (another context's parser waves, when a
    // range goes through several contexts) has slack by design.  Highest issue priority for the chain.
    if (a.prio) __builtin_amdgcn_s_setprio(3);
    uint8_t* st = a.state + ctx * 256;
    const int fpos = chain_pos((int)lane);                              // (folded per lane: a 64-entry constant table)
    const uint32_t padv = 0x100u | lane;                               // what lane 0 and the pads hold: no byte equals it
    // positions 60..63 live in the four lanes the front leaves free (0, 38, 55, 63) of a register of their own, so that ONE byte
    // per lane -- tf's or tx's -- is the table's first 64 positions (snapshots: one select + one store)
    const bool frontlane = fpos >= 0;
    const uint32_t cpos = frontlane ? (uint32_t)fpos : (lane == 0 ? 60u : lane == 38 ? 61u : lane == 55 ? 62u : 63u);
    const uint64_t padmask = (1ull << 0) | (1ull << 38) | (1ull << 55) | (1ull << 63);
    uint32_t tf = frontlane ? (uint32_t)st[cpos] : padv;
    uint32_t tx = frontlane ? padv : (uint32_t)st[cpos];
    uint32_t t1 = st[64 + lane], t2 = st[128 + lane], t3 = st[192 + lane];
    const uint64_t tstart = a.dbg ? __builtin_readcyclecounter() : 0;   // ZLNG_PROFILE=1 (scripts/ctx_probe.py): cycles and slow steps per context
    uint64_t n_ev = 0;

    // A literal that is not in the front.  Positions 60..63 (tx lanes 0 / 38 / 55 / 63) swap with the front positions 57 / 57 / 58 /
    // 59 (lanes 54 / 54 / 37 / 62); positions 64..67 (t1 lanes 0..3) swap with 60..63; from 68 on a position and its partner both lie
    // in t1..t3 (mtfnext[68] = 64): straight-line, the position from three compares, d = table[next] by two v_readlane and a select,
    // the two stores as v_cndmask under one-hot lane masks that are zero for the registers a position is not in.
    auto slow_step = [&](uint32_t c) __attribute__((always_inline)) {
        const uint64_t mx = __builtin_amdgcn_ballot_w64(tx == c);
        if (mx) {
            const uint32_t l = (uint32_t)__builtin_ctzll(mx);          // 0 / 38 / 55 / 63 = position 60 / 61 / 62 / 63
            const uint32_t fl = l <= 38u ? 54u : (l == 55u ? 37u : 62u);
            const uint32_t d = rdl(tf, fl);
            wrl(tx, d, l);
            wrl(tf, c, fl);
            return;
        }
        const uint64_t ml = __builtin_amdgcn_ballot_w64(t1 == c) & 15ull;
        if (ml) {
            const uint32_t k = (uint32_t)__builtin_ctzll(ml);          // position 64 + k, partner 60 + k
            const uint32_t xl = k == 0u ? 0u : (k == 1u ? 38u : (k == 2u ? 55u : 63u));
            const uint32_t d = rdl(tx, xl);
            wrl(t1, d, k);
            wrl(tx, c, xl);
            return;
        }
        uint64_t ma, mb, mc, oh;
        uint32_t i, nx, a_, b_, d, d1, vd;
        asm volatile(
            "v_cmp_eq_u32_e64 %[ma], %[c], %[t1]\n\t"
            "v_cmp_eq_u32_e64 %[mb], %[c], %[t2]\n\t"
            "v_cmp_eq_u32_e64 %[mc], %[c], %[t3]\n\t"
            "s_ff1_i32_b64 %[a], %[ma]\n\t"
            "s_ff1_i32_b64 %[b], %[mb]\n\t"
            "s_ff1_i32_b64 %[i], %[mc]\n\t"
            "s_add_u32 %[i], %[i], 0xc0\n\t"
            "s_add_u32 %[nx], %[b], 0x80\n\t"
            "s_cmp_lt_i32 %[b], 0\n\t"
            "s_cselect_b32 %[i], %[i], %[nx]\n\t"
            "s_add_u32 %[nx], %[a], 64\n\t"
            "s_cmp_lt_i32 %[a], 0\n\t"
            "s_cselect_b32 %[i], %[i], %[nx]\n\t"
            "s_mov_b32 %[a], 0xf337\n\t"                      /* mtf_next_fast */
            "s_mov_b32 %[b], 0x8ccf\n\t"
            "s_cmpk_lt_u32 %[i], 0x80\n\t"
            "s_cselect_b32 %[a], %[a], %[b]\n\t"
            "s_mul_i32 %[nx], %[i], %[a]\n\t"
            "s_lshr_b32 %[nx], %[nx], 16\n\t"
            "v_readlane_b32 %[d1], %[t1], %[nx]\n\t"          /* lane select = nx & 63; nx is 64..140 here */
            "v_readlane_b32 %[d], %[t2], %[nx]\n\t"
            "s_cmpk_lt_u32 %[nx], 0x80\n\t"
            "s_cselect_b32 %[d], %[d1], %[d]\n\t"
            "v_mov_b32 %[vd], %[d]\n\t"                       /* table[i] = d */
            "s_lshl_b64 %[oh], 1, %[i]\n\t"
            "s_lshr_b32 %[a], %[i], 6\n\t"
            "s_cmp_eq_u32 %[a], 1\n\t"
            "s_cselect_b64 %[ma], %[oh], 0\n\t"
            "s_cmp_eq_u32 %[a], 2\n\t"
            "s_cselect_b64 %[mb], %[oh], 0\n\t"
            "s_cmp_eq_u32 %[a], 3\n\t"
            "s_cselect_b64 %[mc], %[oh], 0\n\t"
            "v_cndmask_b32_e64 %[t1], %[t1], %[vd], %[ma]\n\t"
            "v_cndmask_b32_e64 %[t2], %[t2], %[vd], %[mb]\n\t"
            "v_cndmask_b32_e64 %[t3], %[t3], %[vd], %[mc]\n\t"
            "v_mov_b32 %[vd], %[c]\n\t"                       /* table[next] = c */
            "s_lshl_b64 %[oh], 1, %[nx]\n\t"
            "s_lshr_b32 %[a], %[nx], 6\n\t"
            "s_cmp_eq_u32 %[a], 1\n\t"
            "s_cselect_b64 %[mb], %[oh], 0\n\t"
            "s_cmp_eq_u32 %[a], 2\n\t"
            "s_cselect_b64 %[mc], %[oh], 0\n\t"
            "v_cndmask_b32_e64 %[t1], %[t1], %[vd], %[mb]\n\t"
            "v_cndmask_b32_e64 %[t2], %[t2], %[vd], %[mc]\n\t"
            : [t1] "+v"(t1), [t2] "+v"(t2), [t3] "+v"(t3), [ma] "=&s"(ma), [mb] "=&s"(mb), [mc] "=&s"(mc), [oh] "=&s"(oh),
              [i] "=&s"(i), [nx] "=&s"(nx), [a] "=&s"(a_), [b] "=&s"(b_), [d] "=&s"(d), [d1] "=&s"(d1), [vd] "=&v"(vd)
            : [c] "s"(c)
            : "scc");
    };
    // Any literal, one at a time (the run's last, partial tile): position from a ballot, partner through the layout.
    auto generic_step = [&](uint32_t c) __attribute__((always_inline)) {
        const uint64_t m = __builtin_amdgcn_ballot_w64(tf == c);
        if (!m) { slow_step(c); return; }
        const uint32_t l = (uint32_t)__builtin_ctzll(m);
        const uint32_t p = rdl((uint32_t)fpos, l);
        if (p == 0) return;
        const uint32_t pl = (p == 21u) ? 20u : (p == 41u ? 30u : l - 1u);
        const uint32_t d = rdl(tf, pl);
        wrl(tf, d, l);
        wrl(tf, c, pl);
    };

    const uint64_t allow_x = chain_allow_x();
    const uint32_t pad1 = 0x100u | 38u, pad2 = 0x100u | 55u;
    const uint32_t vzero = 0;
    uint8_t* run = a.lit_byte + a.ctx_off[ctx];                       // 64-byte aligned (k_ctx_offsets)
    uint8_t* snap = a.snap + (size_t)a.ctx_off[ctx] * 4;              // 256 B per tile of 64 literals
    // table at the start of a tile, in position order, for k_mtf_replay
#define ZLNG_C_SNAP(BASE)                                                                                          \
    {                                                                                                              \
        uint8_t* sp = snap + (size_t)(BASE) * 4;                                                                   \
        sp[cpos] = (uint8_t)sel_k(tf, tx, padmask);                                                                \
        sp[64 + lane] = (uint8_t)t1; sp[128 + lane] = (uint8_t)t2; sp[192 + lane] = (uint8_t)t3;                   \
    }
    LitQuad a0, a1, a2, a3, b0, b1, b2, b3;
    asm volatile("global_load_dwordx4 %0, %4, %5\n\tglobal_load_dwordx4 %1, %4, %5 offset:16\n\tglobal_load_dwordx4 %2, %4, %5 offset:32\n\t"
                 "global_load_dwordx4 %3, %4, %5 offset:48\n\ts_waitcnt vmcnt(0)"
                 : "=&v"(a0), "=&v"(a1), "=&v"(a2), "=&v"(a3) : "v"(vzero), "s"(run));                    // first tile
    // One full tile: consumes the literals in A*, leaves the next tile's in N*.
#define ZLNG_C_FULL_TILE(BASE, A0, A1, A2, A3, N0, N1, N2, N3)                                                     \
    {                                                                                                              \
        const uint8_t* tile_ptr = run + (BASE);                                                                    \
        uint32_t d_, da_, lv_;                                                                                     \
        ZLNG_C_SNAP(BASE)                                                                                          \
        ZLNG_C_TILE(A0, A1, A2, A3, N0, N1, N2, N3);                                                               \
        while (__builtin_expect(lv_ != 0, 0)) {     /* literal lv_ - 1 (d_) is not in the front: the statement stopped behind it */ \
            n_ev++;                                                                                                \
            slow_step(d_);                                                                                         \
            if (lv_ == 64u) break;                                                                                 \
            const uint32_t ent = lv_;                                                                              \
            ZLNG_C_TILE_RE(A0, A1, A2, A3, ent);                                                                   \
        }                                                                                                          \
    }
    uint32_t base = 0;
    for (; base + 128 <= n; base += 128) {           // two tiles per turn: the literal registers ping-pong, no copy
        ZLNG_C_FULL_TILE(base, a0, a1, a2, a3, b0, b1, b2, b3)
        ZLNG_C_FULL_TILE(base + 64, b0, b1, b2, b3, a0, a1, a2, a3)
    }
    if (base + 64 <= n) {
        ZLNG_C_FULL_TILE(base, a0, a1, a2, a3, b0, b1, b2, b3)
        base += 64;
    }
    if (base < n) {                                  // the run's last, partial tile: the replay ranks its cnt literals, the table moves on here
        const uint32_t cnt = n - base;
        ZLNG_C_SNAP(base)
        const uint32_t v = lane < cnt ? (uint32_t)run[base + lane] : 0u;
        for (uint32_t k = 0; k < cnt; k++) generic_step(rdl(v, k));
        if (lane == 0) tile_kk[base >> 6] = (uint8_t)cnt;
    }
#undef ZLNG_C_FULL_TILE
#undef ZLNG_C_SNAP
    st[cpos] = (uint8_t)sel_k(tf, tx, padmask);
    st[64 + lane] = (uint8_t)t1; st[128 + lane] = (uint8_t)t2; st[192 + lane] = (uint8_t)t3;
    if (a.dbg && lane == 0) { a.dbg[ctx] = __builtin_readcyclecounter() - tstart; a.dbg[256 + ctx] = n_ev; }
}

// ------------------------------------------------------------------------------ K2d'
// Ranks of every tile from the table the chain left at its start: ONE LANE per tile walks the tile's literals with the reference's
// own data structure (table[256] + index[256], src/libzling_lz.cpp:106-117) in LDS -- 64 tiles per wavefront, every tile of the
// stream at once.  A lane's arrays sit kReplayStride bytes apart (an odd number of dwords, so that lanes which touch the same
// position -- they mostly do: small ranks -- fall into different banks).
constexpr uint32_t kReplayStride = 256 + 256 + 64 + 4;                 // table, index, literals / ranks; 145 dwords
__global__ __launch_bounds__(64) void k_mtf_replay(MtfArgs a) {
    __shared__ __attribute__((aligned(16))) uint8_t lds[64 * kReplayStride];
    const uint32_t lane = threadIdx.x;
    const size_t tile = (size_t)blockIdx.x * 64 + lane;
    const size_t end = (size_t)a.ctx_off[255] + (((size_t)a.ctx_total[255] + 63) & ~(size_t)63);
    if (tile * 64 >= end) return;
    const uint32_t kk = a.tile_kk[tile];
    if (kk == 0) return;
    uint8_t* tab = lds + lane * kReplayStride;
    uint8_t* idx = tab + 256;
    uint8_t* lit = tab + 512;
    const uint32_t* sp = reinterpret_cast<const uint32_t*>(a.snap + tile * 256);
    for (uint32_t w = 0; w < 64; w++) {
        const uint32_t v = sp[w];
        *reinterpret_cast<uint32_t*>(tab + 4 * w) = v;
        idx[v & 255u] = (uint8_t)(4 * w); idx[(v >> 8) & 255u] = (uint8_t)(4 * w + 1);
        idx[(v >> 16) & 255u] = (uint8_t)(4 * w + 2); idx[v >> 24] = (uint8_t)(4 * w + 3);
    }
    uint32_t* run = reinterpret_cast<uint32_t*>(a.lit_byte + tile * 64);
    for (uint32_t w = 0; w < 16; w++) *reinterpret_cast<uint32_t*>(lit + 4 * w) = run[w];
    for (uint32_t k = 0; k < kk; k++) {
        const uint32_t c = lit[k], i = idx[c], nx = mtf_next_fast(i), d = tab[nx];
        tab[nx] = (uint8_t)c; tab[i] = (uint8_t)d; idx[c] = (uint8_t)nx; idx[d] = (uint8_t)i;
        lit[k] = (uint8_t)i;
    }
    if (kk == 64) { for (uint32_t w = 0; w < 16; w++) run[w] = *reinterpret_cast<uint32_t*>(lit + 4 * w); }
    else { uint8_t* rb = a.lit_byte + tile * 64; for (uint32_t k = 0; k < kk; k++) rb[k] = lit[k]; }
}

// The stage in three launches, so the host can time the serial chain (the kernel the roofline line is about) by itself.
void launch_lit_partition(const MtfArgs& a, hipStream_t s) {
    const dim3 tiles((unsigned)(a.tok_cap / kLitTile), a.nblocks);
    (void)hipMemsetAsync(a.tile_kk, 64, ((size_t)a.nblocks * a.tok_cap + 256 * 64) / 64, s);   // "the replay ranks the whole tile"
    hipLaunchKernelGGL(k_lit_tile_base, dim3(1), dim3(64), 0, s, a);
    hipLaunchKernelGGL(k_lit_tiles<kModeHist>, tiles, dim3(64), 0, s, a);
    hipLaunchKernelGGL(k_lit_scan, dim3(256), dim3(256), 0, s, a);
    hipLaunchKernelGGL(k_ctx_offsets, dim3(1), dim3(64), 0, s, a);
    hipLaunchKernelGGL(k_lit_tiles<kModeScatter>, tiles, dim3(64), 0, s, a);
}
void launch_mtf_chain(const MtfArgs& a, hipStream_t s) { hipLaunchKernelGGL(k_mtf_chain, dim3(256), dim3(64), 0, s, a); }
void launch_mtf_finish(const MtfArgs& a, hipStream_t s) {
    const dim3 tiles((unsigned)(a.tok_cap / kLitTile), a.nblocks);
    const size_t max_tiles = ((size_t)a.nblocks * a.tok_cap + 256 * 64) / 64;
    hipLaunchKernelGGL(k_mtf_replay, dim3((unsigned)((max_tiles + 63) / 64)), dim3(64), 0, s, a);
    hipLaunchKernelGGL(k_lit_tiles<kModeGather>, tiles, dim3(64), 0, s, a);
}

}  // namespace zlng
