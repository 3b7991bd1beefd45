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
// M0 is a reserved register: an inline-asm clobber of it cannot be honoured by the compiler (clang warns that it "may lead to
// undefined behaviour").  Settled on the ISA of this file's kernels (round 3, `hipcc -S`: every occurrence of m0 in mtf_rank.s is
// one of the s_mov_b32 m0 / v_writelane ..., m0 pairs written here): no compiler-generated instruction reads or writes M0 -- gfx9
// DS operations do not need it, nothing here uses s_movrel / GDS / s_sendmsg -- so nothing is live in M0 across these statements.
// The clobber stays as the statement of intent.  (Saving M0 to a scratch SGPR and putting it back inside the statements was tried:
// correct, and +5 % on the whole chain -- 834 vs 792 ms per GiB on one box, same call -- for no observable fault; not kept.)
__device__ __forceinline__ void wrl(uint32_t& v, uint32_t val, uint32_t lane) {
    asm volatile("s_mov_b32 m0, %2\n\tv_writelane_b32 %0, %1, m0" : "+v"(v) : "s"(val), "s"(lane) : "m0");
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
// gfx950 issues ~2.2 ns per independent and ~3.5 ns per dependent instruction, a not-taken branch costs ~6 ns next
// to its SCC producer, and alternating VALU/SALU is slower than grouping them; the shipped order above measures
// 32 ns/step there, this one 20.6 ns):
//   * the table chain never leaves the vector unit: up[l] = t0[l+1] (DPP wave_shl:1), "c sits one lane up" is
//     v_cmp_eq(c, up) instead of s_not + s_lshr, and the value that lane takes is up itself, so c is only ever a
//     scalar operand (no v_mov).  up's lane 63 is never written (no lane 64 to read): it keeps 0xFFFFFFFF, which
//     no literal equals;
//   * "hit in lanes 0..20" is ONE s_andn2 of vcc with the lane mask (SCC), and its one-hot low word is what the
//     step records (v_writelane) -- the rank is decoded for all 64 lanes at once after the tile (v_ffbl);
//   * the branch on that SCC is taken one step LATE, after the next step's three read-only instructions and before
//     its two selects, so it never waits for its producer.  The out-of-line part of step K therefore finds t0 as
//     step K left it, repairs it, and re-enters step K+1 at its top;
//   * eight literals are fetched by eight back-to-back v_readlane in front of their steps, into register sets A/B
//     alternately (a late out-of-line part still needs the previous group's literal).
// Everything is one asm statement: an asm goto with outputs crashes this compiler's instruction selection, and SCC
// cannot be carried between statements.  s[98:99] is the scratch pair whose low half v_writelane reads (an inline
// asm operand cannot name half of a 64-bit operand).
// DPP hazard (gfx9: VALU write -> DPP read of the same VGPR needs 2 wait states): t0's last VALU writer is the
// previous step's second select, followed by s_andn2 / v_writelane(ranks); the out-of-line part ends in
// v_writelane(ranks) / s_cmp / s_branch.
// The 64 literals of a tile are never unpacked: the tile is sixteen SGPRs (one s_load_dwordx16, issued for the
// NEXT tile at the top of the statement and awaited at its end) and each step compares byte K&3 of register K>>2
// through an SDWA scalar source -- no per-literal fetch instruction at all (19.1 ns/step in the microbenchmark).
#define ZLNG_MTF_G_STEP(PK, B, K, KP)                                                                           \
    "2" #K ":\n\t"                                                                                              \
    "v_mov_b32_dpp %[up], %[t0] wave_shl:1 row_mask:0xf bank_mask:0xf\n\t"                                      \
    "v_cmp_ne_u32_sdwa vcc, %[" #PK "], %[t0] src0_sel:BYTE_" #B " src1_sel:DWORD\n\t"                          \
    "v_cmp_eq_u32_sdwa %[m1], %[" #PK "], %[up] src0_sel:BYTE_" #B " src1_sel:DWORD\n\t"                        \
    "s_cbranch_scc0 1" #KP "f\n\t"                                                                              \
    "v_cndmask_b32_dpp %[t0], %[t0], %[t0], vcc wave_shr:1 row_mask:0xf bank_mask:0xf\n\t"                      \
    "v_cndmask_b32_e64 %[t0], %[t0], %[up], %[m1]\n\t"                                                          \
    "s_andn2_b64 s[98:99], 0x1fffff, vcc\n\t"                                                                   \
    "v_writelane_b32 %[ranks], s98, " #K "\n\t"
// Out-of-line part of step K (all 64 sit behind the steps; entered from inside step KN = K + 1): c was not in
// lanes 0..20.
//   rank 21..63: the step has already swapped c with its left neighbour, so c is at lane rank - 1; from that state
//     the reference's swap(table[rank], table[mtfnext[rank]]) is
//         t0[rank-1] = t0[rank];  t0[rank] = t0[next];  t0[next] = c
//     (next <= rank - 2 from rank 21 on, so t0[next] is untouched); mtfnext = (rank * 62263) >> 16 below 128.
//     The rank is recorded as 0x80000000 | rank.
//   rank >= 64 (c not in t0, nothing was changed): leave the statement with lv = K + 1; lane K of ranks holds 0.
// v_readlane / v_writelane lane selects come from SALU results or M0 (no wait states owed); both reads of t0
// happen before its first write (next <= rank - 2, so the three lanes are distinct).  VCC is free here: step KN
// recomputes it.  SCC is set again before re-entering step KN, whose late branch is evaluated a second time.
// (its code is ZLNG_MTF_R_SLOW below: the round-1 form of the whole tile, ZLNG_MTF_TILE, which started at step 0 only and loaded
// the next tile itself, is gone -- only the re-entrant form is used.)
// four steps = one literal register: Z = the step before A
#define ZLNG_MTF_G_FAST(PK, Z, A, B, C, D)                                                                      \
    ZLNG_MTF_G_STEP(PK, 0, A, Z) ZLNG_MTF_G_STEP(PK, 1, B, A) ZLNG_MTF_G_STEP(PK, 2, C, B) ZLNG_MTF_G_STEP(PK, 3, D, C)

// ---- re-entrant recording tile: the recording steps (ZLNG_MTF_G_STEP), entered at step `ent` through a table of branches, without the
// next tile's load.  It is what finishes a tile of the chain after a literal of rank >= 64 made the state-only form below
// leave: the literal is dealt with outside (slow_step), then the rest of the tile runs here at the speed of the recording
// step instead of the compiler-scheduled ZLNG_MTF_STEP loop (source text: 85 % of the blank's tiles meet such a literal,
// 3.5 of them per tile -- scripts/ctx_probe.py --real).  Like ZLNG_MTF_S_SLOW its out-of-line part notes K + 1 in lv first,
// so a further rank >= 64 is identified on the way out (label 9); lv = 0 means the tile is finished.  SCC is set before the
// jump (entering a step means "no slow path pending"); the seven SALU instructions of the jump also keep the DPP read of t0
// clear of a v_writelane that slow_step may have issued last.
#define ZLNG_MTF_R_SLOW(PK, B, K, KN)                                                                           \
    "1" #K ":\n\t"                                                                                              \
    "s_mov_b32 %[lv], " #K "+1\n\t"                                                                             \
    "v_cmp_eq_u32_sdwa vcc, %[" #PK "], %[t0] src0_sel:BYTE_" #B " src1_sel:DWORD\n\t"                          \
    "s_bfe_u32 %[d], %[" #PK "], (8 * " #B ") | (8 << 16)\n\t"                                                  \
    "s_cbranch_vccz 9f\n\t"                                                                                     \
    "s_ff1_i32_b64 %[nx], vcc\n\t"                                                                              \
    "s_add_u32 %[i], %[nx], 1\n\t"                                                                              \
    "s_mov_b32 m0, %[nx]\n\t"                                                                                   \
    "s_mul_i32 %[nx], %[i], 0xf337\n\t"                                                                         \
    "v_readlane_b32 %[da], %[t0], %[i]\n\t"                                                                     \
    "s_lshr_b32 %[nx], %[nx], 16\n\t"                                                                           \
    "v_readlane_b32 %[db], %[t0], %[nx]\n\t"                                                                    \
    "v_writelane_b32 %[t0], %[da], m0\n\t"                                                                      \
    "s_mov_b32 m0, %[i]\n\t"                                                                                    \
    "v_writelane_b32 %[t0], %[db], m0\n\t"                                                                      \
    "s_mov_b32 m0, %[nx]\n\t"                                                                                   \
    "v_writelane_b32 %[t0], %[d], m0\n\t"                                                                       \
    "s_bitset1_b32 %[i], 31\n\t"                                                                                \
    "v_writelane_b32 %[ranks], %[i], " #K "\n\t"                                                                \
    "s_cmp_eq_u32 0, 0\n\t"                                                                                     \
    "s_branch 2" #KN "b\n\t"
#define ZLNG_MTF_R_COLD(PK, A, B, C, D, N)                                                                      \
    ZLNG_MTF_R_SLOW(PK, 0, A, B) ZLNG_MTF_R_SLOW(PK, 1, B, C) ZLNG_MTF_R_SLOW(PK, 2, C, D) ZLNG_MTF_R_SLOW(PK, 3, D, N)
#define ZLNG_MTF_TILE_RE(PKIN, ENT)                                                                             \
    asm volatile(                                                                                               \
        "s_getpc_b64 s[98:99]\n\t"                                                                              \
        "s_lshl_b32 %[i], %[ent], 2\n\t"                                                                        \
        "s_add_u32 %[i], %[i], 24\n\t"            /* the six 4-byte instructions from here to the table */      \
        "s_add_u32 s98, s98, %[i]\n\t"                                                                          \
        "s_addc_u32 s99, s99, 0\n\t"                                                                            \
        "s_cmp_eq_u32 0, 0\n\t"                                                                                 \
        "s_setpc_b64 s[98:99]\n\t"                                                                              \
        "s_branch 20f\n\t" "s_branch 21f\n\t" "s_branch 22f\n\t" "s_branch 23f\n\t" "s_branch 24f\n\t" "s_branch 25f\n\t" "s_branch 26f\n\t" "s_branch 27f\n\t" "s_branch 28f\n\t" "s_branch 29f\n\t" "s_branch 210f\n\t" "s_branch 211f\n\t" "s_branch 212f\n\t" "s_branch 213f\n\t" "s_branch 214f\n\t" "s_branch 215f\n\t" "s_branch 216f\n\t" "s_branch 217f\n\t" "s_branch 218f\n\t" "s_branch 219f\n\t" "s_branch 220f\n\t" "s_branch 221f\n\t" "s_branch 222f\n\t" "s_branch 223f\n\t" "s_branch 224f\n\t" "s_branch 225f\n\t" "s_branch 226f\n\t" "s_branch 227f\n\t" "s_branch 228f\n\t" "s_branch 229f\n\t" "s_branch 230f\n\t" "s_branch 231f\n\t" "s_branch 232f\n\t" "s_branch 233f\n\t" "s_branch 234f\n\t" "s_branch 235f\n\t" "s_branch 236f\n\t" "s_branch 237f\n\t" "s_branch 238f\n\t" "s_branch 239f\n\t" "s_branch 240f\n\t" "s_branch 241f\n\t" "s_branch 242f\n\t" "s_branch 243f\n\t" "s_branch 244f\n\t" "s_branch 245f\n\t" "s_branch 246f\n\t" "s_branch 247f\n\t" "s_branch 248f\n\t" "s_branch 249f\n\t" "s_branch 250f\n\t" "s_branch 251f\n\t" "s_branch 252f\n\t" "s_branch 253f\n\t" "s_branch 254f\n\t" "s_branch 255f\n\t" "s_branch 256f\n\t" "s_branch 257f\n\t" "s_branch 258f\n\t" "s_branch 259f\n\t" "s_branch 260f\n\t" "s_branch 261f\n\t" "s_branch 262f\n\t" "s_branch 263f\n\t" \
        ZLNG_MTF_G_FAST(p0, 19, 0, 1, 2, 3)  ZLNG_MTF_G_FAST(p1, 3, 4, 5, 6, 7) \
        ZLNG_MTF_G_FAST(p2, 7, 8, 9, 10, 11)  ZLNG_MTF_G_FAST(p3, 11, 12, 13, 14, 15) \
        ZLNG_MTF_G_FAST(p4, 15, 16, 17, 18, 19)  ZLNG_MTF_G_FAST(p5, 19, 20, 21, 22, 23) \
        ZLNG_MTF_G_FAST(p6, 23, 24, 25, 26, 27)  ZLNG_MTF_G_FAST(p7, 27, 28, 29, 30, 31) \
        ZLNG_MTF_G_FAST(p8, 31, 32, 33, 34, 35)  ZLNG_MTF_G_FAST(p9, 35, 36, 37, 38, 39) \
        ZLNG_MTF_G_FAST(p10, 39, 40, 41, 42, 43)  ZLNG_MTF_G_FAST(p11, 43, 44, 45, 46, 47) \
        ZLNG_MTF_G_FAST(p12, 47, 48, 49, 50, 51)  ZLNG_MTF_G_FAST(p13, 51, 52, 53, 54, 55) \
        ZLNG_MTF_G_FAST(p14, 55, 56, 57, 58, 59)  ZLNG_MTF_G_FAST(p15, 59, 60, 61, 62, 63) \
        "264:\n\t"                                                                                              \
        "s_cbranch_scc0 163f\n\t"                                                                               \
        "s_mov_b32 %[lv], 0\n\t"                                                                                \
        "s_branch 9f\n\t"                                                                                       \
        ZLNG_MTF_R_COLD(p0, 0, 1, 2, 3, 4)  ZLNG_MTF_R_COLD(p1, 4, 5, 6, 7, 8) \
        ZLNG_MTF_R_COLD(p2, 8, 9, 10, 11, 12)  ZLNG_MTF_R_COLD(p3, 12, 13, 14, 15, 16) \
        ZLNG_MTF_R_COLD(p4, 16, 17, 18, 19, 20)  ZLNG_MTF_R_COLD(p5, 20, 21, 22, 23, 24) \
        ZLNG_MTF_R_COLD(p6, 24, 25, 26, 27, 28)  ZLNG_MTF_R_COLD(p7, 28, 29, 30, 31, 32) \
        ZLNG_MTF_R_COLD(p8, 32, 33, 34, 35, 36)  ZLNG_MTF_R_COLD(p9, 36, 37, 38, 39, 40) \
        ZLNG_MTF_R_COLD(p10, 40, 41, 42, 43, 44)  ZLNG_MTF_R_COLD(p11, 44, 45, 46, 47, 48) \
        ZLNG_MTF_R_COLD(p12, 48, 49, 50, 51, 52)  ZLNG_MTF_R_COLD(p13, 52, 53, 54, 55, 56) \
        ZLNG_MTF_R_COLD(p14, 56, 57, 58, 59, 60)  ZLNG_MTF_R_COLD(p15, 60, 61, 62, 63, 64) \
        "9:\n\t"                                                                                                \
        : [t0] "+v"(t0), [up] "+v"(up), [ranks] "+v"(ranks), [m1] "=&s"(m1_), [i] "=&s"(i_), [nx] "=&s"(nx_),   \
          [d] "=&s"(d_), [da] "=&s"(da_), [db] "=&s"(db_), [lv] "=&s"(lv_)                                      \
        : [ent] "s"(ENT), [p0] "s"(PKIN[0]), [p1] "s"(PKIN[1]), [p2] "s"(PKIN[2]), [p3] "s"(PKIN[3]),           \
          [p4] "s"(PKIN[4]), [p5] "s"(PKIN[5]), [p6] "s"(PKIN[6]), [p7] "s"(PKIN[7]), [p8] "s"(PKIN[8]), [p9] "s"(PKIN[9]), \
          [p10] "s"(PKIN[10]), [p11] "s"(PKIN[11]), [p12] "s"(PKIN[12]), [p13] "s"(PKIN[13]), [p14] "s"(PKIN[14]), \
          [p15] "s"(PKIN[15])                                                                                   \
        : "vcc", "scc", "s98", "s99", "m0")

// ---- state-only form of the tile (what the hottest chains run).  The serial chain is bound by instructions per literal
// (scripts/ubench/mtfstep.hip: 19.1 ns per literal with the rank record, 17.0 without; the two SALU instructions that
// are left -- slow-path test and its late branch -- cost 6.4 ns of that, the five-instruction table chain alone 10.7),
// so the chain wavefront only carries the TABLE forward: no v_writelane of the rank.  It leaves a snapshot of t0 per
// tile instead, and k_mtf_replay recomputes the ranks of every tile from its snapshot -- thousands of tiles at once.
// Differences from ZLNG_MTF_G_STEP / _SLOW: no rank record; the out-of-line part of step K first notes K + 1 in lv, so a
// literal of rank >= 64 (leave) is identified without the rank word (lv is cleared again at label 264 on the normal way out).
// (DPP hazard, gfx9: a VALU write of t0 must be >= 2 wait states ahead of a DPP read of it.  Without the rank record only
//  s_andn2 follows the step's last select, so the plain compare of the next step goes FIRST and the DPP move second.)
#define ZLNG_MTF_S_STEP(PK, B, K, KP)                                                                           \
    "2" #K ":\n\t"                                                                                              \
    "v_cmp_ne_u32_sdwa vcc, %[" #PK "], %[t0] src0_sel:BYTE_" #B " src1_sel:DWORD\n\t"                          \
    "v_mov_b32_dpp %[up], %[t0] wave_shl:1 row_mask:0xf bank_mask:0xf\n\t"                                      \
    "v_cmp_eq_u32_sdwa %[m1], %[" #PK "], %[up] src0_sel:BYTE_" #B " src1_sel:DWORD\n\t"                        \
    "s_cbranch_scc0 1" #KP "f\n\t"                                                                              \
    "v_cndmask_b32_dpp %[t0], %[t0], %[t0], vcc wave_shr:1 row_mask:0xf bank_mask:0xf\n\t"                      \
    "v_cndmask_b32_e64 %[t0], %[t0], %[up], %[m1]\n\t"                                                          \
    "s_andn2_b64 s[98:99], 0x1fffff, vcc\n\t"
#define ZLNG_MTF_S_SLOW(PK, B, K, KN)                                                                           \
    "1" #K ":\n\t"                                                                                              \
    "s_mov_b32 %[lv], " #K "+1\n\t"                                                                             \
    "v_cmp_eq_u32_sdwa vcc, %[" #PK "], %[t0] src0_sel:BYTE_" #B " src1_sel:DWORD\n\t"                          \
    "s_bfe_u32 %[d], %[" #PK "], (8 * " #B ") | (8 << 16)\n\t"                                                  \
    "s_cbranch_vccz 9f\n\t"                                                                                     \
    "s_ff1_i32_b64 %[nx], vcc\n\t"                                                                              \
    "s_add_u32 %[i], %[nx], 1\n\t"                                                                              \
    "s_mov_b32 m0, %[nx]\n\t"                                                                                   \
    "s_mul_i32 %[nx], %[i], 0xf337\n\t"                                                                         \
    "v_readlane_b32 %[da], %[t0], %[i]\n\t"                                                                     \
    "s_lshr_b32 %[nx], %[nx], 16\n\t"                                                                           \
    "v_readlane_b32 %[db], %[t0], %[nx]\n\t"                                                                    \
    "v_writelane_b32 %[t0], %[da], m0\n\t"                                                                      \
    "s_mov_b32 m0, %[i]\n\t"                                                                                    \
    "v_writelane_b32 %[t0], %[db], m0\n\t"                                                                      \
    "s_mov_b32 m0, %[nx]\n\t"                                                                                   \
    "v_writelane_b32 %[t0], %[d], m0\n\t"                                                                       \
    "s_cmp_eq_u32 0, 0\n\t"                                                                                     \
    "s_branch 2" #KN "b\n\t"
#define ZLNG_MTF_S_FAST(PK, Z, A, B, C, D)                                                                      \
    ZLNG_MTF_S_STEP(PK, 0, A, Z) ZLNG_MTF_S_STEP(PK, 1, B, A) ZLNG_MTF_S_STEP(PK, 2, C, B) ZLNG_MTF_S_STEP(PK, 3, D, C)
#define ZLNG_MTF_S_COLD(PK, A, B, C, D, N)                                                                      \
    ZLNG_MTF_S_SLOW(PK, 0, A, B) ZLNG_MTF_S_SLOW(PK, 1, B, C) ZLNG_MTF_S_SLOW(PK, 2, C, D) ZLNG_MTF_S_SLOW(PK, 3, D, N)
#define ZLNG_MTF_TILE_S(PKIN, NXTOUT)                                                                           \
    asm volatile(                                                                                               \
        "s_load_dwordx16 %[nxt], %[ptr], 0x40\n\t"                                                              \
        "s_cmp_eq_u32 0, 0\n\t"                                                                                 \
        ZLNG_MTF_S_FAST(p0, 19, 0, 1, 2, 3)      ZLNG_MTF_S_FAST(p1, 3, 4, 5, 6, 7)                             \
        ZLNG_MTF_S_FAST(p2, 7, 8, 9, 10, 11)     ZLNG_MTF_S_FAST(p3, 11, 12, 13, 14, 15)                        \
        ZLNG_MTF_S_FAST(p4, 15, 16, 17, 18, 19)  ZLNG_MTF_S_FAST(p5, 19, 20, 21, 22, 23)                        \
        ZLNG_MTF_S_FAST(p6, 23, 24, 25, 26, 27)  ZLNG_MTF_S_FAST(p7, 27, 28, 29, 30, 31)                        \
        ZLNG_MTF_S_FAST(p8, 31, 32, 33, 34, 35)  ZLNG_MTF_S_FAST(p9, 35, 36, 37, 38, 39)                        \
        ZLNG_MTF_S_FAST(p10, 39, 40, 41, 42, 43) ZLNG_MTF_S_FAST(p11, 43, 44, 45, 46, 47)                       \
        ZLNG_MTF_S_FAST(p12, 47, 48, 49, 50, 51) ZLNG_MTF_S_FAST(p13, 51, 52, 53, 54, 55)                       \
        ZLNG_MTF_S_FAST(p14, 55, 56, 57, 58, 59) ZLNG_MTF_S_FAST(p15, 59, 60, 61, 62, 63)                       \
        "264:\n\t"                                                                                              \
        "s_cbranch_scc0 163f\n\t"                                                                               \
        "s_mov_b32 %[lv], 0\n\t"                                                                                \
        "s_branch 9f\n\t"                                                                                       \
        ZLNG_MTF_S_COLD(p0, 0, 1, 2, 3, 4)       ZLNG_MTF_S_COLD(p1, 4, 5, 6, 7, 8)                             \
        ZLNG_MTF_S_COLD(p2, 8, 9, 10, 11, 12)    ZLNG_MTF_S_COLD(p3, 12, 13, 14, 15, 16)                        \
        ZLNG_MTF_S_COLD(p4, 16, 17, 18, 19, 20)  ZLNG_MTF_S_COLD(p5, 20, 21, 22, 23, 24)                        \
        ZLNG_MTF_S_COLD(p6, 24, 25, 26, 27, 28)  ZLNG_MTF_S_COLD(p7, 28, 29, 30, 31, 32)                        \
        ZLNG_MTF_S_COLD(p8, 32, 33, 34, 35, 36)  ZLNG_MTF_S_COLD(p9, 36, 37, 38, 39, 40)                        \
        ZLNG_MTF_S_COLD(p10, 40, 41, 42, 43, 44) ZLNG_MTF_S_COLD(p11, 44, 45, 46, 47, 48)                       \
        ZLNG_MTF_S_COLD(p12, 48, 49, 50, 51, 52) ZLNG_MTF_S_COLD(p13, 52, 53, 54, 55, 56)                       \
        ZLNG_MTF_S_COLD(p14, 56, 57, 58, 59, 60) ZLNG_MTF_S_COLD(p15, 60, 61, 62, 63, 64)                       \
        "9:\n\t"                                                                                                \
        "s_waitcnt lgkmcnt(0)"                                                                                  \
        : [t0] "+v"(t0), [up] "+v"(up), [m1] "=&s"(m1_), [i] "=&s"(i_), [nx] "=&s"(nx_),                        \
          [d] "=&s"(d_), [da] "=&s"(da_), [db] "=&s"(db_), [lv] "=&s"(lv_), [nxt] "=&s"(NXTOUT)                    \
        : [ptr] "s"(tile_ptr), [p0] "s"(PKIN[0]), [p1] "s"(PKIN[1]), [p2] "s"(PKIN[2]), [p3] "s"(PKIN[3]),              \
          [p4] "s"(PKIN[4]), [p5] "s"(PKIN[5]), [p6] "s"(PKIN[6]), [p7] "s"(PKIN[7]), [p8] "s"(PKIN[8]), [p9] "s"(PKIN[9]), \
          [p10] "s"(PKIN[10]), [p11] "s"(PKIN[11]), [p12] "s"(PKIN[12]), [p13] "s"(PKIN[13]), [p14] "s"(PKIN[14]),        \
          [p15] "s"(PKIN[15])                                                                                     \
        : "vcc", "scc", "s98", "s99", "m0")

__global__ __launch_bounds__(64) void k_mtf_dense(MtfArgs a) {
    const uint32_t ctx = blockIdx.x;
    const uint32_t lane = threadIdx.x;
    if (a.skip && a.skip[ctx]) {                                       // ranked elsewhere: only tell the replay to keep off
        const uint32_t tiles = (a.ctx_total[ctx] + 63u) >> 6;
        for (uint32_t t = lane; t < tiles; t += 64) a.tile_kk[(a.ctx_off[ctx] >> 6) + t] = 0;
        return;
    }
    uint8_t* st = a.state + ctx * 256;
    uint32_t t0 = st[lane], t1 = st[64 + lane], t2 = st[128 + lane], t3 = st[192 + lane];
    const uint64_t tstart = a.dbg ? __builtin_readcyclecounter() : 0;   // ZLNG_PROFILE=1 (scripts/ctx_probe.py): cycles and rank >= 64 events per context
    uint64_t n_ev = 0;

    // rank >= 64: c is not in t0.  Straight-line (the compiler's form of the same thing -- four-way switches on the register
    // that holds a position -- came out at ~80 instructions with a dozen taken branches, ~0.4 us on a lone wavefront):
    // the position from three compares, d = table[next] by three v_readlane and two selects (next = mtfnext(i) lies in 60..140),
    // the two stores as v_cndmask under a one-hot lane mask that is zero for the registers the position is not in.
    auto slow_step = [&](uint32_t c) __attribute__((always_inline)) -> uint32_t {
        uint64_t ma, mb, mc, oh;
        uint32_t i, nx, a, b, d, d0, d1, vd;
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
            "v_readlane_b32 %[d0], %[t0], %[nx]\n\t"          /* lane select = nx & 63 */
            "v_readlane_b32 %[d1], %[t1], %[nx]\n\t"
            "v_readlane_b32 %[d], %[t2], %[nx]\n\t"
            "s_cmpk_lt_u32 %[nx], 0x80\n\t"
            "s_cselect_b32 %[d], %[d1], %[d]\n\t"
            "s_cmpk_lt_u32 %[nx], 0x40\n\t"
            "s_cselect_b32 %[d], %[d0], %[d]\n\t"
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
            "s_cmp_eq_u32 %[a], 0\n\t"
            "s_cselect_b64 %[ma], %[oh], 0\n\t"
            "s_cmp_eq_u32 %[a], 1\n\t"
            "s_cselect_b64 %[mb], %[oh], 0\n\t"
            "s_cmp_eq_u32 %[a], 2\n\t"
            "s_cselect_b64 %[mc], %[oh], 0\n\t"
            "v_cndmask_b32_e64 %[t0], %[t0], %[vd], %[ma]\n\t"
            "v_cndmask_b32_e64 %[t1], %[t1], %[vd], %[mb]\n\t"
            "v_cndmask_b32_e64 %[t2], %[t2], %[vd], %[mc]\n\t"
            : [t0] "+v"(t0), [t1] "+v"(t1), [t2] "+v"(t2), [t3] "+v"(t3), [ma] "=&s"(ma), [mb] "=&s"(mb), [mc] "=&s"(mc), [oh] "=&s"(oh),
              [i] "=&s"(i), [nx] "=&s"(nx), [a] "=&s"(a), [b] "=&s"(b), [d] "=&s"(d), [d0] "=&s"(d0), [d1] "=&s"(d1), [vd] "=&v"(vd)
            : [c] "s"(c)
            : "scc");
        return i;
    };

    uint32_t up = 0xFFFFFFFFu;                                         // the tile statements' scratch: t0 shifted down one lane
    uint8_t* run = a.lit_byte + a.ctx_off[ctx];                       // 64-byte aligned (k_ctx_offsets)
    uint8_t* snap = a.snap + a.ctx_off[ctx];
    uint8_t* tile_kk = a.tile_kk + (a.ctx_off[ctx] >> 6);
    const uint32_t n = a.ctx_total[ctx];
    typedef uint32_t Tile16 __attribute__((ext_vector_type(16)));
    Tile16 pk;
    asm volatile("s_load_dwordx16 %0, %1, 0x0\n\ts_waitcnt lgkmcnt(0)" : "=s"(pk) : "s"(run));     // first tile
    // One full tile in the state-only form: consumes the 64 literals in PKIN, leaves the next tile's in NXTOUT.  tile_kk is
    // pre-set to 64 ("the replay ranks the whole tile", launch_lit_partition) and only written when a tile differs.
#define ZLNG_MTF_FULL_TILE(BASE, PKIN, NXTOUT)                                                                     \
    {                                                                                                              \
        const uint8_t* tile_ptr = run + (BASE);                                                                    \
        uint32_t i_, nx_, d_, da_, db_, lv_;                                                                       \
        uint64_t m1_;                                                                                              \
        snap[(BASE) + lane] = (uint8_t)t0;          /* table front at the start of the tile, for k_mtf_replay */   \
        ZLNG_MTF_TILE_S(PKIN, NXTOUT);                                                                             \
        if (__builtin_expect(lv_ != 0, 0)) {        /* literal lv_ - 1 has rank >= 64: the statement stopped there */ \
            uint32_t ranks = 0;                     /* one-hot word or 0x80000000 | rank per lane, as the recording steps leave it */     \
            const uint32_t kk = lv_ - 1;                                                                           \
            uint32_t at = kk;                                                                                      \
            for (;;) {                              /* the literal itself (d_), then the rest of the tile in the recording form */ \
                n_ev++;                                                                                            \
                const uint32_t r = slow_step(d_);                                                                  \
                wrl(ranks, 0x80000000u | r, at);                                                                   \
                if (at == 63u) break;                                                                              \
                const uint32_t ent = at + 1;                                                                       \
                ZLNG_MTF_TILE_RE(PKIN, ent);                                                                       \
                if (lv_ == 0) break;                                                                               \
                at = lv_ - 1;                                                                                      \
            }                                                                                                      \
            const uint32_t rk = (ranks & 0x80000000u) ? (ranks & 0xFFu) : (uint32_t)__builtin_ctz(ranks | 0x40000000u); \
            if (lane >= kk) run[(BASE) + lane] = (uint8_t)rk;      /* lanes below kk keep their literal for the replay */ \
            if (lane == 0) tile_kk[(BASE) >> 6] = (uint8_t)kk;                                                     \
        }                                                                                                          \
    }
#define RANKSTORE(I, K) wrl(ranks, I, K)
    uint32_t base = 0;
    Tile16 pk2;
    for (; base + 128 <= n; base += 128) {           // two tiles per turn: the literal registers ping-pong, no copy
        ZLNG_MTF_FULL_TILE(base, pk, pk2)
        ZLNG_MTF_FULL_TILE(base + 64, pk2, pk)
    }
    if (base + 64 <= n) { ZLNG_MTF_FULL_TILE(base, pk, pk2) base += 64; }
    if (base < n) {                                  // the run's last, partial tile: recording form, nothing left to the replay
        uint32_t ranks = 0xFFFFFFFFu;
        const uint32_t cnt = n - base;
        const uint32_t v = lane < cnt ? (uint32_t)run[base + lane] : 0u;
        for (uint32_t k = 0; k < cnt; k++) ZLNG_MTF_STEP(k)
        if (lane < cnt) run[base + lane] = (uint8_t)ranks;
        if (lane == 0) tile_kk[base >> 6] = 0;
    }
#undef RANKSTORE
#undef ZLNG_MTF_FULL_TILE
    st[lane] = (uint8_t)t0; st[64 + lane] = (uint8_t)t1; st[128 + lane] = (uint8_t)t2; st[192 + lane] = (uint8_t)t3;
    if (a.dbg && lane == 0) { a.dbg[ctx] = __builtin_readcyclecounter() - tstart; a.dbg[256 + ctx] = n_ev; }
}

// ------------------------------------------------------------------------------ K2d'
// Ranks of the tiles k_mtf_dense walked in its state-only form: one wavefront per tile replays the tile's first
// tile_kk literals from the table front the chain left in `snap` (all of them have rank < 64, so the front is all
// the state there is) and records their ranks in place.  Tiles are independent: the whole chip works on them.
__global__ __launch_bounds__(256) void k_mtf_replay(MtfArgs a) {
    const uint32_t lane = threadIdx.x & 63;
    const size_t tile = (size_t)blockIdx.x * 4 + (threadIdx.x >> 6);
    const size_t end = (size_t)a.ctx_off[255] + (((size_t)a.ctx_total[255] + 63) & ~(size_t)63);
    if (tile * 64 >= end) return;
    const uint32_t kk = a.tile_kk[tile];
    if (kk == 0) return;
    uint32_t t0 = a.snap[tile * 64 + lane];
    uint8_t* run = a.lit_byte + tile * 64;
    const uint32_t v = run[lane];
    uint32_t ranks = 0;
    auto slow_step = [&](uint32_t) -> uint32_t { return 255u; };       // unreachable: these literals have rank < 64
#define RANKSTORE(I, K) wrl(ranks, I, K)
    for (uint32_t k = 0; k < kk; k++) ZLNG_MTF_STEP(k)
#undef RANKSTORE
    if (lane < kk) run[lane] = (uint8_t)ranks;
}

// ------------------------------------------------------------------------------ K2d, front / back form (ZLNG_MTF=front: exact, NOT the default)
// mtfnext (src/tables/gen.py:52-56) splits the table: ranks 0..20 swap with their LEFT NEIGHBOUR (mtfnext[i] = i - 1), so a literal
// whose symbol is among the first 21 entries -- the FRONT -- moves only inside the front; ranks 23..255 move only inside the BACK
// (mtfnext[23] = 21, mtfnext[i] >= 21 from there on); the two parts meet in exactly two ranks: 21 (-> 19) and 22 (-> 20).  Between
// such COUPLING literals the front and the back evolve independently of each other, and which part a literal belongs to is a
// property of its symbol (is it in the front SET, which only a coupling changes).  One wavefront per context walks a tile of 64
// literals in two interleaved passes:
//   back pass   the tile's literals are classified in parallel against the front set (a byte table in LDS); every literal outside it
//               is ranked and applied to the back in order (register table, v_readlane / v_writelane); its rank is recorded;
//   front pass  the front lives in lanes 0..20 of one VGPR whose other lanes hold values no byte equals, so the neighbour-swap step
//               needs NO test at all: a literal that is not in the front is a no-op by itself.  Five VALU instructions per literal
//               (scripts/ubench/fstep.hip: 11.7 ns against 17.1 for the tested step; the s_nop is the gfx9 wait state between a
//               VALU write and a DPP read of the same register -- measured: without it the table comes out wrong), sixteen literals
//               per asm statement straight from the scalar registers the tile was loaded into.
// A coupling literal stops the front pass at its position, exchanges the two symbols and re-classifies the rest of the tile.
// MEASURED (round 3, scripts/ctx_probe.py): the test-free step is 11.7 ns per literal, but the boundary at rank 21 lies inside the
// active zone of real tables -- 1.9 % of the blank context's literals on the benchmark text and 3 % on source text are couplings
// (80 % of all its literals outside the front), each costs a partial group in loop form on both sides (~0.5 us), and a literal
// outside the front ~0.3 us of compiler-generated v_readlane / v_writelane code: 37 ns per literal against k_mtf_dense's 19 on the
// benchmark text, 125 against 62 on source text.  Kept as a cross-check of the rank stage and as the record of the experiment.
// The chain carries only the tables forward; k_mtf_replay_front recomputes the front literals' ranks per tile from the front's
// snapshot (and the recorded back ranks: a 21 / 22 among them tells it where a symbol entered the front).
#define ZLNG_F_STEP(PK, B)                                                                                      \
    "v_cmp_ne_u32_sdwa vcc, %[" #PK "], %[tf] src0_sel:BYTE_" #B " src1_sel:DWORD\n\t"                          \
    "s_nop 0\n\t"                                                                                               \
    "v_mov_b32_dpp %[up], %[tf] wave_shl:1 row_mask:0xf bank_mask:0xf\n\t"                                      \
    "v_cmp_eq_u32_sdwa %[m1], %[" #PK "], %[up] src0_sel:BYTE_" #B " src1_sel:DWORD\n\t"                        \
    "v_cndmask_b32_dpp %[tf], %[tf], %[tf], vcc wave_shr:1 row_mask:0xf bank_mask:0xf\n\t"                      \
    "v_cndmask_b32_e64 %[tf], %[tf], %[up], %[m1]\n\t"
#define ZLNG_F_WORD(PK) ZLNG_F_STEP(PK, 0) ZLNG_F_STEP(PK, 1) ZLNG_F_STEP(PK, 2) ZLNG_F_STEP(PK, 3)
// sixteen literals = four literal registers
#define ZLNG_F_GROUP(A, B, C, D)                                                                                \
    asm volatile(ZLNG_F_WORD(pa) ZLNG_F_WORD(pb) ZLNG_F_WORD(pc) ZLNG_F_WORD(pd)                                \
                 : [tf] "+v"(tf), [up] "+v"(up), [m1] "=&s"(m1_)                                                \
                 : [pa] "s"(A), [pb] "s"(B), [pc] "s"(C), [pd] "s"(D)                                           \
                 : "vcc")
// one literal held in a scalar register
#define ZLNG_F_ONE(C)                                                                                           \
    asm volatile("v_cmp_ne_u32_e32 vcc, %[c], %[tf]\n\t"                                                        \
                 "s_nop 0\n\t"                                                                                  \
                 "v_mov_b32_dpp %[up], %[tf] wave_shl:1 row_mask:0xf bank_mask:0xf\n\t"                         \
                 "v_cmp_eq_u32_e64 %[m1], %[c], %[up]\n\t"                                                      \
                 "v_cndmask_b32_dpp %[tf], %[tf], %[tf], vcc wave_shr:1 row_mask:0xf bank_mask:0xf\n\t"         \
                 "v_cndmask_b32_e64 %[tf], %[tf], %[up], %[m1]\n\t"                                             \
                 : [tf] "+v"(tf), [up] "+v"(up), [m1] "=&s"(m1_)                                                \
                 : [c] "s"(C)                                                                                   \
                 : "vcc")

constexpr uint32_t kFront = 21;                      // table positions 0..20

__global__ __launch_bounds__(64) void k_mtf_front(MtfArgs a) {
    __shared__ uint8_t isf[256];                     // 1: the symbol is in the front
    const uint32_t ctx = blockIdx.x;
    const uint32_t lane = threadIdx.x;
    if (a.skip && a.skip[ctx]) {                                       // ranked elsewhere: only tell the replay to keep off
        const uint32_t tiles = (a.ctx_total[ctx] + 63u) >> 6;
        for (uint32_t t = lane; t < tiles; t += 64) a.tile_kk[(a.ctx_off[ctx] >> 6) + t] = 0;
        return;
    }
    uint8_t* st = a.state + ctx * 256;
    const uint32_t poison = 0x100u | lane;
    const uint32_t s0 = st[lane];
    uint32_t tf = lane < kFront ? s0 : poison;       // the front: lanes 0..20
    uint32_t t0 = lane < kFront ? poison : s0;       // the back: lanes 21..63 of t0, then t1 .. t3 (lane l of t[r] = table[64 r + l])
    uint32_t t1 = st[64 + lane], t2 = st[128 + lane], t3 = st[192 + lane];
    for (uint32_t i = lane; i < 256; i += 64) isf[i] = 0;
    __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront"); __builtin_amdgcn_wave_barrier();
    if (lane < kFront) isf[s0] = 1;
    __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront"); __builtin_amdgcn_wave_barrier();

    uint32_t up = 0xFFFFFFFFu;                       // tf shifted down one lane; lane 63 is never written
    uint64_t m1_;
    const unsigned long long tstart = __builtin_readcyclecounter();
    unsigned long long n_nf = 0, n_cp = 0;
    uint8_t* run = a.lit_byte + a.ctx_off[ctx];     // 64-byte aligned (k_ctx_offsets)
    uint8_t* nfr = a.nfr + a.ctx_off[ctx];
    uint8_t* snap = a.snap + a.ctx_off[ctx];
    uint8_t* tile_kk = a.tile_kk + (a.ctx_off[ctx] >> 6);
    const uint32_t n = a.ctx_total[ctx];
    typedef uint32_t Tile16 __attribute__((ext_vector_type(16)));

    auto back_get = [&](uint32_t i) __attribute__((always_inline)) -> uint32_t {
        switch (i >> 6) { case 0: return rdl(t0, i & 63); case 1: return rdl(t1, i & 63); case 2: return rdl(t2, i & 63); default: return rdl(t3, i & 63); }
    };
    auto back_set = [&](uint32_t i, uint32_t val) __attribute__((always_inline)) {
        switch (i >> 6) { case 0: wrl(t0, val, i & 63); break; case 1: wrl(t1, val, i & 63); break; case 2: wrl(t2, val, i & 63); break; default: wrl(t3, val, i & 63); break; }
    };

    Tile16 pk, pkn;
    asm volatile("s_load_dwordx16 %0, %1, 0x0\n\ts_waitcnt lgkmcnt(0)" : "=s"(pk) : "s"(run));          // first tile
    // The per-lane copy of a tile is loaded one tile ahead by hand: vector memory operations retire in order and stores count too, so
    // the wait for it may leave exactly the two stores issued behind it (snapshot, back ranks) in flight.  (Left to the compiler the
    // loop waits for vmcnt(0) at its head, i.e. for the previous tile's stores to be acknowledged: 1.7 us per tile.)
    uint32_t vn;
    asm volatile("global_load_ubyte %0, %1, %2" : "=v"(vn) : "v"(lane), "s"(run) : "memory");
    asm volatile("s_waitcnt vmcnt(0)" : "+v"(vn));
    for (uint32_t base = 0; base < n; base += 64) {
        const uint32_t cnt = n - base < 64u ? n - base : 64u;
        const uint32_t v = vn;                       // the tile's 64 literals, one per lane; pk holds the same bytes in sixteen SGPRs
        {
            const uint8_t* nptr = run + base + 64;   // next tile (the pools leave one tile of read-ahead behind the last run)
            asm volatile("global_load_ubyte %0, %1, %2" : "=v"(vn) : "v"(lane), "s"(nptr) : "memory");
        }
        snap[base + lane] = (uint8_t)tf;            // the front at the start of the tile, for the replay (lanes >= 21: anything)
        uint32_t nfv = 0;                            // lane k: rank of literal k if it was outside the front when it came
        const bool valid = lane < cnt;
        uint64_t nf = __ballot(valid && isf[v] == 0);
        {   // next tile's scalar copy: issued behind the classification's LDS wait, awaited at the end of the tile
            const uint8_t* nptr = run + base + 64;
            asm volatile("s_load_dwordx16 %0, %1, 0x0" : "=s"(pkn) : "s"(nptr));
        }
        uint32_t fpos = 0;
        // literals [fpos, to) through the front
        auto front_advance = [&](uint32_t to) __attribute__((always_inline)) {
            while (fpos < to) {
                if ((fpos & 15u) == 0u && fpos + 16u <= to) {
                    switch (fpos >> 4) {
                        case 0: ZLNG_F_GROUP(pk[0], pk[1], pk[2], pk[3]); break;
                        case 1: ZLNG_F_GROUP(pk[4], pk[5], pk[6], pk[7]); break;
                        case 2: ZLNG_F_GROUP(pk[8], pk[9], pk[10], pk[11]); break;
                        default: ZLNG_F_GROUP(pk[12], pk[13], pk[14], pk[15]); break;
                    }
                    fpos += 16;
                } else {
                    const uint32_t c = rdl(v, fpos);
                    ZLNG_F_ONE(c);
                    fpos++;
                }
            }
        };
        while (nf) {
            const uint32_t k = (uint32_t)__builtin_ctzll(nf);
            const uint32_t c = rdl(v, k);
            // rank = position of c in the back
            const uint64_t m0 = __ballot(t0 == c);
            uint32_t i;
            if (m0) i = (uint32_t)__builtin_ctzll(m0);
            else {
                const uint64_t b1 = __ballot(t1 == c), b2 = __ballot(t2 == c), b3 = __ballot(t3 == c);
                i = b1 ? 64 + (uint32_t)__builtin_ctzll(b1) : (b2 ? 128 + (uint32_t)__builtin_ctzll(b2) : 192 + (uint32_t)__builtin_ctzll(b3));
            }
            wrl(nfv, i, k);
            n_nf++;
            if (i <= kFront + 1) {
                n_cp++;
                // coupling (rank 21 -> 19, 22 -> 20): the front must have seen every literal before this one
                front_advance(k);
                fpos = k + 1;                        // (the coupling literal itself is not a front literal)
                const uint32_t j = i - 2;
                const uint32_t d = rdl(tf, j);
                wrl(tf, c, j);
                wrl(t0, d, i);
                if (lane == 0) { isf[c] = 1; isf[d] = 0; }
                __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront"); __builtin_amdgcn_wave_barrier();
                nf = __ballot(valid && isf[v] == 0) & ~((2ull << k) - 1ull);      // the front set changed: classify the rest again
            } else {
                const uint32_t nx = mtf_next_fast(i);                  // >= 21: stays in the back
                const uint32_t d = back_get(nx);
                back_set(i, d);
                back_set(nx, c);
                nf &= nf - 1ull;
            }
        }
        front_advance(cnt);
        asm volatile("s_waitcnt lgkmcnt(0)" : "+s"(pkn));
        pk = pkn;
        nfr[base + lane] = (uint8_t)nfv;             // (whole tiles: the run's padding takes the surplus of its last one)
        asm volatile("s_waitcnt vmcnt(2)" : "+v"(vn));
        if (cnt < 64u && lane == 0) tile_kk[base >> 6] = (uint8_t)cnt;
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    if (a.dbg && lane == 0) { a.dbg[ctx] = __builtin_readcyclecounter() - tstart; a.dbg[256 + ctx] = n_nf | n_cp << 32; }
    st[lane] = (uint8_t)(lane < kFront ? tf : t0); st[64 + lane] = (uint8_t)t1; st[128 + lane] = (uint8_t)t2; st[192 + lane] = (uint8_t)t3;
}

// Ranks of every tile from the front's snapshot at its start: literals outside the front carry their rank already (nfr != 0;
// a 21 / 22 puts the symbol into the front at 19 / 20), front literals take the recording form of the neighbour-swap step.
__global__ __launch_bounds__(256) void k_mtf_replay_front(MtfArgs a) {
    const uint32_t lane = threadIdx.x & 63;
    const size_t tile = (size_t)blockIdx.x * 4 + (threadIdx.x >> 6);
    const size_t end = (size_t)a.ctx_off[255] + (((size_t)a.ctx_total[255] + 63) & ~(size_t)63);
    if (tile * 64 >= end) return;
    const uint32_t kk = a.tile_kk[tile];
    if (kk == 0) return;
    const uint32_t sv = a.snap[tile * 64 + lane];
    uint32_t t0 = lane < kFront ? sv : (0x100u | lane);
    uint8_t* run = a.lit_byte + tile * 64;
    const uint32_t v = run[lane], nv = a.nfr[tile * 64 + lane];
    uint32_t ranks = 0;
    for (uint32_t k = 0; k < kk; k++) {
        const uint32_t c = rdl(v, k), r = rdl(nv, k);
        uint32_t i;
        if (r) {
            i = r;
            if (r <= kFront + 1) wrl(t0, c, r - 2);
        } else {
            uint64_t m0, m1_;
            uint32_t cv;
            ZLNG_MTF_FAST(k, m0);
        }
        wrl(ranks, i, k);
    }
    if (lane < kk) run[lane] = (uint8_t)ranks;
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
void launch_mtf_chain(const MtfArgs& a, hipStream_t s) {
    if (a.front_split) hipLaunchKernelGGL(k_mtf_front, dim3(256), dim3(64), 0, s, a);
    else hipLaunchKernelGGL(k_mtf_dense, dim3(256), dim3(64), 0, s, a);
}
void launch_mtf_finish(const MtfArgs& a, hipStream_t s) {
    const dim3 tiles((unsigned)(a.tok_cap / kLitTile), a.nblocks);
    const size_t max_tiles = ((size_t)a.nblocks * a.tok_cap + 256 * 64) / 64;
    if (a.front_split) hipLaunchKernelGGL(k_mtf_replay_front, dim3((unsigned)((max_tiles + 3) / 4)), dim3(256), 0, s, a);
    else hipLaunchKernelGGL(k_mtf_replay, dim3((unsigned)((max_tiles + 3) / 4)), dim3(256), 0, s, a);
    hipLaunchKernelGGL(k_lit_tiles<kModeGather>, tiles, dim3(64), 0, s, a);
}

}  // namespace zlng
