// mtf_rank.hip -- K2: the stream-serial literal rank stage.
//
// Replaces ZlingMTFEncoder::Encode (src/libzling_lz.cpp:112-117) as called from the literal
// branch of EncodeImpl (src/libzling_lz.cpp:188).  The 256 tables persist across blocks
// (they are members of the long-lived encoder, src/libzling_lz.h:105, and Reset() does not
// touch them, src/libzling_lz.cpp:197-209), so this stage is one serial chain PER CONTEXT
// over the whole stream.  The parse (K1) leaves literals raw and tags each with its context
// byte, which makes the 256 chains independent of each other: one wavefront per context.
//
// Each wavefront streams the token words of all blocks in order (coalesced 256 B tiles, several
// in flight), ballots the lanes that hold a literal of ITS context, and replays those in lane
// order against its table held in registers.
#include "zlng_common.h"
#include "zlng_kernels.h"

namespace zlng {

constexpr int kMtfTilesInFlight = 8;

// mtfnext without a division: floor(19i/20) for i < 128, floor(11i/20) otherwise (== zlng_common.h mtf_next,
// checked exhaustively below).
__host__ __device__ constexpr uint32_t mtf_next_fast(uint32_t i) { return i < 128 ? (i * 62263u) >> 16 : (i * 36047u) >> 16; }
constexpr bool mtf_next_fast_ok() {
    for (uint32_t i = 0; i < 256; i++) if (mtf_next_fast(i) != (i < 128 ? (i * 95u) / 100u : (i * 55u) / 100u)) return false;
    return true;
}
static_assert(mtf_next_fast_ok(), "mtf_next_fast must equal int(0.95 i) / int(0.55 i)");

__device__ __forceinline__ uint32_t rdl(uint32_t v, uint32_t lane) { return (uint32_t)__builtin_amdgcn_readlane((int)v, (int)lane); }

// The table of one context lives in four VGPRs: lane l of t[r] holds table[64 r + l].  A rank lookup is a
// wave-wide compare + ballot (no index[] array, no LDS round trip); the swap is two lane-predicated moves.
// This matters because one wavefront issues an instruction every ~4 cycles at best: the chain is bound by
// instruction count, and ranks < 64 (almost all literals of text) touch a single register.
__global__ __launch_bounds__(64) void k_mtf_rank(MtfArgs a) {
    const uint32_t ctx = blockIdx.x;
    const uint32_t lane = threadIdx.x;
    uint8_t* st = a.state + ctx * 256;
    uint32_t t0 = st[lane], t1 = st[64 + lane], t2 = st[128 + lane], t3 = st[192 + lane];

    auto put = [&](uint32_t pos, uint32_t val) {          // table[pos] = val   (pos, val wave-uniform)
        const bool me = lane == (pos & 63);
        switch (pos >> 6) {
            case 0: t0 = me ? val : t0; break;
            case 1: t1 = me ? val : t1; break;
            case 2: t2 = me ? val : t2; break;
            default: t3 = me ? val : t3; break;
        }
    };
    auto get = [&](uint32_t pos) -> uint32_t {
        switch (pos >> 6) {
            case 0: return rdl(t0, pos & 63);
            case 1: return rdl(t1, pos & 63);
            case 2: return rdl(t2, pos & 63);
            default: return rdl(t3, pos & 63);
        }
    };

    // Token words are streamed in batches of kMtfTilesInFlight x 64; the next batch's loads are
    // issued before the current batch is replayed, so the serial chain never waits for HBM.
    auto load_batch = [&](const uint32_t* t, uint32_t n, uint32_t base, uint32_t (&v)[kMtfTilesInFlight]) {
#pragma unroll
        for (int u = 0; u < kMtfTilesInFlight; u++) {
            const uint32_t i = base + u * 64 + lane;
            v[u] = i < n ? t[i] : 0xFFFFFFFFu;
        }
    };
    for (uint32_t blk = 0; blk < a.nblocks; blk++) {
        uint32_t* t = a.tok + (size_t)blk * kTokCap;
        const uint32_t n = a.ntok[blk];
        uint32_t v[kMtfTilesInFlight], vn[kMtfTilesInFlight];
        load_batch(t, n, 0, vn);
        for (uint32_t base = 0; base < n; base += 64 * kMtfTilesInFlight) {
#pragma unroll
            for (int u = 0; u < kMtfTilesInFlight; u++) v[u] = vn[u];
            load_batch(t, n, base + 64 * kMtfTilesInFlight, vn);
#pragma unroll
            for (int u = 0; u < kMtfTilesInFlight; u++) {
                const bool mine = (v[u] & 0xFF00u) == 0 && (v[u] >> 16) == ctx;   // sym < 256 and my context
                uint64_t mask = __ballot(mine);
                if (mask == 0) continue;
                uint32_t myrank = 0;
                while (mask) {
                    const uint32_t l = (uint32_t)__builtin_ctzll(mask);
                    mask &= mask - 1;
                    const uint32_t c = rdl(v[u], l) & 0xFF;
                    // ZlingMTFEncoder::Encode: rank = position of c; swap it with the entry at mtfnext[rank]
                    uint32_t i;
                    const uint64_t m0 = __ballot(t0 == c);
                    if (m0) {                                    // rank < 64: everything happens inside t0
                        i = (uint32_t)__builtin_ctzll(m0);
                        const uint32_t nx = (i * 62263u) >> 16;
                        const uint32_t d = rdl(t0, nx);
                        t0 = lane == i ? d : t0;
                        t0 = lane == nx ? c : t0;
                    } else {
                        const uint64_t m1 = __ballot(t1 == c), m2 = __ballot(t2 == c), m3 = __ballot(t3 == c);
                        i = m1 ? 64 + (uint32_t)__builtin_ctzll(m1)
                               : (m2 ? 128 + (uint32_t)__builtin_ctzll(m2) : 192 + (uint32_t)__builtin_ctzll(m3));
                        const uint32_t nx = mtf_next_fast(i);
                        const uint32_t d = get(nx);
                        put(i, d);
                        put(nx, c);
                    }
                    myrank = lane == l ? i : myrank;
                }
                if (mine) t[base + u * 64 + lane] = myrank | ctx << 16;
            }
        }
    }
    st[lane] = (uint8_t)t0; st[64 + lane] = (uint8_t)t1; st[128 + lane] = (uint8_t)t2; st[192 + lane] = (uint8_t)t3;
}

void launch_mtf_rank(const MtfArgs& a, hipStream_t s) {
    hipLaunchKernelGGL(k_mtf_rank, dim3(256), dim3(64), 0, s, a);
}

}  // namespace zlng
