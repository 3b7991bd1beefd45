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
// order against its table held in LDS.
#include "zlng_common.h"
#include "zlng_kernels.h"

namespace zlng {

constexpr int kMtfTilesInFlight = 8;

__global__ __launch_bounds__(64) void k_mtf_rank(MtfArgs a) {
    __shared__ uint8_t table[256];
    __shared__ uint8_t index[256];
    const uint32_t ctx = blockIdx.x;
    const uint32_t lane = threadIdx.x;
    uint8_t* st = a.state + ctx * 256;
    for (uint32_t i = lane; i < 256; i += 64) { uint8_t c = st[i]; table[i] = c; index[c] = (uint8_t)i; }
    __syncthreads();

    for (uint32_t blk = 0; blk < a.nblocks; blk++) {
        uint32_t* t = a.tok + (size_t)blk * kTokCap;
        const uint32_t n = a.ntok[blk];
        for (uint32_t base = 0; base < n; base += 64 * kMtfTilesInFlight) {
            uint32_t v[kMtfTilesInFlight];
#pragma unroll
            for (int u = 0; u < kMtfTilesInFlight; u++) {
                uint32_t i = base + u * 64 + lane;
                v[u] = i < n ? t[i] : 0xFFFFFFFFu;
            }
#pragma unroll
            for (int u = 0; u < kMtfTilesInFlight; u++) {
                const bool mine = (v[u] & 0xFF00u) == 0 && (v[u] >> 16) == ctx;   // sym < 256 and my context
                uint64_t mask = __ballot(mine);
                if (mask == 0) continue;
                uint32_t myrank = 0;
                while (mask) {
                    const int l = __ffsll((long long)mask) - 1;
                    mask &= mask - 1;
                    const uint32_t c = __shfl((int)(v[u] & 0xFF), l);
                    // ZlingMTFEncoder::Encode: rank = index[c]; swap with the entry at mtfnext[rank]
                    const uint32_t i = index[c];
                    const uint32_t nx = mtf_next(i);
                    const uint32_t d = table[nx];
                    if (lane == 0) { index[c] = (uint8_t)nx; index[d] = (uint8_t)i; table[i] = (uint8_t)d; table[nx] = (uint8_t)c; }
                    if ((int)lane == l) myrank = i;
                }
                if (mine) t[base + u * 64 + lane] = myrank | ctx << 16;
            }
        }
    }
    __syncthreads();
    for (uint32_t i = lane; i < 256; i += 64) st[i] = table[i];
}

void launch_mtf_rank(const MtfArgs& a, hipStream_t s) {
    hipLaunchKernelGGL(k_mtf_rank, dim3(256), dim3(64), 0, s, a);
}

}  // namespace zlng
