// rolz_parse.hip -- K0/K1: dictionary reset and the ROLZ block parser for gfx950.
//
// Replaces ZlingRolzEncoder::Reset / Encode / EncodeImpl / MatchAndUpdate / MatchLazy
// (src/libzling_lz.cpp:128-316 of the reference).  One workgroup owns one 16 MiB block:
// the parse of a block is a strict serial chain of token decisions (dictionary inserts
// happen only at token starts, src/libzling_lz.cpp:159), while different blocks are
// independent (Reset per block, src/libzling.cpp:197) -- so blocks are the grid dimension.
//
// Output per block: one u32 word per token (zlng_common.h) with literals still RAW
// (the rank stage K2 is stream-serial and runs afterwards), plus the sub-block cut list.
#include "zlng_common.h"
#include "zlng_kernels.h"
#include "rolz_dev.h"

namespace zlng {

// ------------------------------------------------------------------------------ K0
// Reset(): offset = 0, suffix = 0xFFFF, hash = 0xFFFF for every bucket (src/libzling_lz.cpp:197-209).
// Pure streaming fill: 14.7 MB per block, 16 B per lane per store.  `wide`: the form of the slots the following parse uses
// (zlng_common.h).
// slots: words 0 (wide: {own, link's copy}; paired: own word 0, link 65535 in the upper half); link plane and hash heads: 65535
__global__ __launch_bounds__(256) void k_dict_reset(uint8_t* dict, uint32_t nblocks, uint32_t wide) {
    const size_t vec_per_bkt = kBktBytes / 16;                       // 3584 uint4 per bucket
    const size_t total = (size_t)nblocks * 256 * vec_per_bkt;
    uint4* d = reinterpret_cast<uint4*>(dict);
    const uint32_t hi = wide ? 0u : 0xFFFFFFFFu;
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (size_t)gridDim.x * blockDim.x) {
        uint32_t within = (uint32_t)(i % vec_per_bkt) * 16;
        d[i] = within < 8u * (uint32_t)kRing ? make_uint4(0u, hi, 0u, hi) : make_uint4(0xFFFFFFFFu, 0xFFFFFFFFu, 0xFFFFFFFFu, 0xFFFFFFFFu);
    }
}

// ------------------------------------------------------------------------------ K1 (serial form)
// One lane walks the block exactly as EncodeImpl does (src/libzling_lz.cpp:139-195).  It is the
// bring-up / cross-check form of the parser: slow (every token pays a chain of dependent HBM
// round trips) but a direct statement of the semantics, kept selectable with ZLNG_PARSER=serial.
__global__ __launch_bounds__(64) void k_rolz_parse_serial(ParseArgs a) {
    __shared__ uint16_t heads[256];
    __shared__ uint32_t mru[256];          // slot0 in bits 0..15, slot1 in bits 16..31
    const uint32_t blk = blockIdx.x + a.blk0;
    const size_t base = (size_t)blk * kBlockIn;
    if (base >= a.in_len) return;
    const uint8_t* buf = a.in + base;
    const int ilen = (int)((a.in_len - base) < (size_t)kBlockIn ? (a.in_len - base) : (size_t)kBlockIn);
    uint8_t* dict = a.dict + (size_t)blk * kDictBytes;
    uint32_t* tok = a.tok + (size_t)blk * a.tok_cap;
    SubCut* cuts = a.cuts + (size_t)blk * kMaxSub;

    for (int i = threadIdx.x; i < 256; i += 64) heads[i] = 0;
    __syncthreads();
    if (threadIdx.x != 0) return;

    int ipos = 0, nsub = 0;
    uint32_t nt = 0;
    while (ipos < ilen) {
        const LevelCfg cfg = level_cfg(a.lvl_sched[blk * kMaxSub + (nsub < kMaxSub ? nsub : kMaxSub - 1)]);
        const uint32_t tok_begin = nt;
        int opos = 0;
        for (int i = 0; i < 256; i++) mru[i] = 0;
        if (ipos == 0 && ipos < ilen) { tok[nt++] = buf[ipos++] | kTokRawCtx << 16; opos++; }
        if (ipos == 1 && ipos < ilen) { tok[nt++] = buf[ipos++] | kTokRawCtx << 16; opos++; }
        while (opos + 1 < kSubSyms && ipos < ilen) {
            if (nt + 1 > a.tok_cap) { *a.overflow = 1; a.nsub[blk] = 0; a.ntok[blk] = 0; return; }
            int midx, mlen;
            bool hit = false;
            if (ipos + kSentinel < ilen) {
                const uint32_t c = buf[ipos - 1];
                const uint32_t head = (heads[c] + 1u) & (kRing - 1);
                heads[c] = (uint16_t)head;
                hit = match_exact(dict, buf, ipos, cfg, head, true, midx, mlen);
            }
            if (hit) {
                tok[nt++] = (uint32_t)(258 + mlen - kMatchMin) | (uint32_t)midx << 16;
                opos += 2;
                ipos += mlen;
                uint32_t w = (uint32_t)buf[ipos - 2] << 8 | buf[ipos - 1];
                uint32_t m = mru[buf[ipos - 3]];
                if ((m & 0xFFFF) != w) mru[buf[ipos - 3]] = (m << 16) | w;
                continue;
            }
            if (ipos + 1 < ilen) {
                uint32_t w = (uint32_t)buf[ipos] << 8 | buf[ipos + 1];
                uint32_t m = mru[buf[ipos - 1]];
                if ((m & 0xFFFF) == w) { tok[nt++] = 256; opos++; ipos += 2; continue; }
                if ((m >> 16) == w) { tok[nt++] = 257; opos++; ipos += 2; mru[buf[ipos - 3]] = (m << 16) | w; continue; }
            }
            tok[nt++] = (uint32_t)buf[ipos] | (uint32_t)buf[ipos - 1] << 16;
            opos++;
            ipos++;
            uint32_t m = mru[buf[ipos - 3]];
            mru[buf[ipos - 3]] = (m << 16) | ((uint32_t)buf[ipos - 2] << 8 | buf[ipos - 1]);
        }
        if (nsub < kMaxSub) cuts[nsub] = SubCut{tok_begin, nt, (uint32_t)ipos, (uint32_t)opos};
        nsub++;
    }
    a.nsub[blk] = (uint32_t)nsub;
    a.ntok[blk] = nt;
}

void launch_dict_reset(uint8_t* dict, uint32_t nblocks, hipStream_t s, bool wide) {
    hipLaunchKernelGGL(k_dict_reset, dim3(2048), dim3(256), 0, s, dict, nblocks, wide ? 1u : 0u);
}
void launch_rolz_parse_serial(const ParseArgs& a, uint32_t nblocks, hipStream_t s) {
    hipLaunchKernelGGL(k_rolz_parse_serial, dim3(nblocks - a.blk0), dim3(64), 0, s, a);
}

}  // namespace zlng
