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

namespace zlng {

// ------------------------------------------------------------------------------ K0
// Reset(): offset = 0, suffix = 0xFFFF, hash = 0xFFFF for every bucket (src/libzling_lz.cpp:197-209).
// Pure streaming fill: 10.5 MB per block, 16 B per lane per store.
__global__ __launch_bounds__(256) void k_dict_reset(uint8_t* dict, uint32_t nblocks) {
    const size_t vec_per_bkt = kBktBytes / 16;                       // 2560 uint4 per bucket
    const size_t total = (size_t)nblocks * 256 * vec_per_bkt;
    uint4* d = reinterpret_cast<uint4*>(dict);
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (size_t)gridDim.x * blockDim.x) {
        uint32_t within = (uint32_t)(i % vec_per_bkt) * 16;
        uint32_t v = within < kBktSuffixOff ? 0u : 0xFFFFFFFFu;
        d[i] = make_uint4(v, v, v, v);
    }
}

// ------------------------------------------------------------------------------ helpers
__device__ __forceinline__ uint32_t ld32u(const uint8_t* p) { uint32_t v; __builtin_memcpy(&v, p, 4); return v; }

// HashContext, src/libzling_lz.cpp:55-57
__device__ __forceinline__ uint32_t hash4(const uint8_t* p) {
    uint32_t w = ld32u(p);
    return w + ((w >> 16) & 0xFF) * 137u + (w >> 24) * 13337u;
}

// GetCommonLength, src/libzling_lz.cpp:66-89: 0 unless 4 bytes agree, else byte-wise LCP capped at 259.
__device__ __forceinline__ int common_len(const uint8_t* a, const uint8_t* b) {
    if (ld32u(a) != ld32u(b)) return 0;
    int n = 4;
    while (n + 4 <= kMatchMax) {
        uint32_t x = ld32u(a + n) ^ ld32u(b + n);
        if (x) return n + (__ffs((int)x) - 1) / 8;
        n += 4;
    }
    while (n < kMatchMax && a[n] == b[n]) n++;
    return n;
}

struct Bucket {
    uint32_t* offset; uint16_t* suffix; uint16_t* hash;
    __device__ __forceinline__ Bucket(uint8_t* dict, uint32_t ctx) {
        uint8_t* b = dict + (size_t)ctx * kBktBytes;
        offset = reinterpret_cast<uint32_t*>(b + kBktOffsetOff);
        suffix = reinterpret_cast<uint16_t*>(b + kBktSuffixOff);
        hash   = reinterpret_cast<uint16_t*>(b + kBktHashOff);
    }
};

// MatchLazy, src/libzling_lz.cpp:291-316
__device__ __forceinline__ bool lazy_probe(uint8_t* dict, const uint8_t* buf, int pos, int maxlen, int depth) {
    Bucket B(dict, buf[pos - 1]);
    uint32_t node = B.hash[hash4(buf + pos) % kHashSlots];
    if (node == 65535) return false;
    int m = maxlen - 3;
    for (int i = 0; i < depth; i++) {
        uint32_t off = B.offset[node] & 0xFFFFFF;
        if (ld32u(buf + pos + m) == ld32u(buf + off + m)) return true;
        node = B.suffix[node];
        if (node == 65535 || off <= (B.offset[node] & 0xFFFFFF)) break;
    }
    return false;
}

// MatchAndUpdate, src/libzling_lz.cpp:211-289 (insert first, then walk <= depth chain nodes)
__device__ __forceinline__ bool match_and_update(uint8_t* dict, uint16_t* heads, const uint8_t* buf, int pos,
                                                 const LevelCfg cfg, int& match_idx, int& match_len) {
    uint32_t h = hash4(buf + pos);
    uint32_t chk = (h / kHashSlots) & 255u;
    uint32_t hc = h % kHashSlots;
    uint32_t ctx = buf[pos - 1];
    Bucket B(dict, ctx);
    uint32_t node = B.hash[hc];
    uint32_t head = (heads[ctx] + 1u) & (kRing - 1);
    heads[ctx] = (uint16_t)head;
    B.suffix[head] = (uint16_t)node;
    B.offset[head] = (uint32_t)pos | chk << 24;
    B.hash[hc] = (uint16_t)head;
    if (node == 65535 || node == head) return false;

    int maxlen = kMatchMin - 1;
    uint32_t maxnode = 0;
    for (int i = 0; i < cfg.depth; i++) {
        uint32_t ov = B.offset[node];
        uint32_t off = ov & 0xFFFFFF;
        if ((ov >> 24) == chk && buf[pos + maxlen] == buf[off + maxlen]) {
            int len = common_len(buf + pos, buf + off);
            if (len > maxlen) { maxnode = node; maxlen = len; if (maxlen == kMatchMax) break; }
        }
        node = B.suffix[node];
        if (node == 65535 || off <= (B.offset[node] & 0xFFFFFF)) break;
    }
    if (maxlen < kMatchMin) return false;
    if (maxlen < kLazyLimit) {
        if (cfg.lazy1 > 0 && lazy_probe(dict, buf, pos + 1, maxlen, cfg.lazy1)) return false;
        if (cfg.lazy2 > 0 && lazy_probe(dict, buf, pos + 2, maxlen, cfg.lazy2)) return false;
    }
    match_len = maxlen;
    match_idx = (int)((head - maxnode) & (kRing - 1));
    return true;
}

// ------------------------------------------------------------------------------ K1 (serial form)
// One lane walks the block exactly as EncodeImpl does (src/libzling_lz.cpp:139-195).  It is the
// bring-up / cross-check form of the parser: slow (every token pays a chain of dependent HBM
// round trips) but a direct statement of the semantics, kept selectable with ZLNG_PARSER=serial.
__global__ __launch_bounds__(64) void k_rolz_parse_serial(ParseArgs a) {
    __shared__ uint16_t heads[256];
    __shared__ uint32_t mru[256];          // slot0 in bits 0..15, slot1 in bits 16..31
    const uint32_t blk = blockIdx.x;
    const size_t base = (size_t)blk * kBlockIn;
    if (base >= a.in_len) return;
    const uint8_t* buf = a.in + base;
    const int ilen = (int)((a.in_len - base) < (size_t)kBlockIn ? (a.in_len - base) : (size_t)kBlockIn);
    uint8_t* dict = a.dict + (size_t)blk * kDictBytes;
    uint32_t* tok = a.tok + (size_t)blk * kTokCap;
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
            int midx, mlen;
            if (ipos + kSentinel < ilen && match_and_update(dict, heads, buf, ipos, cfg, midx, mlen)) {
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

void launch_dict_reset(uint8_t* dict, uint32_t nblocks, hipStream_t s) {
    hipLaunchKernelGGL(k_dict_reset, dim3(2048), dim3(256), 0, s, dict, nblocks);
}
void launch_rolz_parse_serial(const ParseArgs& a, uint32_t nblocks, hipStream_t s) {
    hipLaunchKernelGGL(k_rolz_parse_serial, dim3(nblocks), dim3(64), 0, s, a);
}

void launch_rolz_parse_wave(const ParseArgs& a, uint32_t nblocks, hipStream_t s) {
    launch_rolz_parse_serial(a, nblocks, s);   // TEMP until the wavefront parser lands
}

}  // namespace zlng
