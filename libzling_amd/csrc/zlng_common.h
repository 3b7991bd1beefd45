// zlng_common.h -- constants, HBM layouts and small device helpers shared by the gfx950 kernels.
//
// Format constants restate src/libzling.cpp:63-72 and src/libzling_lz.h:44-48 of the
// reference; the HBM layouts are this library's own.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

namespace zlng {

constexpr int kBlockIn     = 1 << 24;   // kBlockSizeIn
constexpr int kSubSyms     = 262144;    // kBlockSizeRolz (u16 entries per sub-block)
constexpr int kPayloadMax  = 393216;    // kBlockSizeHuffman
constexpr int kSentinel    = 275;       // kMatchMaxLen + 16
constexpr int kNsym1       = 514;
constexpr int kNsym2       = 32;
constexpr int kNsymAll     = kNsym1 + kNsym2;   // 546: one row of freq/len/code tables
constexpr int kMaxLen1     = 15;
constexpr int kMaxLen1Fast = 10;        // kHuffmanMaxLen1Fast, src/libzling.cpp:67: the decoder asks a 2^10-entry table first
constexpr int kMaxLen2     = 8;
constexpr int kRing        = 4096;      // kBucketItemSize
constexpr int kHashSlots   = 8192;      // kBucketItemHash
constexpr int kMatchMin    = 4;
constexpr int kMatchMax    = 259;
constexpr int kLazyLimit   = 128;       // kMatchMinLenEnableLazy
constexpr int kTableBytes  = 273;       // 257 + 16 nibble-packed length bytes
constexpr int kHeaderBytes = 13;        // flag + 3 x u32be

// One 16 MiB block can hold at most 16Mi one-byte tokens -> <= 65 sub-blocks; 80 leaves slack.
constexpr int kMaxSub = 80;

// ---- HBM layout of one context's dictionary: 256 buckets x 57,344 B ------------------
// (the reference's ZlingEncodeBucket, src/libzling_lz.h:98-103, re-laid as three planes; `head` lives in LDS
//  during a parse.)  Two forms of the slot plane, chosen per parse kernel (rolz_dev.h BucketT):
//   paired   8-byte slot = own word (pos | hash_check << 24) + the slot's link (u16): the generic chain walk of levels 1-4
//            (and the serial / pipelined parsers) reads a node's word and its successor's index from one record;
//   wide     8-byte slot = own word + a copy of the word of the slot it links to, taken when the link was made:
//            the level-0 wave parser reads a chain's second node without a dependent load (`speculate_l0w`
//            says when the copy is what the reference would read and what follows when it is not).
constexpr uint32_t kBktBytes     = 8 * kRing + 2 * kRing + 2 * kHashSlots;   // 57,344 (the wide form)
constexpr size_t   kDictBytes    = (size_t)256 * kBktBytes;                  // 14,680,064 per block

// ---- token word (same as oracle/zlng_oracle.h) ---------------------------------------
//  bits 0..15  alphabet-1 symbol (raw byte before the rank stage, rank after it)
//  bits 16..31 match: match_idx | literal: context byte, 0xFFFF for the 2 raw block-opening bytes
constexpr uint32_t kTokRawCtx = 0xFFFFu;
// Token words reserved per block (`tok_cap` in the kernel argument blocks) are a property of the context: text needs
// 0.25-0.29 tokens per byte, incompressible data one per byte.  A context starts at kTokCapDefault and, when a parse
// reports an overflow, grows once to the worst case kTokCapMax and repeats the call (zlng_api.hip).  Multiples of 4096
// (one partition tile of the rank stage).
constexpr uint32_t kTokCapMax     = (uint32_t)kBlockIn;          // one token per input byte
constexpr uint32_t kTokCapDefault = 7u << 20;                    // 0.4375 tokens per byte
constexpr uint32_t kDbgSlots      = 24;                          // parser profile counters per block (ZLNG_PROFILE=1)

struct SubCut {            // one sub-block of one block
    uint32_t tok_begin;    // first token (index into the block's token array)
    uint32_t tok_end;
    uint32_t encpos;       // input offset reached (cumulative within the block)
    uint32_t rlen;         // reference u16 count (tokens + matches)
};

struct LevelCfg { int depth, lazy1, lazy2; };
// src/libzling_lz.cpp:128-137
__host__ __device__ inline LevelCfg level_cfg(int level) {
    switch (level) {
        case 0: return {2, 1, 0};
        case 1: return {4, 1, 0};
        case 2: return {6, 2, 0};
        case 3: return {8, 3, 1};
        default: return {16, 4, 2};
    }
}

// src/tables/gen.py:10-18 as arithmetic: bucket code, extra-bit count and extra bits of a match_idx.
__host__ __device__ inline void matchidx_split(uint32_t idx, uint32_t& code, uint32_t& blen, uint32_t& extra) {
    if (idx < 4) { code = idx; blen = 0; extra = 0; }
    else if (idx < 512) {
#if defined(__HIP_DEVICE_COMPILE__)
        uint32_t k = 31u - (uint32_t)__clz((int)idx);
#else
        uint32_t k = 31u - (uint32_t)__builtin_clz(idx);
#endif
        code = 2 * k + ((idx >> (k - 1)) & 1u); blen = k - 1; extra = idx & ((1u << blen) - 1u);
    } else { code = 16 + (idx >> 8); blen = 8; extra = idx & 255u; }
}
__host__ __device__ inline uint32_t matchidx_blen_of_code(uint32_t c) { return c < 4 ? 0u : (c < 18 ? (c - 2) / 2 : 8u); }

// src/tables/gen.py:52-56: int(0.95*i) for i < 128, int(0.55*i) otherwise (exact in integers).
__host__ __device__ inline uint32_t mtf_next(uint32_t i) { return i < 128 ? (i * 95u) / 100u : (i * 55u) / 100u; }

#define ZLNG_HIP_CHECK(expr)                                                      \
    do { hipError_t e_ = (expr); if (e_ != hipSuccess) { last_hip_error = e_; return ZLNG_E_DEVICE; } } while (0)

}  // namespace zlng
