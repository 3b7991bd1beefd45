// zlng_kernels.h -- kernel argument blocks and launchers (internal to libzlng_hip.so).
#pragma once
#include "zlng_common.h"

namespace zlng {

// ---- K0/K1 ---------------------------------------------------------------------------
struct ParseArgs {
    const uint8_t* in;        // NB x 16 MiB contiguous input (readable for in_len + 512)
    size_t         in_len;
    uint8_t*       dict;      // NB x kDictBytes
    uint32_t*      tok;       // NB x kTokCap token words (literals RAW on exit)
    SubCut*        cuts;      // NB x kMaxSub
    uint32_t*      nsub;      // NB
    uint32_t*      ntok;      // NB
    const uint8_t* lvl_sched; // NB x kMaxSub: level of each sub-block (src/libzling.cpp:261-266 speculation)
    unsigned long long* dbg;  // optional NB x 16 counters (cycles per phase, rounds, redos); may be null
};
void launch_dict_reset(uint8_t* dict, uint32_t nblocks, hipStream_t s);
void launch_rolz_parse_serial(const ParseArgs& a, uint32_t nblocks, hipStream_t s);
void launch_rolz_parse_wave(const ParseArgs& a, uint32_t nblocks, hipStream_t s);

// ---- K2 ------------------------------------------------------------------------------
struct MtfArgs {
    uint32_t*       tok;       // NB x kTokCap, ranked in place
    const uint32_t* ntok;      // NB
    uint32_t        nblocks;
    uint8_t*        state;     // 256 x 256 MTF tables, context-major; updated in place
};
void launch_mtf_rank(const MtfArgs& a, hipStream_t s);

// ---- K3..K6 --------------------------------------------------------------------------
struct HuffArgs {
    const uint32_t* tok;
    const SubCut*   cuts;
    const uint32_t* nsub;
    uint32_t        nblocks;
    uint32_t*       freq;      // [NB*kMaxSub][kNsymAll]
    uint8_t*        lens;      // [NB*kMaxSub][kNsymAll]
    uint16_t*       codes;     // [NB*kMaxSub][kNsymAll] (bit-reversed canonical codes)
    uint32_t*       olen;      // [NB*kMaxSub] payload bytes (273 + bitstream)
    uint64_t*       sub_off;   // [NB*kMaxSub] byte offset of the sub-block's 0x01 flag in the output
    uint64_t*       blk_end;   // [NB] end offset of each block's bytes
    uint64_t*       summary;   // [0] total bytes, [1] error flag (ZLNG_E_PAYLOAD as positive), [2] scratch
    uint8_t*        out;
    uint64_t        out_cap;
};
void launch_histogram(const HuffArgs& a, hipStream_t s);
void launch_lengths(const HuffArgs& a, hipStream_t s);
void launch_layout(const HuffArgs& a, hipStream_t s);
void launch_pack(const HuffArgs& a, hipStream_t s);

}  // namespace zlng
