// zlng_kernels.h -- kernel argument blocks and launchers (internal to libzlng_hip.so).
#pragma once
#include "zlng_common.h"

namespace zlng {

// ---- K0/K1 ---------------------------------------------------------------------------
struct ParseArgs {
    const uint8_t* in;        // NB x 16 MiB contiguous input (readable for in_len + 512)
    size_t         in_len;
    uint8_t*       dict;      // NB x kDictBytes
    uint32_t*      tok;       // NB x tok_cap token words (literals RAW on exit)
    SubCut*        cuts;      // NB x kMaxSub
    uint32_t*      nsub;      // NB
    uint32_t*      ntok;      // NB
    const uint8_t* lvl_sched; // NB x kMaxSub: level of each sub-block (src/libzling.cpp:261-266 speculation)
    unsigned long long* dbg;  // optional NB x kDbgSlots counters (cycles per phase, rounds, redos); may be null
    int            min_restart; // levels 1-4: < 0 replays every hard token by the serial code instead of starting the next round at it (ZLNG_MIN_RESTART)
    int            prefix_pct;  // after a round's first iteration: commit the tokens in front of the first changed one instead of iterating when they
                                // are at least this share (%) of the round's tokens (0: always iterate, the default; ZLNG_PREFIX_PCT)
    uint32_t       tok_cap;     // token words reserved per block
    uint32_t       blk0;        // first block of this launch (a level-schedule repair re-parses a tail of the range)
    uint32_t*      overflow;    // 1: a block ran out of token words (its output is then incomplete; the host grows the pools once and repeats);
                                // 2: a "cannot happen" guard of the parser fired (the host fails the call, ZLNG_E_DEVICE)
    int            ring_fix;    // levels 1-4: a chain node taken over by a token of the round ends the walk in front of it instead of making the token hard (ZLNG_RING_FIX).
                                // LAST on purpose: the fields in front keep the kernarg offsets of the kernels that last ran on a GPU, so the level-0
                                // instantiations (which never read this) stay instruction-identical to them (scripts/isa_diff.py)
};
// the workgroup-wide parser (rolz_wg.hip): nw wavefronts per block, window of 64 nw positions
void launch_rolz_parse_wg(const ParseArgs& a, uint32_t nblocks, hipStream_t s, bool all_level0, int nw, bool wide, bool hot);   // wide: slot plane form at level 0; hot: one bucket mirrored in LDS
void launch_dict_reset(uint8_t* dict, uint32_t nblocks, hipStream_t s, bool wide);   // wide: the slot plane form of the level-0 wave parser
// one lane walks the block token by token, exactly as the reference does: the on-device cross-check (ZLNG_PARSER=serial); blocks [a.blk0, nblocks)
void launch_rolz_parse_serial(const ParseArgs& a, uint32_t nblocks, hipStream_t s);

// ---- K2 ------------------------------------------------------------------------------
struct MtfArgs {
    uint32_t*       tok;       // NB x tok_cap, ranked in place (pointer to the first block of the group being ranked)
    const uint32_t* ntok;      // NB
    uint32_t        nblocks;
    uint32_t        tok_cap;
    uint8_t*        state;     // 256 x 256 MTF tables, context-major; updated in place
    uint32_t*       tile_base; // [NB + 1] dense tile number of each block's first tile
    uint32_t*       tile_hist; // [tiles][256] literals per context per tile -> exclusive prefix per context
    uint32_t*       ctx_total; // [256]
    uint32_t*       ctx_off;   // [256] start of each context's dense run
    uint8_t*        lit_byte;  // dense literal bytes, context-major, stream order inside a context; ranks in place
    uint8_t*        snap;      // the context's table (256 B, position order) at the start of every 64-literal tile of lit_byte
    uint8_t*        tile_kk;   // per tile: literals whose ranks k_mtf_replay computes from the snapshot (0 = none)
    const uint8_t*  skip;      // optional [256]: contexts k_mtf_chain leaves alone (the measured host-chain alternative, zlng_api.hip)
    unsigned long long* dbg;   // optional [512]: cycles and slow steps (literals outside the table front) per context (ZLNG_PROFILE=1 with >= 22 blocks)
    uint32_t        prio;      // k_mtf_chain raises its wavefronts' issue priority (s_setprio 3): beside other contexts' parser waves on the same SIMD the chain wins every arbitration (ZLNG_CHAIN_PRIO=0 switches it off)
};
void launch_lit_partition(const MtfArgs& a, hipStream_t s);   // literals -> one dense run per context
void launch_mtf_chain(const MtfArgs& a, hipStream_t s);       // k_mtf_chain: the serial chains
void launch_mtf_finish(const MtfArgs& a, hipStream_t s);      // rank replay per tile + ranks back into the token words

// ---- K3..K6 --------------------------------------------------------------------------
struct HuffArgs {
    const uint32_t* tok;
    const SubCut*   cuts;
    const uint32_t* nsub;
    uint32_t        nblocks;
    uint32_t        tok_cap;
    uint32_t        blk0;      // histogram / lengths only: first block to (re)compute
    uint32_t*       freq;      // [NB*kMaxSub][kNsymAll]
    uint8_t*        lens;      // [NB*kMaxSub][kNsymAll]
    uint16_t*       codes;     // [NB*kMaxSub][kNsymAll] (bit-reversed canonical codes)
    uint32_t*       olen;      // [NB*kMaxSub] payload bytes (273 + bitstream)
    uint64_t*       sub_off;   // [NB*kMaxSub] byte offset of the sub-block's 0x01 flag in the output
    uint64_t*       blk_end;   // [NB] end offset of each block's bytes
    uint64_t*       summary;   // [0] total bytes, [1] error flags: 1 payload > 393,216 B, 2 internal, 4 out_cap, 8 token-pool overflow
    const uint32_t* overflow;  // set by the parser when a block ran out of token words
    uint8_t*        out;
    uint64_t        out_cap;
};
void launch_histogram(const HuffArgs& a, hipStream_t s);
void launch_lengths(const HuffArgs& a, hipStream_t s);
void launch_layout(const HuffArgs& a, hipStream_t s);
void launch_pack(const HuffArgs& a, hipStream_t s);

// ---- K7..K9 (decode) --------------------------------------------------------------------
// error codes as in include/zlng.h (kept numerically identical; checked by static_assert in zlng_api.hip)
constexpr int ZLNG_DEC_E_LIMIT = -1, ZLNG_DEC_E_FLAG = -10, ZLNG_DEC_E_BLOCKSIZE = -11, ZLNG_DEC_E_CODE1 = -12,
              ZLNG_DEC_E_CODE2 = -13, ZLNG_DEC_E_EXBITS = -14, ZLNG_DEC_E_LZ = -15;
constexpr uint32_t kDecSubsPerBlock = 1024;      // sub-block table capacity per block of a decode call

struct DecSub   { uint64_t payload_off, tok_off; uint32_t encpos, rlen, olen, blk; };
struct DecBlock { uint32_t first_sub, nsub; uint64_t out_off; uint32_t size, pad; uint64_t z_end; };   // z_end: compressed bytes up to and including the block's 0x00
struct DecodeArgs {
    const uint8_t* z;          // compressed bytes
    uint64_t       z_len;
    uint32_t       max_blocks, max_subs;
    uint64_t       tok_cap;    // token words available
    DecSub*        subs;
    DecBlock*      blocks;
    uint32_t*      sub_ntok;
    uint32_t*      tok;
    uint32_t*      ring;       // [256][4096] ZlingDecodeBucket::offset (src/libzling_lz.h:132-135)
    uint8_t*       mtf_state;  // 256 x 256, persists across calls
    uint8_t*       mtf_snap;   // 256 x 256 scratch: tables at the start of the block being replayed (restored when it fails)
    uint8_t*       out;
    uint64_t       out_cap;
    uint64_t*      summary;    // [0] bytes consumed [1] frame error behind the complete blocks (positive code) [2] complete blocks
                               // [3] sub-blocks [4] output bytes [5] first block that failed in K8/K9 (== [2] if none) [6] its error code
};
void launch_frame_walk(const DecodeArgs& a, hipStream_t s);
void launch_huff_decode(const DecodeArgs& a, uint32_t nsubs, hipStream_t s);
void launch_rolz_decode(const DecodeArgs& a, hipStream_t s);

}  // namespace zlng
