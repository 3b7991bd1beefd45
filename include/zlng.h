/*
 * zlng.h -- C-ABI of the MI355X-native ROLZ+Huffman block codec (libzlng_hip.so).
 *
 * This is the drop-in boundary for libzling's hot path.  It sits exactly where
 * the reference's block driver (src/libzling.cpp:187-284 for Encode, :306-420 for
 * Decode) calls into its private codec classes:
 *
 *   reference call site                                   replaced by
 *   ---------------------------------------------------   --------------------------
 *   EncodeResource ctor            src/libzling.cpp:108    zlng_create(.., is_encode=1)
 *   lzencoder->Reset()             src/libzling.cpp:197  \
 *   lzencoder->Encode(..)          src/libzling.cpp:206   |
 *   freq loop                      src/libzling.cpp:219   |  zlng_encode_blocks /
 *   ZlingMakeLengthTable x2        src/libzling.cpp:225   |  zlng_encode_blocks_device
 *   ZlingMakeEncodeTable x2        src/libzling.cpp:228   |  (a whole range of 16 MiB
 *   table + bit-pack loops         src/libzling.cpp:232   |   blocks per call)
 *   level adaptation               src/libzling.cpp:261   |
 *   sub-block / block framing      src/libzling.cpp:200,269-279 /
 *   DecodeResource ctor            src/libzling.cpp:136    zlng_create(.., is_encode=0)
 *   Decode body                    src/libzling.cpp:306-420 zlng_decode_blocks
 *   ~EncodeResource/~DecodeResource                        zlng_destroy
 *
 * The C++ API baidu::zling::{Encode,Decode,Inputter,Outputter,ActionHandler}
 * (src/libzling.h:44-45, src/libzling_utils.h:48-119) is re-implemented on top of
 * this ABI in libzling_amd/cxx/ (libzling_amd.so, headers in include/libzling/).
 *
 * Plain C types only; no exceptions cross the ABI; one context is used from one
 * thread at a time; all working memory lives in HBM and belongs to the context.
 * The stream state that the reference keeps inside its long-lived encoder object --
 * the 256 MTF tables (src/libzling_lz.h:105; they are NOT reset per block,
 * src/libzling_lz.cpp:197-209) and `current_level` (src/libzling.cpp:185) --
 * is carried by the context from one call to the next and can be exported /
 * imported for block-range sharding across GPUs.
 */
#ifndef ZLNG_H
#define ZLNG_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define ZLNG_BLOCK_SIZE   16777216u   /* kBlockSizeIn, src/libzling.cpp:70 */
#define ZLNG_MTF_STATE    65536u      /* 256 tables x 256 bytes, src/libzling_lz.h:50-57 */

/* error codes (negative); 0 = success */
enum {
    ZLNG_OK            =  0,
    ZLNG_E_ARG         = -1,   /* bad argument (level outside 0..4, NULL, misaligned range) */
    ZLNG_E_NOMEM       = -2,   /* HBM / host allocation failed (reference: std::bad_alloc)   */
    ZLNG_E_CAP         = -3,   /* output capacity too small                                  */
    ZLNG_E_DEVICE      = -4,   /* HIP runtime error, or no gfx950 device / kernels missing   */
    ZLNG_E_PAYLOAD     = -5,   /* a sub-block payload exceeds 393,216 B (kBlockSizeHuffman); the
                                  reference would overrun obuf here, src/libzling.cpp:232-257  */
    /* decode: one code per reference exception (src/libzling.cpp:316,327,382,392,399,407) */
    ZLNG_E_FLAG        = -10,  /* "invalid encflag."                         */
    ZLNG_E_BLOCKSIZE   = -11,  /* "invalid block size."                      */
    ZLNG_E_CODE1       = -12,  /* "invalid huffman stream. (bad code1)"      */
    ZLNG_E_CODE2       = -13,  /* "invalid huffman stream. (bad code2)"      */
    ZLNG_E_EXBITS      = -14,  /* "invalid huffman stream. (bad ex-bits)"    */
    ZLNG_E_LZ          = -15,  /* "lzdecode failed."                         */
    ZLNG_E_TRUNC       = -16   /* stream ends inside a sub-block (reference reads garbage) */
};

typedef struct zlng_ctx zlng_ctx;

/* Number of usable gfx950 devices (0 if none; the library never falls back to the CPU). */
int zlng_device_count(void);

/* Create a stream context on HIP device `device`.  level 0..4 (ignored for decode).
 * max_blocks = largest number of 16 MiB blocks a single call will pass (sizes the HBM pools; 1..240 --
 * longer streams are fed in several calls, the stream state is carried by the context).
 * Returns NULL on error; *err (optional) receives the code.
 * A call that fails (ZLNG_E_CAP, ZLNG_E_PAYLOAD, ZLNG_E_DEVICE ...) leaves the stream state -- MTF tables and
 * current_level -- as it found it, so it can be repeated, e.g. with a larger output buffer. */
zlng_ctx* zlng_create(int device, int level, int is_encode, int max_blocks, int* err);
void      zlng_destroy(zlng_ctx*);

/* Worst-case .zlng size for in_len input bytes. */
size_t zlng_encode_bound(size_t in_len);

/* Encode a range of blocks held in HOST memory.  `in_len` must be a multiple of
 * ZLNG_BLOCK_SIZE except for the final call of a stream.  Emits the blocks' complete
 * framing (sub-block headers, payloads, per-block 0x00).  per_block_out_end (optional,
 * ceil(in_len/16Mi) entries) receives the end offset of each block's bytes inside `out`,
 * so the C++ shim can push bytes and fire ActionHandler::OnProcess in the reference's order. */
int zlng_encode_blocks(zlng_ctx*, const uint8_t* in, size_t in_len,
                       uint8_t* out, size_t out_cap, size_t* out_len, size_t* per_block_out_end);

/* Same, with input and output already resident in this device's HBM (hipMalloc'd or a
 * torch tensor's data_ptr()).  d_in must be readable for in_len + 512 bytes. */
int zlng_encode_blocks_device(zlng_ctx*, const void* d_in, size_t in_len,
                              void* d_out, size_t out_cap, size_t* out_len, size_t* per_block_out_end);

/* Split form for block-range sharding: the parse of a range does not depend
 * on the incoming MTF state, the rank + Huffman stages do.  parse -> (import state) -> finish. */
int zlng_encode_parse_device(zlng_ctx*, const void* d_in, size_t in_len);
int zlng_encode_finish_device(zlng_ctx*, void* d_out, size_t out_cap, size_t* out_len, size_t* per_block_out_end);
/* A range that goes through several contexts of ONE device (more than 240 blocks, or to overlap its stages): whatever is queued
 * on `ctx` after this call starts once the parse last queued on `first` has finished.  Keeping two parses in flight makes the
 * parses END in stream order, so that context k's rank stage (the stream-ordered part, src/libzling_lz.cpp:112-117 through the
 * persistent tables) runs beside context k + 1's parse instead of behind all of them.  No reference counterpart: the reference
 * has one block in flight (src/libzling.cpp:187-284). */
int zlng_encode_parse_after(zlng_ctx* ctx, zlng_ctx* first);

/* The same split with caller-owned host buffers, for a host-side pipeline over two contexts (the
 * C++ shim): zlng_encode_parse copies `in` to the device and queues the parse on
 * the context's own stream WITHOUT waiting for it; zlng_encode_finish imports nothing by itself
 * (call zlng_set_state first when the range does not open the stream), runs rank + Huffman,
 * copies the bytes back and returns when they are in `out`.  While one context finishes range k,
 * another one can already parse range k+1. */
int zlng_encode_parse(zlng_ctx*, const uint8_t* in, size_t in_len);
int zlng_encode_finish(zlng_ctx*, uint8_t* out, size_t out_cap, size_t* out_len, size_t* per_block_out_end);

/* zlng_encode_finish in two halves, for a driver that finishes several contexts in stream order (zlng_group_encode_finish): _staged
 * runs rank + Huffman + framing and leaves the bytes in the context's own HBM staging buffer -- the stream state the range leaves
 * (zlng_get_state) is final when it returns, so the next context's finish can start --, zlng_encode_copy_out then brings exactly
 * those `n` bytes to the host (blocking; may be called from another thread while OTHER contexts work: it touches only this one).
 * out_cap bounds what the range may produce, like zlng_encode_finish's (ZLNG_E_CAP leaves the range pending and the state as it was). */
int zlng_encode_finish_staged(zlng_ctx*, size_t out_cap, size_t* out_len, size_t* per_block_out_end);
int zlng_encode_copy_out(zlng_ctx*, uint8_t* out, size_t n);

/* Stream state hand-off: 65,536 bytes of MTF tables (context-major) + current_level (0, or the context's level:
 * src/libzling.cpp:261-266 -- anything else is rejected). */
int zlng_get_state(zlng_ctx*, uint8_t mtf[ZLNG_MTF_STATE], int* current_level);
int zlng_set_state(zlng_ctx*, const uint8_t mtf[ZLNG_MTF_STATE], int current_level);
/* Same with the 65,536 table bytes in this device's HBM (e.g. the buffer of an RCCL send/recv). */
int zlng_get_state_device(zlng_ctx*, void* d_mtf, int* current_level);
int zlng_set_state_device(zlng_ctx*, const void* d_mtf, int current_level);

/* MEASURED ALTERNATIVE, off by default (SURVEY 8(e) Option C; the product path is all-device): the k longest rank chains of every
 * following encode call (k <= 8; 0 = off) are walked by host threads -- their literal runs cross PCIe and come back as ranks --
 * while the device walks the other chains.  One wavefront walks a chain at ~19 ns per literal, a host core at ~3.4 ns
 * (src/libzling_lz.cpp:112-117 is a dependent chain either way), and the chain of the hottest context is the term that does not
 * shard across GPUs.  Same bytes.  ZLNG_HOST_RANK_CONTEXTS=<k> sets it for every context of a process. */
int zlng_set_host_rank_contexts(zlng_ctx*, int k);

/* ---- one stream over several devices (replaces the same reference call sites as zlng_encode_blocks) ----
 * A group owns one encode context per entry of `devices` (a device may be listed more than once: several contexts per
 * GPU).  One call takes up to members x max_blocks_per_member blocks of ONE stream: contiguous block ranges are copied
 * to and parsed on all members at once; rank + Huffman then run member by member in stream order with the 64 KiB MTF
 * tables and current_level handed from one member to the next, because the reference carries both through the whole
 * stream (src/libzling.cpp:185, 197, 261-266; src/libzling_lz.cpp:197-209).  The bytes are exactly those of a
 * single-context encode.  The stream state lives in the group between calls.  After a failed finish the group's state
 * is unchanged and the range has to be submitted again. */
typedef struct zlng_group zlng_group;
zlng_group* zlng_group_create(const int* devices, int ndev, int level, int max_blocks_per_member, int* err);
void        zlng_group_destroy(zlng_group*);
int         zlng_group_members(const zlng_group*);
size_t      zlng_group_capacity(const zlng_group*);      /* input bytes one call may pass */
int zlng_group_encode_blocks(zlng_group*, const uint8_t* in, size_t in_len,
                             uint8_t* out, size_t out_cap, size_t* out_len, size_t* per_block_out_end);
int zlng_group_encode_parse(zlng_group*, const uint8_t* in, size_t in_len);      /* returns when the copies are staged */
int zlng_group_encode_finish(zlng_group*, uint8_t* out, size_t out_cap, size_t* out_len, size_t* per_block_out_end);
int zlng_group_get_state(zlng_group*, uint8_t mtf[ZLNG_MTF_STATE], int* current_level);
int zlng_group_set_state(zlng_group*, const uint8_t mtf[ZLNG_MTF_STATE], int current_level);
int zlng_group_set_host_rank_contexts(zlng_group*, int k);      /* zlng_set_host_rank_contexts on every member */

/* Decode a stream prefix made of whole blocks from HOST memory; *in_used gets the bytes consumed. */
int zlng_decode_blocks(zlng_ctx*, const uint8_t* in, size_t in_len, size_t* in_used,
                       uint8_t* out, size_t out_cap, size_t* out_len, size_t* per_block_out_end);
int zlng_decode_blocks_device(zlng_ctx*, const void* d_in, size_t in_len, size_t* in_used,
                              void* d_out, size_t out_cap, size_t* out_len, size_t* per_block_out_end);

/* Per-stage device time of the last encode/decode call in milliseconds (HIP events on the
 * context's stream).  names[i] is a static string; returns the number of stages filled. */
int zlng_last_timings(zlng_ctx*, const char** names, float* ms, int cap);

/* Environment read when a context is created (testing / measurement aids; none of them changes the bytes produced):
 *   ZLNG_PARSER=serial            the one-lane cross-check form of the block parser (default: the workgroup-wide window parser,
 *                                 rolz_wg.hip); ZLNG_WG_WAVES=2|4|8: wavefronts per block of the default, ZLNG_WG_COMPACT=1: the paired slot
 *                                 records of levels 1-4 at level 0 too (default there: the wide form), ZLNG_WG_HOT=1: the hottest
 *                                 context's bucket mirrored in LDS
 *   ZLNG_TOK_CAP=<words>          token words per block the pools start with (they grow once on overflow)
 *   ZLNG_HOST_RANK_CONTEXTS=<k>   MEASURED ALTERNATIVE, off by default: the k longest rank chains of a call are walked
 *                                 by host threads (literal runs over PCIe and back) while the device walks the others.
 *                                 The product path is all-device; this mode exists because SURVEY 8(e) asks to choose by
 *                                 measurement and bench.py reports it as a separate line.
 *   ZLNG_PROFILE=1                parser phase counters (scripts/perf_probe.py)
 *   ZLNG_MIN_RESTART=-1           levels 1-4: replay every hard token by the serial code (default: the next round starts at it);
 *                                 -2: additionally recompute the token-chain closure in full after every iteration (A/B of the incremental update)
 *   ZLNG_PREFIX_PCT=<0..100>      parser: after a round's first iteration, commit the tokens in front of the first changed one instead of
 *                                 iterating when they are at least this share of the round's tokens (0 = always iterate, the default: measured within +-1 % on every workload)
 *   ZLNG_RING_FIX=1               parser, levels 1-4: a chain node whose ring slot a token of the same round has taken over ends the walk in front
 *                                 of it (exact: the reference's chain-end test stops there) instead of making the token a serially replayed one.
 *                                 Selects separate kernel instantiations (rolz_wg.hip kRingRule): the default ones hold none of its code.  Exact in the
 *                                 CPU model at every level (scripts/experiments/wg_parser_model.c), never run on a GPU: off
 *   ZLNG_CHAIN_PRIO=0             k_mtf_chain without its raised issue priority (s_setprio 3; on by default: the rank chain is one dependent
 *                                 chain per wavefront, and beside another context's parser waves on its SIMD it should win every arbitration)
 *   ZLNG_DEBUG_PACK_LDS=<bytes>   extra dynamic LDS for the bit packer's launch (occupancy experiments)
 * (Not read by the library, but relevant to it: GPU_MAX_HW_QUEUES -- the HIP runtime maps user streams onto a few hardware queues by
 *  default; a process that drives a range through several contexts at once should give every stream its own queue, as bench.py does.)
 * (The C++ shim reads ZLNG_DEVICE, ZLNG_DEVICES, ZLNG_BATCH_BLOCKS and ZLNG_PIPELINE: INTEGRATION.md.  The build reads
 *  ZLNG_HIPCC_FLAGS and ZLNG_BUILD_FORCE (__graft_entry__.py); bench.py reads ZLNG_ENWIK9 / ZLNG_ENWIK8 (a real enwik file
 *  to use instead of the generator) and ZLNG_BENCH_ONE_DEVICE (all ranks on device 0, collectives over gloo: a test hook), ZLNG_BENCH_STANDIN (with ZLNG_HIP_SO naming
 *  the CPU stand-in of this ABI that the test suite builds, tests/cxx/zlng_stub.c: bench.py's host logic on the CPU -- a test hook, never a measurement);
 *  scripts/sanitize.sh sets ZLNG_ORACLE_SO (the ASan build of the oracle), ZLNG_NO_REF, ZLNG_DEMO, ZLNG_PROTOCOL_TEST, ZLNG_HIP_SO (another build of libzlng_hip.so) and ZLNG_SYSTEM_HIP (the ASan build of zling_demo) for the tests.) */

/* Test hooks (used by tests/ and scripts/ only; no stability promise):
 *   zlng_debug_fetch     copy an internal per-block buffer of the last encode call to the host (what: 0 tokens, 1 cuts, 2 freq,
 *                        3 lens, 4 olen, 5 ntok/nsub, 6 codes, 7 sub-block output offsets, 8 literals per context)
 *   zlng_debug_lengths   run K4 (code lengths + canonical codes) on caller-supplied rows of 546 counts
 *   zlng_debug_passes    parse passes the last encode call needed (1 = no level-schedule repair, no pool growth)
 *   zlng_debug_counters  kDbgSlots (24) profile counters per block of the last parse (ZLNG_PROFILE=1) */
int zlng_debug_fetch(zlng_ctx*, int what, int blk, void* dst, size_t bytes);
int zlng_debug_lengths(zlng_ctx*, const uint32_t* freq, int nrows, uint8_t* lens, uint16_t* codes);
int zlng_debug_passes(zlng_ctx*);
int zlng_debug_counters(zlng_ctx*, unsigned long long* out, int nblocks);

/* The hipStream_t the context launches on (as void*), for callers that need to order work. */
void* zlng_stream(zlng_ctx*);

const char* zlng_strerror(int code);

#ifdef __cplusplus
}
#endif
#endif /* ZLNG_H */
