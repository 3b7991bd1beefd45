// decode.hip -- K7 frame walk, K8 Huffman decode, K9 ROLZ + MTF replay: the inverse path.
//
// Replaces the body of baidu::zling::Decode (src/libzling.cpp:306-420):
//   framing walk / size checks        src/libzling.cpp:312-332
//   length tables -> decode LUTs      src/libzling.cpp:347-365, src/libzling_huffman.cpp:114-153
//   symbol loop                       src/libzling.cpp:368-402
//   ZlingRolzDecoder::Decode          src/libzling_lz.cpp:318-399  (+ ZlingMTFDecoder :119-126)
//
// Parallel structure: sub-block headers are hop-able (olen), so one lane walks them; Huffman
// decode is independent per sub-block (one wavefront each, 15-bit LUT in LDS); the ROLZ replay is
// serial per block AND -- because the literal MTF tables persist across blocks and the context of
// a literal is decoded data -- serial across the whole stream.  K9 is
// therefore one wavefront per stream; its lanes cooperate on the match copies.
#include "zlng_common.h"
#include "zlng_kernels.h"
#include "replay_loop.h"

namespace zlng {

// ------------------------------------------------------------------------------ K7 frame walk
// Errors are met in STREAM order, like the reference's one loop (src/libzling.cpp:312-404): a sub-block's Huffman stream and
// replay come before the flag and sizes of the sub-block behind it.  So a framing error inside a block does not hide what is in
// front of it: the sub-blocks up to it are kept as a PROBE block -- decoded and replayed like any other; if one of them fails, that
// is the call's error; if they all pass, the framing error (`pending`) is.  A probe block is never reported as output.
__global__ __launch_bounds__(64) void k_frame_walk(DecodeArgs a) {
    if (threadIdx.x != 0) return;
    const uint8_t* z = a.z;
    const uint64_t n = a.z_len;
    uint64_t p = 0, used = 0, out_total = 0, tok_total = 0;
    uint32_t nblk = 0, nsub = 0, err = 0;
    while (p < n && nblk < a.max_blocks && !err) {
        const uint32_t first_sub = nsub;
        const uint64_t first_tok = tok_total;
        uint32_t last_encpos = 0, pending = 0;
        bool closed = false, too_many = false;
        while (p < n) {
            const uint32_t flag = z[p++];
            if (flag != 0 && flag != 1) { pending = (uint32_t)(-ZLNG_DEC_E_FLAG); break; }               // src/libzling.cpp:315-317
            if (flag == 0) { closed = true; break; }
            if (p + 12 > n) break;                                   // header cut: block incomplete
            auto be32 = [&](uint64_t o) { return (uint32_t)z[o] << 24 | (uint32_t)z[o + 1] << 16 | (uint32_t)z[o + 2] << 8 | z[o + 3]; };
            const uint32_t encpos = be32(p), rlen = be32(p + 4), olen = be32(p + 8);
            p += 12;
            if (rlen > (uint32_t)kSubSyms || olen > (uint32_t)kPayloadMax) { pending = (uint32_t)(-ZLNG_DEC_E_BLOCKSIZE); break; }   // :326-328
            // a payload shorter than its own length tables: the reference would take the missing table bytes from whatever its
            // buffer held (:347-356); rejected
            if (olen < (uint32_t)kTableBytes) { pending = (uint32_t)(-ZLNG_DEC_E_LZ); break; }
            if (p + olen > n) break;                                 // payload cut: block incomplete
            // every u16 entry stands for at least one output byte, so a block with more entries than 16 Mi cannot land on its
            // encpos: the replay of an earlier sub-block fails first (lzdecode failed) -- the walk goes on only to see whether the
            // block is complete, and records nothing more
            if (!too_many && (tok_total - first_tok) + rlen > (uint64_t)kTokCapMax + 64) too_many = true;
            if (!too_many) {
                if (nsub >= a.max_subs) { err = (uint32_t)(-ZLNG_DEC_E_LIMIT); break; }
                // (an encpos beyond the block size fails in the replay, behind the sub-block's Huffman checks; it reserves no output)
                if (tok_total + rlen > a.tok_cap || (encpos <= (uint32_t)kBlockIn && out_total + encpos > a.out_cap)) { err = (uint32_t)(-ZLNG_DEC_E_LIMIT); break; }
                a.subs[nsub] = DecSub{p, tok_total, encpos, rlen, olen, nblk};
                nsub++;
                tok_total += rlen;
                last_encpos = encpos;
            }
            p += olen;
        }
        if (err || !(closed || pending)) { nsub = first_sub; break; }  // a resource limit, or the input ends inside the block: dropped
        if (too_many) pending = (uint32_t)(-ZLNG_DEC_E_LZ);
        if (pending) {
            err = pending;
            if (nsub > first_sub) { a.blocks[nblk] = DecBlock{first_sub, nsub - first_sub, out_total, 0, pending, p}; nblk++; }
            break;
        }
        a.blocks[nblk] = DecBlock{first_sub, nsub - first_sub, out_total, last_encpos, 0, p};
        out_total += last_encpos;
        nblk++;
        used = p;
    }
    a.summary[0] = used;
    a.summary[1] = err;
    a.summary[2] = nblk;
    a.summary[3] = nsub;
    a.summary[4] = out_total;
    a.summary[5] = nblk;
    a.summary[6] = 0;
}

// ------------------------------------------------------------------------------ K8 Huffman decode
// One wavefront per sub-block.  The 64 lanes build the 2^15-entry LUT of alphabet 1 and the 2^8
// LUT of alphabet 2 in LDS, then stream the payload through a 256-byte register window (lane l
// holds dword l; the next window is already in flight) while the symbol loop runs wave-uniformly.
constexpr int kLut1Bits = kMaxLen1, kLut2Bits = kMaxLen2;

// Per alphabet, for the exact table build below: symbols of every length in symbol order (`sorted`, lengths ascending), where each
// length's run starts (`off`), how many it holds (`cnt`) and the canonical code of its first symbol (`start`, not reduced mod 2^l).
struct LenClasses { uint32_t off[16], cnt[16], start[16]; };

// Returns true when the length set is OVER-SUBSCRIBED (Kraft sum above one): codes then collide in the decode table, and which
// symbol an entry ends up with is defined by the reference's fill ORDER (below).
__device__ bool build_codes_lane0(const uint8_t* len, uint16_t* code, int n, int limit, LenClasses* lc, uint16_t* sorted) {   // src/libzling_huffman.cpp:114-138
    uint32_t next[16];
    uint32_t count[16];
    for (int l = 0; l < 16; l++) count[l] = 0;
    for (int i = 0; i < n; i++) count[len[i]]++;
    uint32_t c = 0, o = 0, kraft = 0;
    for (int l = 1; l <= limit; l++) {
        next[l] = c; lc->start[l] = c; lc->cnt[l] = count[l]; lc->off[l] = o;
        o += count[l];
        kraft += count[l] << (limit - l);
        c = (c + count[l]) * 2;
    }
    uint32_t fill[16];
    for (int l = 1; l <= limit; l++) fill[l] = lc->off[l];
    for (int i = 0; i < n; i++) {
        const uint32_t l = len[i];
        // lengths above the limit (alphabet 2's nibbles go up to 15) get no code and no table entry (huffman.cpp:119-126, 146)
        code[i] = (l && l <= (uint32_t)limit) ? (uint16_t)((__brev(next[l]++) >> 16 & 0xFFFFu) >> (16 - l)) : (uint16_t)0;
        if (l && l <= (uint32_t)limit) sorted[fill[l]++] = (uint16_t)i;
    }
    return kraft > (1u << limit);
}

// The decode-table entry of index i as the reference's fill leaves it (src/libzling_huffman.cpp:140-153): symbols are entered in
// ascending order and a later one overwrites an earlier one, so an entry belongs to the LARGEST symbol whose code matches; and the
// symbol loop asks the 2^`fast`-entry table of the codes up to `fast` bits first (src/libzling.cpp:361, 376-379), so any such code
// beats every longer one.  A symbol of length L matches index i when its code, bit-reversed, equals i mod 2^L; the codes of one
// length are consecutive integers from start[L] (only their low L bits survive the 16-bit reversal, huffman.cpp:128-135), so the
// matching symbols of a length are every 2^L-th of its run and the largest is computed, not searched.
__device__ __forceinline__ uint32_t lut_entry_exact(uint32_t i, const LenClasses& lc, const uint16_t* sorted, int limit, int fast) {
    int best_fast = -1, best_slow = -1;
    for (int L = 1; L <= limit; L++) {
        const uint32_t cnt = lc.cnt[L];
        if (!cnt) continue;
        const uint32_t m = (1u << L) - 1u;
        const uint32_t q = __brev(i & m) >> (32 - L);
        const uint32_t k0 = (q - lc.start[L]) & m;
        if (k0 >= cnt) continue;
        const int c = (int)sorted[lc.off[L] + k0 + (((cnt - 1u - k0) >> L) << L)];
        if (L <= fast) best_fast = c > best_fast ? c : best_fast; else best_slow = c > best_slow ? c : best_slow;
    }
    return best_fast >= 0 ? (uint32_t)best_fast : (best_slow >= 0 ? (uint32_t)best_slow : 0xFFFFu);
}

__global__ __launch_bounds__(64) void k_huff_decode(DecodeArgs a) {
    __shared__ uint16_t lut1[1 << kLut1Bits];
    __shared__ uint16_t lut2[1 << kLut2Bits];
    __shared__ uint8_t  len[kNsymAll + 2];
    __shared__ uint16_t code[kNsymAll];
    __shared__ uint16_t sorted[kNsymAll];
    __shared__ LenClasses lc1, lc2;
    __shared__ uint32_t over[2];
    const uint32_t s = blockIdx.x;
    if (s >= (uint32_t)a.summary[3]) return;
    const DecSub sb = a.subs[s];
    const uint8_t* pay = a.z + sb.payload_off;
    const uint32_t lane = threadIdx.x;

    // nibble tables (src/libzling.cpp:347-356)
    for (uint32_t j = lane; j < (uint32_t)kTableBytes; j += 64) {
        const uint32_t b = pay[j];
        const uint32_t s0 = j < 257 ? 2 * j : kNsym1 + 2 * (j - 257);
        len[s0] = (uint8_t)(b >> 4);
        len[s0 + 1] = (uint8_t)(b & 15);
    }
    for (uint32_t i = lane; i < (1u << kLut1Bits); i += 64) lut1[i] = 0xFFFF;
    for (uint32_t i = lane; i < (1u << kLut2Bits); i += 64) lut2[i] = 0xFFFF;
    __syncthreads();
    if (lane == 0) {
        over[0] = build_codes_lane0(len, code, kNsym1, kMaxLen1, &lc1, sorted);
        over[1] = build_codes_lane0(len + kNsym1, code + kNsym1, kNsym2, kMaxLen2, &lc2, sorted + kNsym1);
    }
    __syncthreads();
    // ZlingMakeDecodeTable (src/libzling_huffman.cpp:140-153).  A complete or under-subscribed length set gives prefix-free codes:
    // no two symbols share a table entry and the symbols fill theirs side by side.  An over-subscribed one (hostile streams only)
    // makes entries collide, and every entry is computed instead, as the reference's fill order and two-level lookup define it.
    if (!over[0]) {
        for (uint32_t c = lane; c < (uint32_t)kNsym1; c += 64) {
            const uint32_t l = len[c];
            if (l > 0 && l <= (uint32_t)kLut1Bits) for (uint32_t i = code[c]; i < (1u << kLut1Bits); i += 1u << l) lut1[i] = (uint16_t)c;
        }
    } else {
        for (uint32_t i = lane; i < (1u << kLut1Bits); i += 64) lut1[i] = (uint16_t)lut_entry_exact(i, lc1, sorted, kMaxLen1, kMaxLen1Fast);
    }
    if (!over[1]) {
        for (uint32_t c = lane; c < (uint32_t)kNsym2; c += 64) {
            const uint32_t l = len[kNsym1 + c];
            if (l > 0 && l <= (uint32_t)kLut2Bits) for (uint32_t i = code[kNsym1 + c]; i < (1u << kLut2Bits); i += 1u << l) lut2[i] = (uint16_t)c;
        }
    } else {
        for (uint32_t i = lane; i < (1u << kLut2Bits); i += 64) lut2[i] = (uint16_t)lut_entry_exact(i, lc2, sorted + kNsym1, kMaxLen2, kMaxLen2);
    }
    __syncthreads();

    // bitstream starts right after the tables; bytes past olen read as zero (the reference reads
    // whatever follows in obuf; a valid stream never consumes them)
    const uint32_t nbytes = sb.olen - kTableBytes;
    const uint8_t* bits = pay + kTableBytes;
    auto load_win = [&](uint32_t w) -> uint32_t {                    // dword (64 w + lane) of the bitstream
        const uint32_t o = (w * 64 + lane) * 4;
        uint32_t v = 0;
        if (o + 4 <= nbytes) __builtin_memcpy(&v, bits + o, 4);
        else for (uint32_t k = 0; k < 4; k++) if (o + k < nbytes) v |= (uint32_t)bits[o + k] << (8 * k);
        return v;
    };
    uint32_t win = load_win(0), win_next = load_win(1);
    uint32_t widx = 0;                                               // next dword to pull (wave-uniform)
    uint64_t acc = 0;
    int nb = 0;
    uint32_t* tok = a.tok + sb.tok_off;
    uint32_t nt = 0, tokv = 0, err = 0, outlen = 0;
    for (uint32_t i = 0; i < sb.rlen; i++) {
        if (nb < 32) {                                               // src/libzling.cpp:369-374
            const uint32_t w = (uint32_t)__builtin_amdgcn_readlane((int)win, (int)(widx & 63));
            acc |= (uint64_t)w << nb;
            nb += 32;
            widx++;
            if ((widx & 63) == 0) { win = win_next; win_next = load_win((widx >> 6) + 1); }
        }
        const uint32_t sym = lut1[(uint32_t)acc & ((1u << kLut1Bits) - 1)];
        if (sym >= (uint32_t)kNsym1) { err = (uint32_t)(-ZLNG_DEC_E_CODE1); break; }
        const uint32_t l1 = len[sym];
        acc >>= l1; nb -= (int)l1;
        // a literal's token word carries its swap partner mtf_next(rank) in bits 16..23, so that the replay (one wavefront
        // for the whole stream) does not have to compute it; the sub-block's decoded length is summed here for the same reason
        uint32_t t = sym < 256 ? sym | mtf_next(sym) << 16 : sym;
        outlen += sym < 256 ? 1u : (sym < 258 ? 2u : sym - 258u + (uint32_t)kMatchMin);
        if (sym >= 258) {                                            // src/libzling.cpp:386-401
            const uint32_t c = lut2[(uint32_t)acc & ((1u << kLut2Bits) - 1)];
            if (c >= (uint32_t)kNsym2) { err = (uint32_t)(-ZLNG_DEC_E_CODE2); break; }
            const uint32_t l2 = len[kNsym1 + c];
            acc >>= l2; nb -= (int)l2;
            const uint32_t bl = matchidx_blen_of_code(c);
            const uint32_t ex = (uint32_t)acc & ((1u << bl) - 1u);
            acc >>= bl; nb -= (int)bl;
            const uint32_t base = c < 4 ? c : (c < 18 ? (2u + (c & 1u)) << ((c >> 1) - 1) : (c - 16) << 8);
            const uint32_t idx = base + ex;
            // (a match symbol in the LAST counted entry keeps its index: the reference stores it at tbuf[rlen] and its replay reads
            //  it there, src/libzling.cpp:398, src/libzling_lz.cpp:355-356; the largest index the 32 codes name is 3840 + 255)
            if (idx >= (uint32_t)kRing) { err = (uint32_t)(-ZLNG_DEC_E_EXBITS); break; }
            t |= idx << 16;
            i++;                                                     // a match occupies two u16 entries
        }
        tokv = lane == (nt & 63) ? t : tokv;
        nt++;
        if ((nt & 63) == 0) tok[nt - 64 + lane] = tokv;
    }
    if (lane < (nt & 63)) tok[(nt & ~63u) + lane] = tokv;
    // a failed sub-block is marked in its token count; the replay, which walks the stream in order, stops at the first one
    if (lane == 0) { a.sub_ntok[s] = err ? (0x80000000u | err) : nt; a.sub_ntok[a.max_subs + s] = outlen; }
}

// ------------------------------------------------------------------------------ K9, hand-written token loop
// The same replay with the token loop of a sub-block written out by hand.  A lone wavefront issues one instruction every
// ~5 cycles whatever it is, so the replay is bound by the instructions per token, not by bytes: the compiler's form above
// spends ~100 per literal (it keeps every wave-uniform value in scalar registers and pays a v_readfirstlane and an exec
// save/restore around every lane-0 store); this one runs the loop with EXEC = lane 0, keeps the context-derived addresses
// in vector registers of that lane (an LDS read feeds the next LDS address without leaving the vector unit) and the
// token-derived values (type, length, ring index: data-independent) in scalar registers, so that every branch is a scalar
// branch.  Per literal: 33 instructions.  A match opens EXEC to 64 lanes for the copy only.
//   heads[c] holds the NEXT slot of context c (= head + 1 mod 4096): one ds_inc_rtn_u32 returns the slot to write and
//   steps the counter (src/libzling_lz.cpp:388-399 inserts before it looks up, so a ring index of 0 names the token itself:
//   rejected, like k_rolz_decode's src >= pos).
//   The per-token `pos + len > encpos` test of src/libzling_lz.cpp:336-369 is made once per sub-block: K8 sums the decoded
//   length of its tokens (lengths do not depend on decoded data), and a sub-block whose sum does not land on encpos fails
//   before a byte of it is written -- so the loop cannot write past the block.
//   The last 64 KiB of the block's output are mirrored in LDS (position p at window[p & 0xFFFF]): 71 % of the benchmark
//   text's match sources lie that near, and a match served from the window waits for one LDS access instead of a second
//   memory round trip behind the ring slot.  Every output byte is written to both.
// LDS layout (the array must sit at LDS address 0; the kernel traps otherwise): mtf[256][256] u8 @0, window @0x10000,
// mru[256] u32 @0x20000, heads[256] u32 @0x20400.
constexpr uint32_t kLdsMtf = 0, kLdsWin = 0x10000, kLdsMru = 0x20000, kLdsHeads = 0x20400, kLdsReplay = 0x20800;

__device__ __forceinline__ void replay_tokens(uint8_t* out, uint32_t* ring, const uint32_t* tok, uint32_t nt, uint32_t ti0,
                                              uint32_t lane, uint32_t& opos, uint32_t& b1, uint32_t& b2, uint32_t& err) {
    uint32_t o_opos, o_b1, o_b2, o_err;
    auto uni64 = [](const void* p) {                                  // the "s" constraint wants values the compiler knows to be uniform
        const uint64_t v = (uint64_t)(uintptr_t)p;
        return (uint64_t)(uint32_t)__builtin_amdgcn_readfirstlane((int)(uint32_t)v) |
               (uint64_t)(uint32_t)__builtin_amdgcn_readfirstlane((int)(uint32_t)(v >> 32)) << 32;
    };
    const uint64_t p_out = uni64(out), p_ring = uni64(ring), p_tok = uni64(tok);
    // The loop itself is generated (scripts/gen_replay_asm.py -> replay_loop.h): it is software-pipelined over two register
    // sets, so that the LDS reads of token t+1 are in flight while token t is written out.  Shared registers: s[40:41] out,
    // s[42:43] ring, s[44:45] next token window, s46 nt, s47 index of the next token to fetch, s48 end of the window held in
    // v22 (v23: the one after), s49 pos, s53 match length, s54 source, s55 distance, s[60:61] EXEC at entry, s62 &mru[],
    // s63 error, s68 &window; v20 lane, v21 lane*4, v37 pos (lane 0), v38 4095, v45 0xFFFF.  Per set (A / B): token, symbol,
    // ring index or swap partner (s50-52 / s70-72); &mru[b1] (heads[b1] at +1024), &mru[b2], table row b1*256, ring row
    // b1<<14, b1, ring slot, the two table addresses and entries, the MRU pair (v24-29, v31-35 / v54-59, v61-65).
    asm volatile(ZLNG_REPLAY_LOOP_ASM
        : [o_opos] "=s"(o_opos), [o_b1] "=s"(o_b1), [o_b2] "=s"(o_b2), [o_err] "=s"(o_err)
        : [out] "s"(p_out), [ring] "s"(p_ring), [tok] "s"(p_tok), [nt] "s"(nt), [ti] "s"(ti0), [opos] "s"(opos),
          [lane] "v"(lane), [b1] "s"(b1), [b2] "s"(b2)
        : "memory", "vcc", "scc",
          "s40", "s41", "s42", "s43", "s44", "s45", "s46", "s47", "s48", "s49", "s50", "s51", "s52", "s53", "s54", "s55", "s56", "s57", "s58", "s59",
          "s60", "s61", "s62", "s63", "s64", "s65", "s66", "s67", "s68", "s70", "s71", "s72",
          "v20", "v21", "v22", "v23", "v24", "v25", "v26", "v27", "v28", "v29", "v30", "v31", "v32", "v33", "v34", "v35", "v36", "v37", "v38", "v39",
          "v40", "v41", "v42", "v43", "v44", "v45", "v46", "v47", "v48", "v49", "v50", "v54", "v55", "v56", "v57", "v58", "v59", "v61", "v62", "v63", "v64", "v65");
    opos = o_opos; b1 = o_b1; b2 = o_b2; err = o_err;
}

__global__ __launch_bounds__(64) void k_rolz_replay(DecodeArgs a) {
    __shared__ __attribute__((aligned(16))) uint8_t lds[kLdsReplay];
    uint32_t* mru = reinterpret_cast<uint32_t*>(lds + kLdsMru);
    uint32_t* heads = reinterpret_cast<uint32_t*>(lds + kLdsHeads);
    uint8_t* mtf = lds + kLdsMtf;
    uint8_t* win = lds + kLdsWin;
    const uint32_t lane = threadIdx.x;
    auto ufl = [](uint32_t v) { return (uint32_t)__builtin_amdgcn_readfirstlane((int)v); };
    if ((uint32_t)(uintptr_t)lds != 0) __builtin_trap();               // the token loop addresses LDS absolutely
    const uint32_t nblk = (uint32_t)a.summary[2];
    for (uint32_t i = lane; i < 256 * 256 / 4; i += 64) reinterpret_cast<uint32_t*>(mtf)[i] = reinterpret_cast<const uint32_t*>(a.mtf_state)[i];
    __syncthreads();
    uint32_t err = 0;
    for (uint32_t b = 0; b < nblk && !err; b++) {
        const DecBlock bk = a.blocks[b];
        uint8_t* out = a.out + bk.out_off;
        for (uint32_t i = lane; i < 256 * 256 / 4; i += 64) reinterpret_cast<uint32_t*>(a.mtf_snap)[i] = reinterpret_cast<const uint32_t*>(mtf)[i];
        for (uint32_t i = lane; i < 256u * kRing; i += 64) a.ring[i] = 0;  // Reset(), src/libzling_lz.cpp:378-386
        for (uint32_t i = lane; i < 256; i += 64) heads[i] = 1;            // next slot = head + 1
        // the token loop reads ring slots with scalar loads (they do not queue behind the stores in flight) and only bounds
        // how many vector operations may still be outstanding: the reset must have landed before it starts
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __syncthreads();
        uint32_t opos = 0;
        for (uint32_t k = 0; k < bk.nsub && !err; k++) {
            const DecSub sb = a.subs[bk.first_sub + k];
            const uint32_t* tok = a.tok + sb.tok_off;
            const uint32_t nt = ufl(a.sub_ntok[bk.first_sub + k]);
            if (nt & 0x80000000u) { err = nt & 0xFFFFu; break; }           // K8 rejected this sub-block's bitstream
            // an encpos beyond the block size: the reference's replay would write past its buffer before its size test
            // (src/libzling_lz.cpp:363-373); rejected here, behind the Huffman checks like every replay error
            if (sb.encpos > (uint32_t)kBlockIn) { err = (uint32_t)(-ZLNG_DEC_E_LZ); break; }
            // The block's two opening entries are copied as raw bytes (src/libzling_lz.cpp:327-328): a word symbol there is the byte
            // 0 or 1 (the u16 entry truncated) and stands for ONE byte, not the two K8 counted.  A match symbol there would split
            // its (symbol, index) pair and re-read the index as a symbol -- lengths up to 3,841 past the reference's sentinel:
            // rejected (decoder deviations).
            uint32_t outlen = ufl(a.sub_ntok[a.max_subs + bk.first_sub + k]);
            for (uint32_t q = opos, t2 = 0; q < 2 && t2 < nt; q++, t2++) {
                const uint32_t sym = ufl(tok[t2]) & 0xFFFFu;
                if (sym >= 258u) { err = (uint32_t)(-ZLNG_DEC_E_LZ); break; }
                if (sym >= 256u) outlen -= 1u;
            }
            if (err) break;
            if (opos + outlen != sb.encpos) { err = (uint32_t)(-ZLNG_DEC_E_LZ); break; }
            for (uint32_t i = lane; i < 256; i += 64) mru[i] = 0;
            __syncthreads();
            uint32_t ti = 0;
            while (opos < 2 && ti < nt) {
                const uint32_t v = ufl(tok[ti++]);
                if (lane == 0) { out[opos] = (uint8_t)v; win[opos] = (uint8_t)v; }
                opos++;
            }
            if (!err && ti < nt) {
                uint32_t b1 = ufl(out[opos - 1]), b2 = ufl(out[opos - 2]), bad = 0, op = ufl(opos);
                replay_tokens(out, a.ring, tok, nt, ufl(ti), lane, op, b1, b2, bad);
                opos = op;
                if (bad) err = (uint32_t)(-ZLNG_DEC_E_LZ);
            }
            if (!err && opos != sb.encpos) err = (uint32_t)(-ZLNG_DEC_E_LZ);   // src/libzling_lz.cpp:371-373
        }
        if (!err && bk.pad) err = bk.pad;                                  // a probe block: its sub-blocks passed, so the framing error behind them stands
        if (err && lane == 0) { a.summary[5] = b; a.summary[6] = err; }
    }
    __syncthreads();
    const uint8_t* keep = err ? a.mtf_snap : mtf;                    // a failed block leaves the tables as it found them
    __threadfence_block();
    for (uint32_t i = lane; i < 256 * 256 / 4; i += 64) reinterpret_cast<uint32_t*>(a.mtf_state)[i] = reinterpret_cast<const uint32_t*>(keep)[i];
}

void launch_frame_walk(const DecodeArgs& a, hipStream_t s) { hipLaunchKernelGGL(k_frame_walk, dim3(1), dim3(64), 0, s, a); }
void launch_huff_decode(const DecodeArgs& a, uint32_t nsubs_upper, hipStream_t s) {
    hipLaunchKernelGGL(k_huff_decode, dim3(nsubs_upper), dim3(64), 0, s, a);
}
void launch_rolz_decode(const DecodeArgs& a, hipStream_t s) { hipLaunchKernelGGL(k_rolz_replay, dim3(1), dim3(64), 0, s, a); }

}  // namespace zlng
