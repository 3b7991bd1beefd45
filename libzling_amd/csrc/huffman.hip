// huffman.hip -- K3 histogram, K4 code lengths + canonical codes, K5 layout scan, K6 bit-pack + framing.
//
// Replaces, per sub-block, the Huffman half of the reference's block driver:
//   histogram loop                 src/libzling.cpp:219-224
//   ZlingMakeLengthTable x2        src/libzling_huffman.cpp:41-112   (libstdc++ heap order)
//   ZlingMakeEncodeTable x2        src/libzling_huffman.cpp:114-138
//   nibble tables + ZlingCodebuf   src/libzling.cpp:232-258
//   framing                        src/libzling.cpp:200, 269-279
// Sub-blocks are independent here, so the grid is (sub-block, block).
#include "zlng_common.h"
#include "zlng_kernels.h"
#include <cstdlib>

namespace zlng {

__device__ __forceinline__ uint32_t sub_index(uint32_t blk, uint32_t sub) { return blk * kMaxSub + sub; }

// ------------------------------------------------------------------------------ K3 histogram
// 256 lanes stream the sub-block's token words (coalesced) and count into LDS.
__global__ __launch_bounds__(256) void k_histogram(HuffArgs a) {
    __shared__ uint32_t h[kNsymAll];
    const uint32_t blk = blockIdx.y + a.blk0, sub = blockIdx.x;
    if (sub >= a.nsub[blk]) return;
    for (uint32_t i = threadIdx.x; i < kNsymAll; i += 256) h[i] = 0;
    __syncthreads();
    const SubCut c = a.cuts[sub_index(blk, sub)];
    const uint32_t* t = a.tok + (size_t)blk * a.tok_cap;
    for (uint32_t i = c.tok_begin + threadIdx.x; i < c.tok_end; i += 256) {
        const uint32_t v = t[i], sym = v & 0xFFFF;
        atomicAdd(&h[sym], 1u);
        if (sym >= 258) {
            uint32_t code, bl, ex;
            matchidx_split(v >> 16, code, bl, ex);
            atomicAdd(&h[kNsym1 + code], 1u);
        }
    }
    __syncthreads();
    uint32_t* f = a.freq + (size_t)sub_index(blk, sub) * kNsymAll;
    for (uint32_t i = threadIdx.x; i < kNsymAll; i += 256) f[i] = h[i];
}

// ------------------------------------------------------------------------------ K4 lengths + codes
// The code lengths must reproduce std::priority_queue's tie order exactly (the comparator has no
// tie-break key, src/libzling_huffman.cpp:63-67), so the libstdc++ binary-heap primitives
// (GCC 11 bits/stl_heap.h: __push_heap :134-147, __adjust_heap :223-248, __pop_heap :253-265,
// __make_heap :339-360) are restated over an LDS array.  A heap entry packs (weight << 11 | node)
// into 64 bits so one LDS read yields both (any 32-bit weight sum is exact, not only the <= 262,144
// symbols of a real sub-block); comparisons look at the weight only.
// The build is inherently serial (one lane); the wavefront's other lanes help with the
// embarrassingly parallel parts (leaf setup, depth -> length, totals).
constexpr int kHeapNodeBits = 11;                    // node ids < 1027 < 2048
typedef unsigned long long HeapEnt;
__device__ __forceinline__ uint32_t hw(HeapEnt e) { return (uint32_t)(e >> kHeapNodeBits); }

__device__ __forceinline__ void heap_sift_up(HeapEnt* h, int hole, int top, HeapEnt v) {
    int parent = (hole - 1) / 2;
    while (hole > top && hw(h[parent]) > hw(v)) { h[hole] = h[parent]; hole = parent; parent = (hole - 1) / 2; }
    h[hole] = v;
}
__device__ __forceinline__ void heap_adjust(HeapEnt* h, int hole, int len, HeapEnt v) {
    const int top = hole;
    int kid = hole;
    while (kid < (len - 1) / 2) {
        kid = 2 * (kid + 1);
        const HeapEnt r = h[kid], l = h[kid - 1];
        if (hw(r) > hw(l)) { kid--; h[hole] = l; } else { h[hole] = r; }
        hole = kid;
    }
    if ((len & 1) == 0 && kid == (len - 2) / 2) {
        kid = 2 * (kid + 1);
        h[hole] = h[kid - 1];
        hole = kid - 1;
    }
    heap_sift_up(h, hole, top, v);
}

// Builds lengths for one alphabet.  freq/len are LDS arrays of n entries.  Runs on lane 0.
__device__ void build_lengths_lane0(const uint32_t* freq, uint8_t* len, int n, int limit, HeapEnt* heap,
                                    uint16_t* kid0, uint16_t* kid1, uint16_t* leafsym, uint8_t* depth) {
    for (int i = 0; i < n; i++) len[i] = 0;
    for (int scaling = 0;; scaling++) {
        int nn = 0;
        for (int i = 0; i < n; i++) {
            const uint32_t f = freq[i];
            if (f > 0) {
                const uint32_t w = (f + ((1u << scaling) - 1)) >> scaling;
                leafsym[nn] = (uint16_t)i;
                heap[nn] = (HeapEnt)w << kHeapNodeBits | (uint32_t)nn;
                nn++;
            }
        }
        if (nn == 0) return;
        const int nleaf = nn;
        int hn = nn;
        if (hn >= 2) for (int p = (hn - 2) / 2; p >= 0; p--) { const HeapEnt v = heap[p]; heap_adjust(heap, p, hn, v); }
        while (hn > 1) {
            const HeapEnt e1 = heap[0];
            { const HeapEnt v = heap[hn - 1]; heap[hn - 1] = e1; heap_adjust(heap, 0, hn - 1, v); hn--; }
            const HeapEnt e2 = heap[0];
            if (hn > 1) { const HeapEnt v = heap[hn - 1]; heap[hn - 1] = e2; heap_adjust(heap, 0, hn - 1, v); }
            hn--;
            kid0[nn] = (uint16_t)(e1 & ((1u << kHeapNodeBits) - 1));
            kid1[nn] = (uint16_t)(e2 & ((1u << kHeapNodeBits) - 1));
            const HeapEnt e = (HeapEnt)(hw(e1) + hw(e2)) << kHeapNodeBits | (uint32_t)nn;
            heap_sift_up(heap, hn, 0, e);
            hn++;
            nn++;
        }
        // children are always created before their parent: walk from the root down
        const int root = nn - 1;
        depth[root] = 0;
        for (int v = root; v >= nleaf; v--) { const uint8_t d = depth[v] + 1; depth[kid0[v]] = d; depth[kid1[v]] = d; }
        int maxlen = 0;
        for (int v = 0; v < nleaf; v++) {
            const int l = depth[v] > 1 ? depth[v] : 1;
            len[leafsym[v]] = (uint8_t)l;
            maxlen = l > maxlen ? l : maxlen;
        }
        if (maxlen <= limit) return;
    }
}

// ZlingMakeEncodeTable (src/libzling_huffman.cpp:114-138): canonical codes, length-major /
// symbol-minor, then 16-bit reversed and shifted so they can be emitted LSB-first.  Runs on lane 0.
__device__ void assign_codes_lane0(const uint8_t* len, uint16_t* code, int n, int limit) {
    uint32_t count[16], next[16];
    for (int l = 0; l < 16; l++) count[l] = 0;
    for (int i = 0; i < n; i++) count[len[i]]++;
    uint32_t c = 0;
    for (int l = 1; l <= limit; l++) { next[l] = c; c = (c + count[l]) * 2; }
    for (int i = 0; i < n; i++) {
        const uint32_t l = len[i];
        uint32_t v = 0;
        if (l) v = (__brev(next[l]++) >> 16 & 0xFFFFu) >> (16 - l);
        code[i] = (uint16_t)v;
    }
}

__global__ __launch_bounds__(64) void k_lengths(HuffArgs a) {
    __shared__ uint32_t freq[kNsymAll];
    __shared__ uint8_t  len[kNsymAll + 2];
    __shared__ uint16_t code[kNsymAll];
    __shared__ HeapEnt heap[kNsym1];
    __shared__ uint16_t kid0[2 * kNsym1], kid1[2 * kNsym1], leafsym[kNsym1];
    __shared__ uint8_t  depth[2 * kNsym1];
    const uint32_t blk = blockIdx.y + a.blk0, sub = blockIdx.x;
    if (sub >= a.nsub[blk]) return;
    const uint32_t si = sub_index(blk, sub);
    const uint32_t lane = threadIdx.x;
    const uint32_t* f = a.freq + (size_t)si * kNsymAll;
    for (uint32_t i = lane; i < kNsymAll; i += 64) freq[i] = f[i];
    __syncthreads();
    if (lane == 0) {
        build_lengths_lane0(freq, len, kNsym1, kMaxLen1, heap, kid0, kid1, leafsym, depth);
        build_lengths_lane0(freq + kNsym1, len + kNsym1, kNsym2, kMaxLen2, heap, kid0, kid1, leafsym, depth);
        assign_codes_lane0(len, code, kNsym1, kMaxLen1);
        assign_codes_lane0(len + kNsym1, code + kNsym1, kNsym2, kMaxLen2);
    }
    __syncthreads();
    // payload size: 273 table bytes + ceil(bits / 8); bits are a dot product of freq and lengths
    uint64_t bits = 0;
    for (uint32_t i = lane; i < kNsymAll; i += 64) {
        uint32_t l = len[i];
        if (i >= kNsym1) l += matchidx_blen_of_code(i - kNsym1);
        bits += (uint64_t)freq[i] * l;
    }
    for (int o = 32; o > 0; o >>= 1) bits += __shfl_down(bits, o);
    uint8_t* gl = a.lens + (size_t)si * kNsymAll;
    uint16_t* gc = a.codes + (size_t)si * kNsymAll;
    for (uint32_t i = lane; i < kNsymAll; i += 64) { gl[i] = len[i]; gc[i] = code[i]; }
    if (lane == 0) a.olen[si] = (uint32_t)(kTableBytes + (bits + 7) / 8);
}

// ------------------------------------------------------------------------------ K5 layout
// Exclusive scan of (13 + olen) over sub-blocks in stream order, +1 per block for the 0x00
// terminator (src/libzling.cpp:278).  Tiny: one workgroup.
__global__ __launch_bounds__(1024) void k_layout(HuffArgs a) {
    __shared__ uint64_t wave_tot[16];
    __shared__ uint64_t carry;
    __shared__ uint32_t err;
    const uint32_t tid = threadIdx.x;
    if (tid == 0) { carry = 0; err = 0; }
    __syncthreads();
    for (uint32_t b0 = 0; b0 < a.nblocks; b0 += 1024) {
        const uint32_t blk = b0 + tid;
        uint64_t size = 0;
        if (blk < a.nblocks) {
            const uint32_t ns = a.nsub[blk];
            if (ns > kMaxSub) atomicOr(&err, 2u);
            for (uint32_t s = 0; s < ns && s < kMaxSub; s++) {
                const uint32_t ol = a.olen[sub_index(blk, s)];
                if (ol > (uint32_t)kPayloadMax) atomicOr(&err, 1u);
                size += kHeaderBytes + ol;
            }
            size += 1;
        }
        // workgroup exclusive scan of `size`
        uint64_t incl = size;
        for (int o = 1; o < 64; o <<= 1) { uint64_t y = __shfl_up(incl, o); if ((tid & 63) >= (uint32_t)o) incl += y; }
        if ((tid & 63) == 63) wave_tot[tid >> 6] = incl;
        __syncthreads();
        uint64_t before = carry;
        for (uint32_t w = 0; w < (tid >> 6); w++) before += wave_tot[w];
        const uint64_t begin = before + incl - size;
        if (blk < a.nblocks) {
            uint64_t off = begin;
            const uint32_t ns = a.nsub[blk];
            for (uint32_t s = 0; s < ns && s < kMaxSub; s++) {
                a.sub_off[sub_index(blk, s)] = off;
                off += kHeaderBytes + a.olen[sub_index(blk, s)];
            }
            a.blk_end[blk] = off + 1;
        }
        __syncthreads();
        if (tid == 1023) carry = before + incl;
        __syncthreads();
    }
    if (tid == 0) {
        a.summary[0] = carry;
        a.summary[1] = err | (carry > a.out_cap ? 4u : 0u) | (*a.overflow == 1u ? 8u : 0u) | (*a.overflow >= 2u ? 2u : 0u);   // 1: a block ran out of token words; 2: the parser reported an internal fault
    }
}

// ------------------------------------------------------------------------------ K6 bit-pack + framing
// One workgroup per sub-block.  Its bytes form one contiguous run of the output:
//   [0x01][encpos][rlen][olen] [257+16 nibble-table bytes] [LSB-first bitstream] ([0x00] if last of block)
// The run is assembled in an LDS window of 32-bit words that is congruent to the OUTPUT's word
// grid (window word w <-> aligned global word), so whole words are stored with one dword store
// and only the run's first/last partial words fall back to byte stores (neighbouring sub-blocks
// share those words).  Bit positions come from a workgroup exclusive scan of code lengths; bits
// are deposited with LDS atomic OR (a token's <= 31 bits straddle at most two words).
constexpr int kPackThreads = 256;
constexpr int kPackPerThread = 4;
constexpr int kPackTile = kPackThreads * kPackPerThread;          // tokens per tile
constexpr int kPackWin = kPackTile + 16;                          // words: 31 bits/token < 1 word/token

__device__ __forceinline__ void win_put(uint32_t* win, uint64_t rel_bit, uint32_t bits, uint32_t nbits) {
    if (nbits == 0) return;
    const uint32_t w = (uint32_t)(rel_bit >> 5), sh = (uint32_t)(rel_bit & 31);
    const uint64_t v = (uint64_t)bits << sh;
    atomicOr(&win[w], (uint32_t)v);
    if (sh + nbits > 32) atomicOr(&win[w + 1], (uint32_t)(v >> 32));
}

// Store window words [0, nwords) to the output.  Word w covers global bytes [g0 + 4w, g0 + 4w + 4);
// only bytes inside [lo, hi) belong to this sub-block.
__device__ __forceinline__ void win_flush(const uint32_t* win, uint32_t nwords, uint8_t* out, uint64_t g0,
                                          uint64_t lo, uint64_t hi) {
    for (uint32_t w = threadIdx.x; w < nwords; w += kPackThreads) {
        const uint64_t g = g0 + 4ull * w;
        const uint32_t v = win[w];
        if (g >= lo && g + 4 <= hi) {
            *reinterpret_cast<uint32_t*>(out + g) = v;
        } else {
            for (int k = 0; k < 4; k++) if (g + k >= lo && g + k < hi) out[g + k] = (uint8_t)(v >> (8 * k));
        }
    }
}

__global__ __launch_bounds__(kPackThreads) void k_pack(HuffArgs a) {
    __shared__ uint32_t win[kPackWin];
    __shared__ uint16_t code[kNsymAll];
    __shared__ uint8_t  len[kNsymAll + 2];
    __shared__ uint32_t wave_tot[kPackThreads / 64];
    const uint32_t blk = blockIdx.y, sub = blockIdx.x;
    const uint32_t ns = a.nsub[blk];
    if (sub >= ns || a.summary[1] != 0) return;
    const uint32_t si = sub_index(blk, sub);
    const uint32_t tid = threadIdx.x;
    const SubCut c = a.cuts[si];
    const uint32_t olen = a.olen[si];
    const uint64_t lo = a.sub_off[si];
    const uint64_t hi = lo + kHeaderBytes + olen + (sub + 1 == ns ? 1 : 0);
    uint64_t g0 = lo & ~3ull;                                   // global byte address of win[0]

    for (uint32_t i = tid; i < kNsymAll; i += kPackThreads) {
        code[i] = a.codes[(size_t)si * kNsymAll + i];
        len[i] = a.lens[(size_t)si * kNsymAll + i];
    }
    if (tid == 0) { len[kNsymAll] = 0; len[kNsymAll + 1] = 0; }
    for (uint32_t i = tid; i < kPackWin; i += kPackThreads) win[i] = 0;
    __syncthreads();

    // header + nibble tables, deposited as 8-bit fields
    uint64_t cur = (lo - g0) * 8;                               // bit cursor relative to win[0]
    for (uint32_t i = tid; i < kHeaderBytes + kTableBytes; i += kPackThreads) {
        uint32_t b;
        if (i == 0) b = 1;
        else if (i < 5) b = (c.encpos >> (8 * (4 - i))) & 0xFF;
        else if (i < 9) b = (c.rlen >> (8 * (8 - i))) & 0xFF;
        else if (i < 13) b = (olen >> (8 * (12 - i))) & 0xFF;
        else {
            const uint32_t j = i - kHeaderBytes;                // 0..272; tables are 257 + 16 bytes
            const uint32_t s0 = j < 257 ? 2 * j : kNsym1 + 2 * (j - 257);
            b = (uint32_t)len[s0] * 16 + len[s0 + 1];
        }
        win_put(win, cur + 8ull * i, b, 8);
    }
    cur += 8ull * (kHeaderBytes + kTableBytes);
    __syncthreads();

    const uint32_t* t = a.tok + (size_t)blk * a.tok_cap;
    for (uint32_t base = c.tok_begin; ; base += kPackTile) {
        // flush completed words, slide the window
        cur = ((uint64_t)(uint32_t)__builtin_amdgcn_readfirstlane((int)(uint32_t)(cur >> 32)) << 32) |
              (uint32_t)__builtin_amdgcn_readfirstlane((int)(uint32_t)cur);
        const uint32_t nfull = (uint32_t)(cur >> 5);
        if (nfull) {
            win_flush(win, nfull, a.out, g0, lo, hi);
            __syncthreads();
            const uint32_t keep = win[nfull];
            __syncthreads();
            for (uint32_t i = tid; i < kPackWin; i += kPackThreads) win[i] = (i == 0) ? keep : 0;
            g0 += 4ull * nfull;
            cur &= 31;
            __syncthreads();
        }
        if (base >= c.tok_end) break;

        uint32_t bits[kPackPerThread], nb[kPackPerThread], mine = 0;
#pragma unroll
        for (int k = 0; k < kPackPerThread; k++) {
            const uint32_t i = base + tid * kPackPerThread + k;
            bits[k] = 0; nb[k] = 0;
            if (i < c.tok_end) {
                const uint32_t v = t[i], sym = v & 0xFFFF;
                uint32_t b = code[sym], n = len[sym];
                if (sym >= 258) {
                    uint32_t cd, bl, ex;
                    matchidx_split(v >> 16, cd, bl, ex);
                    b |= (uint32_t)code[kNsym1 + cd] << n; n += len[kNsym1 + cd];
                    b |= ex << n; n += bl;
                }
                bits[k] = b; nb[k] = n;
            }
            mine += nb[k];
        }
        uint32_t incl = mine;
        for (int o = 1; o < 64; o <<= 1) { uint32_t y = __shfl_up(incl, o); if ((tid & 63) >= (uint32_t)o) incl += y; }
        if ((tid & 63) == 63) wave_tot[tid >> 6] = incl;
        __syncthreads();
        uint32_t before = 0, total = 0;
        for (uint32_t w = 0; w < kPackThreads / 64; w++) { if (w < (tid >> 6)) before += wave_tot[w]; total += wave_tot[w]; }
        uint64_t p = cur + before + incl - mine;
#pragma unroll
        for (int k = 0; k < kPackPerThread; k++) { win_put(win, p, bits[k], nb[k]); p += nb[k]; }
        cur += total;
        __syncthreads();
    }
    // tail: partial last word (zero padded) and, for the block's last sub-block, the 0x00 terminator
    const uint32_t tail_words = (uint32_t)((hi - g0 + 3) / 4);
    win_flush(win, tail_words, a.out, g0, lo, hi);
}

void launch_histogram(const HuffArgs& a, hipStream_t s) {
    hipLaunchKernelGGL(k_histogram, dim3(kMaxSub, a.nblocks - a.blk0), dim3(256), 0, s, a);
}
void launch_lengths(const HuffArgs& a, hipStream_t s) {
    hipLaunchKernelGGL(k_lengths, dim3(kMaxSub, a.nblocks - a.blk0), dim3(64), 0, s, a);
}
void launch_layout(const HuffArgs& a, hipStream_t s) {
    hipLaunchKernelGGL(k_layout, dim3(1), dim3(1024), 0, s, a);
}
void launch_pack(const HuffArgs& a, hipStream_t s) {
    const char* e = getenv("ZLNG_DEBUG_PACK_LDS");
    const unsigned dyn = e ? (unsigned)atoi(e) : 0u;
    hipLaunchKernelGGL(k_pack, dim3(kMaxSub, a.nblocks), dim3(kPackThreads), dyn, s, a);
}

}  // namespace zlng
