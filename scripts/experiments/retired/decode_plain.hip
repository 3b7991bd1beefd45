// Retired in round 4 (not built): the compiler-scheduled replay loop of rounds 1-2 (ZLNG_DEC=plain), cut out of csrc/decode.hip.
// ------------------------------------------------------------------------------ K9 ROLZ + MTF replay
// One wavefront per stream.  Control flow is wave-uniform (every lane follows the same token);
// lane 0 performs the scalar stores, all lanes share the match copies.
__global__ __launch_bounds__(64) void k_rolz_decode(DecodeArgs a) {
    __shared__ uint8_t  mtf[256 * 256];
    __shared__ uint32_t mru[256];
    __shared__ uint16_t heads[256];
    const uint32_t lane = threadIdx.x;
    const uint32_t nblk = (uint32_t)a.summary[2];
    for (uint32_t i = lane; i < 256 * 256 / 4; i += 64) reinterpret_cast<uint32_t*>(mtf)[i] = reinterpret_cast<const uint32_t*>(a.mtf_state)[i];
    __syncthreads();
    uint32_t err = 0;
    uint32_t* ring = a.ring;                                          // [256][4096] source positions
    for (uint32_t b = 0; b < nblk && !err; b++) {
        const DecBlock bk = a.blocks[b];
        uint8_t* out = a.out + bk.out_off;
        // tables at the start of this block: if it fails, they are what the context keeps (the blocks before it are good and
        // are reported; the caller meets the error again at the head of its next call, as the reference's loop would)
        for (uint32_t i = lane; i < 256 * 256 / 4; i += 64) reinterpret_cast<uint32_t*>(a.mtf_snap)[i] = reinterpret_cast<const uint32_t*>(mtf)[i];
        for (uint32_t i = lane; i < 256u * kRing; i += 64) ring[i] = 0;    // Reset(), src/libzling_lz.cpp:378-386
        for (uint32_t i = lane; i < 256; i += 64) heads[i] = 0;
        __syncthreads();
        uint32_t opos = 0;
        for (uint32_t k = 0; k < bk.nsub && !err; k++) {
            const DecSub sb = a.subs[bk.first_sub + k];
            const uint32_t* tok = a.tok + sb.tok_off;
            const uint32_t nt = a.sub_ntok[bk.first_sub + k];
            if (nt & 0x80000000u) { err = nt & 0xFFFFu; break; }           // K8 rejected this sub-block's bitstream
            for (uint32_t i = lane; i < 256; i += 64) mru[i] = 0;
            __syncthreads();
            uint32_t ti = 0;
            // Every value that steers the loop is wave-uniform (all lanes replay the same token); readfirstlane
            // says so to the compiler, which keeps opos / ti / the branch conditions in scalar registers.
            // Token words are pulled 64 at a time into a register window (lane l holds word base + l).
            auto ufl = [](uint32_t v) { return (uint32_t)__builtin_amdgcn_readfirstlane((int)v); };
            uint32_t twin = lane < nt ? tok[lane] : 0u, tbase = 0;
            auto next_token = [&]() -> uint32_t {
                if (ti - tbase >= 64u) { tbase += 64u; twin = tbase + lane < nt ? tok[tbase + lane] : 0u; }
                return (uint32_t)__builtin_amdgcn_readlane((int)twin, (int)(ti++ - tbase));
            };
            // first two bytes of a block are raw (src/libzling_lz.cpp:327-328)
            while (opos < 2 && ti < nt) {
                const uint32_t v = next_token();
                if ((v & 0xFFFF) >= 256 || opos + 1 > sb.encpos) { err = (uint32_t)(-ZLNG_DEC_E_LZ); break; }
                if (lane == 0) out[opos] = (uint8_t)v;
                opos++;
            }
            // the last three bytes written are tracked in scalars (b3 = out[opos-3], b2, b1 = out[opos-1]);
            // only a match has to look them up again
            uint32_t b1 = 0, b2 = 0, b3 = 0;
            if (opos >= 1) b1 = ufl(out[opos - 1]);
            if (opos >= 2) b2 = ufl(out[opos - 2]);
            if (opos >= 3) b3 = ufl(out[opos - 3]);
            while (ti < nt && !err) {
                const uint32_t v = next_token(), sym = v & 0xFFFF;
                const uint32_t c1 = b1;                               // order-1 context
                // GetMatchAndUpdate: every token inserts its start position (src/libzling_lz.cpp:388-399)
                const uint32_t head = (ufl(heads[c1]) + 1u) & (kRing - 1);
                uint32_t* r = ring + c1 * kRing;
                if (lane == 0) { heads[c1] = (uint16_t)head; r[head] = opos; }
                if (sym < 256) {                                      // literal: ZlingMTFDecoder::Decode :122-126
                    if (opos + 1 > sb.encpos) { err = (uint32_t)(-ZLNG_DEC_E_LZ); break; }
                    uint8_t* t = mtf + c1 * 256;
                    const uint32_t nx = mtf_next(sym);
                    const uint32_t cc = ufl(t[sym]), dd = ufl(t[nx]);
                    if (lane == 0) { t[sym] = (uint8_t)dd; t[nx] = (uint8_t)cc; out[opos] = (uint8_t)cc; }
                    opos++;
                    b3 = b2; b2 = b1; b1 = cc;
                    if (lane == 0) mru[b3] = (mru[b3] << 16) | (b2 << 8 | b1);
                } else if (sym < 258) {                               // word MRU slot 0 / 1
                    if (opos + 2 > sb.encpos) { err = (uint32_t)(-ZLNG_DEC_E_LZ); break; }
                    const uint32_t m = ufl(mru[c1]);
                    const uint32_t w = sym == 256 ? (m & 0xFFFF) : (m >> 16);
                    if (lane == 0) {
                        out[opos] = (uint8_t)(w >> 8); out[opos + 1] = (uint8_t)w;
                        if (sym == 257) mru[c1] = (m << 16) | w;
                    }
                    opos += 2;
                    b3 = b1; b2 = w >> 8; b1 = w & 0xFF;
                } else {                                              // match
                    const uint32_t mlen = sym - 258 + kMatchMin;
                    const uint32_t src = ufl(r[(head - (v >> 16)) & (kRing - 1)]);
                    if (opos + mlen > sb.encpos || src >= opos) { err = (uint32_t)(-ZLNG_DEC_E_LZ); break; }
                    const uint32_t dist = opos - src;
                    // cooperative forward copy (src/libzling_lz.cpp:91-104 semantics: byte j comes from src + j, which for
                    // an overlapping match is the period-`dist` pattern); every lane keeps the last byte it moved so the
                    // new context bytes come from registers instead of re-reading bytes that were just stored
                    uint32_t lastv = 0;
                    for (uint32_t j = lane; j < mlen; j += 64) {
                        const uint32_t sj = (dist >= 64 || dist >= mlen) ? j : j % dist;
                        lastv = out[src + sj];
                        out[opos + j] = (uint8_t)lastv;
                    }
                    opos += mlen;
                    b1 = (uint32_t)__builtin_amdgcn_readlane((int)lastv, (int)((mlen - 1) & 63));
                    b2 = (uint32_t)__builtin_amdgcn_readlane((int)lastv, (int)((mlen - 2) & 63));
                    b3 = (uint32_t)__builtin_amdgcn_readlane((int)lastv, (int)((mlen - 3) & 63));
                    const uint32_t w = b2 << 8 | b1;
                    const uint32_t m = ufl(mru[b3]);
                    if (lane == 0 && (m & 0xFFFF) != w) mru[b3] = (m << 16) | w;
                }
            }
            if (!err && opos != sb.encpos) err = (uint32_t)(-ZLNG_DEC_E_LZ);   // src/libzling_lz.cpp:371-373
            __syncthreads();
        }
        if (err && lane == 0) { a.summary[5] = b; a.summary[6] = err; }
    }
    __syncthreads();
    const uint8_t* keep = err ? a.mtf_snap : mtf;                    // a failed block leaves the tables as it found them
    __threadfence_block();
    for (uint32_t i = lane; i < 256 * 256 / 4; i += 64) reinterpret_cast<uint32_t*>(a.mtf_state)[i] = reinterpret_cast<const uint32_t*>(keep)[i];
}


