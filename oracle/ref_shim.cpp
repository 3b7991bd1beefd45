// ref_shim.cpp -- extern "C" handles onto the REAL reference, compiled in place.
//
// TEST INFRASTRUCTURE ONLY.  oracle/Makefile compiles this file together with the
// reference's own sources where they lie under $(REF)/src (nothing is copied into
// the repository) into oracle/_ref/libzling_ref.so.  It is used to pin
// zlng_oracle.c, to generate tests/golden/, and optionally as bench.py's
// cpu_baseline ("kind": "reference").  It contains no codec logic of its own:
// memory-backed Inputter/Outputter subclasses and thin calls into
// baidu::zling::{Encode,Decode} (src/libzling.h:44-45) and, for stage probes,
// huffman::ZlingMakeLengthTable (src/libzling_huffman.h:49) and
// lz::ZlingRolzEncoder (src/libzling_lz.h:68-107).
#include <cstdint>
#include <cstring>
#include <stdexcept>

#include "libzling.h"
#include "libzling_huffman.h"
#include "libzling_lz.h"

namespace {

struct MemIn : baidu::zling::Inputter {
    const unsigned char* p; size_t n, pos;
    MemIn(const unsigned char* p_, size_t n_) : p(p_), n(n_), pos(0) {}
    size_t GetData(unsigned char* buf, size_t len) override {
        size_t k = len < n - pos ? len : n - pos;
        memcpy(buf, p + pos, k); pos += k; return k;
    }
    bool IsEnd() override { return pos >= n; }
    bool IsErr() override { return false; }
};
struct MemOut : baidu::zling::Outputter {
    unsigned char* p; size_t cap, pos; bool err;
    MemOut(unsigned char* p_, size_t cap_) : p(p_), cap(cap_), pos(0), err(false) {}
    size_t PutData(unsigned char* buf, size_t len) override {
        if (pos + len > cap) { err = true; return 0; }
        memcpy(p + pos, buf, len); pos += len; return len;
    }
    bool IsErr() override { return err; }
};

}  // namespace

extern "C" {

int ref_encode(const uint8_t* in, size_t n, int level, uint8_t* out, size_t cap, size_t* out_len) {
    MemIn i(in, n); MemOut o(out, cap);
    int rc = baidu::zling::Encode(&i, &o, NULL, level);
    *out_len = o.pos;
    return rc;
}

// returns 0 ok, -1 io error, -2 runtime_error (message copied to msg if non-null)
int ref_decode(const uint8_t* in, size_t n, uint8_t* out, size_t cap, size_t* out_len, char* msg, size_t msg_cap) {
    MemIn i(in, n); MemOut o(out, cap);
    int rc;
    try {
        rc = baidu::zling::Decode(&i, &o, NULL);
    } catch (const std::runtime_error& e) {
        if (msg && msg_cap) { strncpy(msg, e.what(), msg_cap - 1); msg[msg_cap - 1] = 0; }
        *out_len = o.pos;
        return -2;
    }
    *out_len = o.pos;
    return rc;
}

void ref_make_length_table(const uint32_t* freq, uint32_t* len, int n, int limit) {
    baidu::zling::huffman::ZlingMakeLengthTable(freq, len, n, limit);
}

void ref_make_encode_table(const uint32_t* len, uint16_t* code, int n, int limit) {
    baidu::zling::huffman::ZlingMakeEncodeTable(len, code, n, limit);
}

// Stage probe: run the reference ROLZ encoder over one block (<= 16 MiB, caller supplies
// >= 275 bytes of slack after ilen) and dump its u16 stream sub-block by sub-block.
// tok16 receives all sub-blocks back to back; cuts receives (encpos, rlen) pairs.
// A fresh encoder (fresh MTF) is used unless `enc` is passed from ref_rolz_new.
void* ref_rolz_new() { return new baidu::zling::lz::ZlingRolzEncoder(); }
void ref_rolz_free(void* e) { delete static_cast<baidu::zling::lz::ZlingRolzEncoder*>(e); }
int ref_rolz_block(void* e, int level, uint8_t* ibuf, int ilen, uint16_t* tok16, size_t tok_cap, int* cuts, int cuts_cap) {
    auto* enc = static_cast<baidu::zling::lz::ZlingRolzEncoder*>(e);
    enc->Reset();
    int encpos = 0, nsub = 0;
    size_t o = 0;
    while (encpos < ilen) {
        if (o + 262144 + 275 > tok_cap || nsub >= cuts_cap) return -1;
        int rlen = enc->Encode(level, ibuf, tok16 + o, ilen, 262144, &encpos);
        cuts[2 * nsub] = encpos; cuts[2 * nsub + 1] = rlen; nsub++;
        o += (size_t)rlen;
    }
    return nsub;
}

// Stage probe for bench.py's `rank_chain` line: the reference's own MTF encoder (src/libzling_lz.cpp:106-117) over the
// literal bytes of ONE context, from the initial table.  Returns a checksum of the ranks so the loop cannot be elided.
uint64_t ref_mtf_chain(const uint8_t* lits, size_t n, uint8_t* ranks) {
    baidu::zling::lz::ZlingMTFEncoder m;
    uint64_t sum = 0;
    for (size_t i = 0; i < n; i++) { const unsigned char r = m.Encode(lits[i]); if (ranks) ranks[i] = r; sum += r; }
    return sum;
}

}  // extern "C"
