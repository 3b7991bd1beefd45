// libzling_shim.cpp -- baidu::zling::{Encode, Decode} and the stdio helpers, implemented on the
// C-ABI of include/zlng.h (HIP kernels in libzlng_hip.so).
//
// What is kept from the reference's block driver (src/libzling.cpp:174-291, 293-427) is its
// OBSERVABLE protocol: OnInit before any I/O; blocks are read with a GetData loop until 16 MiB or
// end of input; every block's bytes are pushed (PutData loop, short writes allowed) before that
// block's OnProcess(raw block, size); OnDone on every exit path; -1 iff a stream reports an error.
// What changes is the schedule: a batch of blocks is handed to the GPU at once so many 16 MiB
// blocks are parsed concurrently; per_block_out_end lets the bytes and callbacks still be emitted
// block by block in stream order.
#include <algorithm>
#include <cstring>
#include <future>
#include <memory>
#include <string>
#include <vector>

#include "libzling.h"
#include "zlng.h"

namespace baidu {
namespace zling {

// ---- Inputter / Outputter helpers (src/libzling_utils.cpp:40-65) ---------------------------
int Inputter::GetChar() {
    unsigned char b = 0;
    GetData(&b, 1);
    return b;
}
uint32_t Inputter::GetUInt32() {
    uint32_t v = 0;
    for (int i = 0; i < 4; i++) v = v << 8 | (uint32_t)GetChar();
    return v;
}
int Outputter::PutChar(int v) {
    unsigned char b = (unsigned char)v;
    PutData(&b, 1);
    return b;
}
uint32_t Outputter::PutUInt32(uint32_t v) {
    for (int shift = 24; shift >= 0; shift -= 8) PutChar((int)(v >> shift & 0xFF));
    return v;
}

// ---- stdio implementations (src/libzling_utils.cpp:67-92) -----------------------------------
size_t FileInputter::GetData(unsigned char* buf, size_t len) {
    size_t n = fread(buf, 1, len, fp_);
    consumed_ += n;
    return n;
}
bool FileInputter::IsEnd() {
    int c = fgetc(fp_);
    if (c == EOF) return true;
    ungetc(c, fp_);
    return false;
}
bool FileInputter::IsErr() { return ferror(fp_) != 0; }
size_t FileInputter::GetInputSize() { return consumed_; }

size_t FileOutputter::PutData(unsigned char* buf, size_t len) {
    size_t n = fwrite(buf, 1, len, fp_);
    produced_ += n;
    return n;
}
bool FileOutputter::IsErr() { return ferror(fp_) != 0; }
size_t FileOutputter::GetOutputSize() { return produced_; }

namespace {

const size_t kBlock = ZLNG_BLOCK_SIZE;

// Blocks handed to the GPU per call.  More blocks in flight = more parallel parse chains; the
// batch is bounded so memory stays bounded for unbounded streams.  ZLNG_BATCH_BLOCKS overrides.
int batch_blocks() {
    const char* e = getenv("ZLNG_BATCH_BLOCKS");
    int n = e ? atoi(e) : 64;
    return n < 1 ? 1 : (n > 240 ? 240 : n);
}
int pick_device() {
    const char* e = getenv("ZLNG_DEVICE");
    return e ? atoi(e) : 0;
}
// Devices an Encode() call spreads one stream over: ZLNG_DEVICES="0,1,2,3" (an index may repeat: several contexts on
// one GPU), else the single device of ZLNG_DEVICE, else device 0.
std::vector<int> pick_devices() {
    std::vector<int> d;
    if (const char* e = getenv("ZLNG_DEVICES")) {
        for (const char* p = e; *p;) {
            char* end = nullptr;
            long v = strtol(p, &end, 10);
            if (end == p) break;
            d.push_back((int)v);
            p = *end == ',' ? end + 1 : end;
        }
    }
    if (d.empty()) d.push_back(pick_device());
    return d;
}

struct CtxGuard {
    zlng_ctx* c;
    explicit CtxGuard(zlng_ctx* p) : c(p) {}
    ~CtxGuard() { zlng_destroy(c); }
};

zlng_ctx* make_ctx(int level, bool encode, int blocks) {
    int err = 0;
    zlng_ctx* c = zlng_create(pick_device(), level, encode ? 1 : 0, blocks, &err);
    if (!c) {
        if (err == ZLNG_E_NOMEM) throw std::bad_alloc();
        throw std::runtime_error(std::string("zling: no gfx950 device (") + zlng_strerror(err) + ")");
    }
    return c;
}

bool push_all(Outputter* out, unsigned char* p, size_t n) {       // src/libzling.cpp:273-276
    for (size_t off = 0; off < n && !out->IsErr();) off += out->PutData(p + off, n - off);
    return !out->IsErr();
}

// One batch of blocks on its way through the GPU.
// Uninitialised byte buffer: a std::vector would zero-fill gigabytes that a short stream never touches.
struct RawBuf {
    std::unique_ptr<unsigned char[]> p;
    size_t n = 0;
    void resize(size_t bytes) { p.reset(new unsigned char[bytes]); n = bytes; }
    unsigned char* data() { return p.get(); }
    size_t size() const { return n; }
};

// Growable byte buffer that does not zero-fill what it hands out (std::vector::resize would memset every payload before the
// read overwrites it) and keeps its storage across clear().
struct ByteBuf {
    std::unique_ptr<unsigned char[]> p;
    size_t n = 0, cap = 0;
    void clear() { n = 0; }
    size_t size() const { return n; }
    unsigned char* data() { return p.get(); }
    unsigned char* grow(size_t k) {                      // k more bytes at the end; returns where they start
        if (n + k > cap) {
            size_t c = std::max(n + k, std::max<size_t>(cap * 2, 1u << 20));
            std::unique_ptr<unsigned char[]> q(new unsigned char[c]);
            if (n) memcpy(q.get(), p.get(), n);
            p.swap(q);
            cap = c;
        }
        unsigned char* at = p.get() + n;
        n += k;
        return at;
    }
    void push(unsigned char b) { *grow(1) = b; }
    void append(const unsigned char* src, size_t k) { memcpy(grow(k), src, k); }
};

struct EncodeSlot {
    zlng_group* grp = nullptr;    // one context per device of ZLNG_DEVICES; a batch is split into contiguous block ranges
    RawBuf in, out;
    std::vector<size_t> ilen, ends;
    int have = 0;                 // blocks read into `in`
    size_t total = 0;             // their bytes
    ~EncodeSlot() { if (grp) zlng_group_destroy(grp); }
};

zlng_group* make_group(const std::vector<int>& devices, int level, int blocks_per_member) {
    int err = 0;
    zlng_group* g = zlng_group_create(devices.data(), (int)devices.size(), level, blocks_per_member, &err);
    if (!g) {
        if (err == ZLNG_E_NOMEM) throw std::bad_alloc();
        throw std::runtime_error(std::string("zling: no gfx950 device (") + zlng_strerror(err) + ")");
    }
    return g;
}

void throw_rc(int rc) {
    if (rc == ZLNG_E_NOMEM) throw std::bad_alloc();
    if (rc != ZLNG_OK) throw std::runtime_error(std::string("baidu::zling::Encode(): ") + zlng_strerror(rc));
}

// fill up to nb blocks exactly as the reference fills one (src/libzling.cpp:193-196); false on an input error
bool read_batch(Inputter* inputter, EncodeSlot& s, int nb) {
    s.have = 0;
    s.total = 0;
    while (s.have < nb && !inputter->IsEnd() && !inputter->IsErr()) {
        size_t got = 0;
        unsigned char* dst = s.in.data() + (size_t)s.have * kBlock;
        while (!inputter->IsEnd() && !inputter->IsErr() && got < kBlock) {
            got += inputter->GetData(dst + got, kBlock - got);
            if (inputter->IsErr()) break;
        }
        if (inputter->IsErr()) return false;
        s.ilen[(size_t)s.have] = got;
        s.total += got;
        s.have++;
        if (got < kBlock) break;              // a short block can only be the stream's last
    }
    return true;
}

}  // namespace

// Schedule: two contexts take the batches alternately.  For batch k the caller's thread reads the
// blocks and queues the parse (zlng_encode_parse returns once the copy is staged), then a helper thread runs the
// blocking GPU half -- import the 64 KiB stream state left by batch k-1, rank + Huffman + copy back, export the
// state -- while the caller's thread writes batch k-1 and reads batch k+1.  On the GPU the parse of batch k+1 (own
// stream, does not depend on the MTF state) runs beside batch k's rank stage.  Bytes and callbacks leave block by
// block, in stream order, on the caller's thread only.  ZLNG_PIPELINE=0 keeps everything on one context and thread.
int Encode(Inputter* inputter, Outputter* outputter, ActionHandler* handler, int level) {
    if (handler) {
        handler->SetInputterOutputter(inputter, outputter, true);
        handler->OnInit();
    }
    bool bad_level = level < 0 || level > 4;     // the reference would emit a corrupt stream
    if (!bad_level) {
        // where and how this call runs: a per-call trait of the handler (EncodePlacement, libzling.h) in front of the environment
        const EncodePlacement* place = handler ? dynamic_cast<EncodePlacement*>(handler) : nullptr;
        const std::vector<int> devices = (place && place->devices && place->ndevices > 0)
                                             ? std::vector<int>(place->devices, place->devices + place->ndevices) : pick_devices();
        const int host_chains = place ? place->host_rank_contexts : -1;
        const int ndev = (int)devices.size();
        const int per_member = std::max(1, batch_blocks());
        const int nb = per_member * ndev;                 // blocks per batch: every device gets `per_member` of them
        const char* pe = getenv("ZLNG_PIPELINE");
        const bool pipelined = !(pe && atoi(pe) == 0);
        const int nslots = pipelined ? 2 : 1;
        EncodeSlot slot[2];
        // Host buffers first (their pages are only touched as far as the stream goes); the context -- whose device
        // pools cost ~10 ms per block to create -- after the first batch is in, sized to it when the stream is short.
        auto prepare_host = [&](EncodeSlot& s) {
            if (s.in.size()) return;
            s.in.resize((size_t)nb * kBlock);
            s.out.resize(zlng_encode_bound((size_t)nb * kBlock));
            s.ilen.resize((size_t)nb);
            s.ends.resize((size_t)nb);
        };
        auto prepare_ctx = [&](EncodeSlot& s, bool stream_ended) {
            if (!s.grp) {
                s.grp = make_group(devices, level, stream_ended ? std::max((s.have + ndev - 1) / ndev, 1) : per_member);
                if (host_chains >= 0) throw_rc(zlng_group_set_host_rank_contexts(s.grp, host_chains));
            }
        };
        std::vector<unsigned char> state(ZLNG_MTF_STATE);
        int state_level = level;
        bool have_state = false, failed = false;
        // GPU half of a batch (any thread; touches only its slot and the state hand-off, never the streams)
        auto gpu_finish = [&](EncodeSlot* s) {
            if (have_state) throw_rc(zlng_group_set_state(s->grp, state.data(), state_level));
            size_t produced = 0;
            throw_rc(zlng_group_encode_finish(s->grp, s->out.data(), s->out.size(), &produced, s->ends.data()));
            throw_rc(zlng_group_get_state(s->grp, state.data(), &state_level));
            have_state = true;
        };
        // bytes, then the callback, block by block (src/libzling.cpp:273-283); caller's thread
        auto emit = [&](EncodeSlot& s) {
            size_t prev = 0;
            for (int b = 0; b < s.have; b++) {
                if (!push_all(outputter, s.out.data() + prev, s.ends[(size_t)b] - prev)) { failed = true; return; }
                prev = s.ends[(size_t)b];
                if (handler) handler->OnProcess(s.in.data() + (size_t)b * kBlock, s.ilen[(size_t)b]);
            }
        };
        std::future<void> pending;               // GPU half of the batch in slot[(k-1) % 2]
        EncodeSlot* pending_slot = nullptr;
        struct Joiner {                          // never leave the scope with the helper still using the slots
            std::future<void>& f;
            ~Joiner() { if (f.valid()) f.wait(); }
        } joiner{pending};
        int k = 0;
        while (!failed && !inputter->IsEnd() && !inputter->IsErr()) {
            EncodeSlot& cur = slot[k % nslots];
            prepare_host(cur);
            if (!read_batch(inputter, cur, nb)) { failed = true; break; }
            if (cur.have == 0) break;
            prepare_ctx(cur, cur.have < nb || inputter->IsEnd());
            throw_rc(zlng_group_encode_parse(cur.grp, cur.in.data(), cur.total));
            EncodeSlot* done = pending_slot;
            if (pending.valid()) pending.get();                      // batch k-1 is back (rethrows its error)
            if (pipelined) {
                pending = std::async(std::launch::async, gpu_finish, &cur);
                pending_slot = &cur;
            } else {
                gpu_finish(&cur);
                done = &cur;
            }
            if (done) emit(*done);
            k++;
        }
        if (pending.valid()) {
            pending.get();
            if (!failed && !inputter->IsErr()) emit(*pending_slot);
        }
    }
    if (handler) handler->OnDone();
    return (bad_level || inputter->IsErr() || outputter->IsErr()) ? -1 : 0;
}

namespace {

// Blocks in a run of .zlng bytes, by hopping the sub-block headers (flag, encpos, rlen, olen: src/libzling.cpp:312-324); a tail
// that is not a whole block counts as one.  Sizing only: the decoder itself validates every field.
size_t blocks_left(const unsigned char* z, size_t n) {
    size_t blocks = 0, p = 0;
    bool open = false;
    while (p < n) {
        if (z[p] == 0) { blocks++; p++; open = false; continue; }
        if (z[p] != 1 || n - p < 13) { open = true; break; }
        const size_t olen = (size_t)z[p + 9] << 24 | (size_t)z[p + 10] << 16 | (size_t)z[p + 11] << 8 | z[p + 12];
        open = true;
        if (olen > n - p - 13) break;
        p += 13 + olen;
    }
    return blocks + (open ? 1 : 0);
}

// Decode with an ActionHandler installed: the reference pulls exactly the bytes a block consists of before it calls
// OnProcess (src/libzling.cpp:306-336: GetChar, three GetUInt32, GetData(olen) per sub-block, the 0x00 that closes the block), and a
// handler may itself read from the inputter inside OnProcess -- the Adler32 variant of the demo does (demo/zling.cpp:124-132) --
// so nothing may be read ahead.  One block per GPU call; the context carries the literal tables from block to block.
// A bad flag or bad sizes stop the READING of the block, not the decoding of what was read: the reference meets errors in stream
// order (a sub-block's Huffman stream and replay come before the next sub-block's flag), so the bytes up to and including the
// offending flag / header go to zlng_decode_blocks, which reports the first error in that order (csrc/decode.hip, k_frame_walk).
// Returns false on an I/O error (the caller still fires OnDone and returns -1, src/libzling.cpp:421-426).
bool decode_block_by_block(Inputter* inputter, Outputter* outputter, ActionHandler* handler) {
    CtxGuard ctx(make_ctx(0, false, 1));
    ByteBuf z;                                                            // one block's compressed bytes; its storage is reused from block to block
    RawBuf raw;
    raw.resize(kBlock);
    while (!inputter->IsEnd()) {                                          // src/libzling.cpp:306
        z.clear();
        bool closed = false, bad = false;
        int bad_code = ZLNG_E_FLAG;
        while (!inputter->IsEnd()) {                                      // :312
            const int flag = inputter->GetChar();
            z.push((unsigned char)flag);
            if (flag != 0 && flag != 1) { bad = true; bad_code = ZLNG_E_FLAG; break; }   // :315-317 ("invalid encflag." unless something in front of it fails first)
            if (flag == 0) { closed = true; break; }                       // :318-320
            uint32_t hdr[3];                                               // encpos, rlen, olen (:322-324)
            unsigned char hb[12];
            for (int k = 0; k < 3; k++) {
                hdr[k] = inputter->GetUInt32();
                if (inputter->IsErr()) return false;
                for (int j = 0; j < 4; j++) hb[4 * k + j] = (unsigned char)(hdr[k] >> (24 - 8 * j) & 0xFF);
            }
            z.append(hb, sizeof hb);
            if (hdr[1] > 262144u || hdr[2] > 393216u) { bad = true; bad_code = ZLNG_E_BLOCKSIZE; break; }   // :326-328 ("invalid block size.", likewise)
            const size_t olen = hdr[2];
            unsigned char* dst = z.grow(olen);                              // uninitialised: every byte of it is read into below or the call throws
            size_t got = 0;
            while (!inputter->IsEnd() && got < olen) {                     // :329-332
                got += inputter->GetData(dst + got, olen - got);
                if (inputter->IsErr()) return false;
            }
            if (got < olen) throw std::runtime_error(zlng_strerror(ZLNG_E_TRUNC));   // the reference decodes what its buffer held before
        }
        if (z.size() == 0) break;
        if (!closed && !bad) z.push(0);       // end of input inside a block: the reference's inner loop ends there too (:312) and writes the block
        size_t used = 0, produced = 0, end = 0;
        if (z.size() > 1 || bad) {
            const int rc = zlng_decode_blocks(ctx.c, z.data(), z.size(), &used, raw.data(), raw.size(), &produced, &end);
            if (rc == ZLNG_E_NOMEM) throw std::bad_alloc();
            if (rc != ZLNG_OK) throw std::runtime_error(zlng_strerror(rc));
            // The buffer ended in a flag / header that is not one: whatever the library made of the bytes in front of it (it reports
            // the first error in stream order, which is normally this very one), the stream does not go on behind it.
            if (bad) throw std::runtime_error(zlng_strerror(bad_code));
        }
        if (!push_all(outputter, raw.data(), produced)) return false;      // :412-415
        handler->OnProcess(raw.data(), produced);                          // :417-419
    }
    return true;
}

}  // namespace

int Decode(Inputter* inputter, Outputter* outputter, ActionHandler* handler) {
    if (handler) {
        handler->SetInputterOutputter(inputter, outputter, false);
        handler->OnInit();
    }
    // Per call: a handler that is also a baidu::zling::DecodeReadAhead (libzling.h) promises never to touch the inputter inside
    // OnProcess and takes the batched path; ZLNG_DECODE_READAHEAD=0|1, read at every call, overrides the handler's trait either way.
    const char* ra = getenv("ZLNG_DECODE_READAHEAD");
    const bool readahead = ra ? atoi(ra) != 0 : (handler && dynamic_cast<DecodeReadAhead*>(handler) != nullptr);
    if (handler && !readahead) {
        // exact pull order (see decode_block_by_block).  (false = an I/O error: reported through IsErr() in the return below, like the reference.)
        (void)decode_block_by_block(inputter, outputter, handler);
    } else {
        const int nb_full = std::min(batch_blocks(), 64);
        // Small streams are the common case for a library call: start with a 4-block context (the decode pools cost ~80 MB
        // of HBM per block) and an uninitialised buffer, and move to the full batch once the stream has filled the small one.
        int nb = std::min(nb_full, 4);
        CtxGuard ctx(make_ctx(0, false, nb));
        // A .zlng stream has no index: the compressed bytes are pulled in chunks, whole blocks found in
        // the accumulated prefix are decoded, the unconsumed tail is kept for the next round.
        std::vector<unsigned char> z;
        RawBuf raw;
        raw.resize((size_t)nb * kBlock);
        std::vector<size_t> ends((size_t)nb);
        const size_t chunk = 8u << 20;
        size_t zoff = 0;                                  // consumed prefix of z (compacted now and then, not per round)
        size_t total_out = 0;                             // decoded bytes so far
        bool failed = false, eof = false, closed_tail = false;
        while (!failed) {
            if (!eof) {
                if (zoff > (z.size() >> 1)) { z.erase(z.begin(), z.begin() + (long)zoff); zoff = 0; }
                size_t old = z.size();
                z.resize(old + chunk);
                size_t got = 0;
                while (got < chunk && !inputter->IsEnd() && !inputter->IsErr()) got += inputter->GetData(z.data() + old + got, chunk - got);
                z.resize(old + got);
                if (inputter->IsErr()) { failed = true; break; }
                eof = inputter->IsEnd();
            }
            if (z.size() == zoff) break;
            size_t used = 0, produced = 0;
            std::fill(ends.begin(), ends.end(), (size_t)0);
            int rc = zlng_decode_blocks(ctx.c, z.data() + zoff, z.size() - zoff, &used, raw.data(), raw.size(), &produced, ends.data());
            if (rc == ZLNG_E_TRUNC && eof && !closed_tail) {
                // The reference's inner loop also ends at end of input (src/libzling.cpp:312-313), so a final block without
                // its 0x00 terminator is still decoded and written: close it and look again.
                z.push_back(0);
                closed_tail = true;
                continue;
            }
            if (rc == ZLNG_E_NOMEM) throw std::bad_alloc();
            if (rc == ZLNG_E_TRUNC && !eof) { continue; }            // need more bytes for the next block
            if (rc != ZLNG_OK) throw std::runtime_error(zlng_strerror(rc));
            // whole good blocks only (a corrupt block is met again, alone, at the head of the next call and throws there --
            // after every block before it went out, like the reference, src/libzling.cpp:412-419)
            size_t prev = 0;
            for (size_t b = 0; b < ends.size() && prev < produced; b++) {
                if (!push_all(outputter, raw.data() + prev, ends[b] - prev)) { failed = true; break; }
                if (handler) handler->OnProcess(raw.data() + prev, ends[b] - prev);
                prev = ends[b];
            }
            zoff += used;
            total_out += produced;
            // The stream has filled this context once over: take a larger one -- four times the blocks per step (4, 16, 64: a
            // context's pools cost ~10 ms per block to create, so a 6-block stream must not pay for 64), and when the input has
            // ended, no more than the blocks that are left (counted by hopping the headers of the unconsumed bytes).
            int nb_next = std::min(nb_full, nb * 4);
            if (eof) nb_next = std::min(nb_next, std::max(nb, (int)std::min<size_t>(blocks_left(z.data() + zoff, z.size() - zoff), (size_t)nb_full)));
            if (nb_next > nb && total_out >= (size_t)nb * kBlock) {
                std::vector<unsigned char> st(ZLNG_MTF_STATE);
                int lv = 0;                                               // (current_level: an encoder-side scalar, carried by the state call, unused here)
                if (zlng_get_state(ctx.c, st.data(), &lv) != ZLNG_OK) throw std::runtime_error(zlng_strerror(ZLNG_E_DEVICE));
                zlng_destroy(ctx.c);
                ctx.c = nullptr;
                nb = nb_next;
                ctx.c = make_ctx(0, false, nb);
                if (zlng_set_state(ctx.c, st.data(), 0) != ZLNG_OK) throw std::runtime_error(zlng_strerror(ZLNG_E_DEVICE));
                raw.resize((size_t)nb * kBlock);
                ends.assign((size_t)nb, 0);
            }
            if (used == 0 && eof) {
                if (z.size() != zoff) throw std::runtime_error(zlng_strerror(ZLNG_E_TRUNC));
                break;
            }
            if (z.size() == zoff && eof) break;
        }
    }
    if (handler) handler->OnDone();
    return (inputter->IsErr() || outputter->IsErr()) ? -1 : 0;
}

}  // namespace zling
}  // namespace baidu
