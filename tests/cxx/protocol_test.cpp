// protocol_test.cpp -- drives baidu::zling::Encode / Decode (include/libzling/libzling.h) the way a `-lzling` user does: with
// caller-written Inputter / Outputter classes and an ActionHandler.  What the reference's block driver makes observable
// (src/libzling.cpp:174-291, 293-427) is checked here or recorded for tests/test_gpu_protocol.py:
//   * GetData may deliver fewer bytes than asked for, PutData may accept fewer than offered (1..70,000 at random);
//   * OnInit comes first, then for block k: ALL of its bytes are pushed (none of block k + 1), then OnProcess(k), in stream
//     order, on the caller's thread; OnDone last, also when a stream reports an error -- then the call returns -1;
//   * a handler may WRITE to the outputter (Encode) and READ from the inputter (Decode) inside OnProcess: the Adler32 pair of
//     the reference's demo (demo/zling.cpp:60-70, 124-132, compiled out there by ENABLE_ADLER32_CHECKSUM 0).
// Usage: protocol_test <level> <input file> <out prefix>   -> writes <prefix>.zlng, <prefix>.dec, <prefix>.adler.zlng and a JSON log
// on stdout.  Test tool (tests/), built by libzling_amd/build.py; independent of the reference's sources.
#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <stdexcept>
#include <string>
#include <thread>
#include <vector>

#include "libzling.h"

namespace {

struct Rng {
    uint64_t s;
    uint32_t next() { s = s * 6364136223846793005ull + 1442695040888963407ull; return (uint32_t)(s >> 33); }
    size_t chunk() { return 1 + next() % 70000; }
};

struct MemInputter : baidu::zling::Inputter {
    const std::vector<unsigned char>& src;
    size_t pos = 0, err_at;
    bool err = false;
    Rng rng;
    MemInputter(const std::vector<unsigned char>& v, uint64_t seed, size_t fail_after = SIZE_MAX) : src(v), err_at(fail_after), rng{seed} {}
    size_t GetData(unsigned char* buf, size_t len) override {
        if (pos >= err_at) { err = true; return 0; }
        size_t n = std::min(std::min(len, rng.chunk()), src.size() - pos);
        n = std::min(n, err_at - pos);
        memcpy(buf, src.data() + pos, n);
        pos += n;
        return n;
    }
    bool IsEnd() override { return err || pos >= src.size(); }      // (an inputter in error has nothing more to give)
    bool IsErr() override { return err; }
};

struct MemOutputter : baidu::zling::Outputter {
    std::vector<unsigned char> dst;
    size_t err_at;
    bool err = false;
    Rng rng;
    explicit MemOutputter(uint64_t seed, size_t fail_after = SIZE_MAX) : err_at(fail_after), rng{seed} {}
    size_t PutData(unsigned char* buf, size_t len) override {
        if (dst.size() >= err_at) { err = true; return 0; }
        size_t n = std::min(std::min(len, rng.chunk()), err_at - dst.size());
        dst.insert(dst.end(), buf, buf + n);
        return n;
    }
    bool IsErr() override { return err; }
};

uint32_t adler32(const unsigned char* p, size_t n) {
    uint32_t a = 1, b = 0;
    for (size_t i = 0; i < n; i++) { a = (a + p[i]) % 65521; b = (b + a) % 65521; }
    return b << 16 | a;
}

struct Event { int kind; size_t size, in_pos, out_pos; uint32_t adler; bool own_thread; };   // kind 0 OnInit, 1 OnProcess, 2 OnDone

// records every callback; mode 1 writes the block's Adler32 behind it (Encode), mode 2 reads and checks it (Decode)
struct Recorder : baidu::zling::ActionHandler {
    std::vector<Event> ev;
    MemInputter* in = nullptr;
    MemOutputter* out = nullptr;
    int mode;
    std::thread::id caller = std::this_thread::get_id();
    explicit Recorder(int m = 0) : mode(m) {}
    void note(int kind, size_t size, uint32_t ad) { ev.push_back(Event{kind, size, in->pos, out->dst.size(), ad, std::this_thread::get_id() == caller}); }
    void OnInit() override {
        in = static_cast<MemInputter*>(GetInputter());
        out = static_cast<MemOutputter*>(GetOutputter());
        note(0, 0, 0);
    }
    void OnProcess(unsigned char* data, size_t size) override {
        const uint32_t ad = adler32(data, size);
        note(1, size, ad);
        if (mode == 1 && IsEncode()) GetOutputter()->PutUInt32(ad);
        if (mode == 2 && !IsEncode() && GetInputter()->GetUInt32() != ad) throw std::runtime_error("adler32 checksum not match.");
    }
    void OnDone() override { note(2, 0, 0); }
};

void dump(const std::string& path, const std::vector<unsigned char>& v) {
    FILE* f = fopen(path.c_str(), "wb");
    if (!f || fwrite(v.data(), 1, v.size(), f) != v.size()) { perror(path.c_str()); exit(2); }
    fclose(f);
}
void print_events(const char* name, const Recorder& r, int rc, const char* what) {
    printf("  \"%s\": {\"rc\": %d, \"threw\": \"%s\", \"events\": [", name, rc, what);
    for (size_t i = 0; i < r.ev.size(); i++)
        printf("%s[%d, %zu, %zu, %zu, %u, %d]", i ? ", " : "", r.ev[i].kind, r.ev[i].size, r.ev[i].in_pos, r.ev[i].out_pos, r.ev[i].adler, (int)r.ev[i].own_thread);
    printf("]}");
}

}  // namespace

int main(int argc, char** argv) {
    if (argc < 4) { fprintf(stderr, "usage: %s <level> <input> <out prefix>\n", argv[0]); return 2; }
    const int level = atoi(argv[1]);
    const std::string prefix = argv[3];
    std::vector<unsigned char> x;
    {
        FILE* f = fopen(argv[2], "rb");
        if (!f) { perror(argv[2]); return 2; }
        fseek(f, 0, SEEK_END); x.resize((size_t)ftell(f)); fseek(f, 0, SEEK_SET);
        if (fread(x.data(), 1, x.size(), f) != x.size()) return 2;
        fclose(f);
    }
    printf("{\n");
    // 1. Encode, short reads and writes, recording handler
    std::vector<unsigned char> z;
    {
        MemInputter in(x, 1); MemOutputter out(2); Recorder rec;
        const int rc = baidu::zling::Encode(&in, &out, &rec, level);
        print_events("encode", rec, rc, ""); printf(",\n");
        z.swap(out.dst);
        dump(prefix + ".zlng", z);
    }
    // 2. Decode of that stream, short reads and writes, recording handler (exact pull order: nothing read ahead)
    {
        MemInputter in(z, 3); MemOutputter out(4); Recorder rec;
        const int rc = baidu::zling::Decode(&in, &out, &rec);
        print_events("decode", rec, rc, ""); printf(",\n");
        dump(prefix + ".dec", out.dst);
    }
    // 3. the Adler32 pair: the handler writes behind every block while encoding, reads and checks while decoding
    std::vector<unsigned char> za;
    {
        MemInputter in(x, 5); MemOutputter out(6); Recorder rec(1);
        const int rc = baidu::zling::Encode(&in, &out, &rec, level);
        print_events("encode_adler", rec, rc, ""); printf(",\n");
        za.swap(out.dst);
        dump(prefix + ".adler.zlng", za);
    }
    {
        MemInputter in(za, 7); MemOutputter out(8); Recorder rec(2);
        std::string what;
        int rc = 99;
        try { rc = baidu::zling::Decode(&in, &out, &rec); } catch (const std::exception& e) { what = e.what(); }
        print_events("decode_adler", rec, rc, what.c_str()); printf(",\n");
        dump(prefix + ".adler.dec", out.dst);
    }
    {   // a damaged checksum (the last four bytes of the stream) must surface as the handler's exception
        std::vector<unsigned char> bad = za;
        if (bad.size() >= 1) bad[bad.size() - 1] ^= 0x5A;
        MemInputter in(bad, 9); MemOutputter out(10); Recorder rec(2);
        std::string what;
        int rc = 99;
        try { rc = baidu::zling::Decode(&in, &out, &rec); } catch (const std::exception& e) { what = e.what(); }
        print_events("decode_adler_damaged", rec, rc, what.c_str()); printf(",\n");
    }
    // 4. a stream that turns bad in the middle: -1, OnDone still fires (src/libzling.cpp:165-169, 286-290, 421-426)
    {
        MemInputter in(x, 11); MemOutputter out(12, z.size() / 2); Recorder rec;
        const int rc = baidu::zling::Encode(&in, &out, &rec, level);
        print_events("encode_output_error", rec, rc, ""); printf(",\n");
    }
    {
        MemInputter in(x, 13, x.size() / 2); MemOutputter out(14); Recorder rec;
        const int rc = baidu::zling::Encode(&in, &out, &rec, level);
        print_events("encode_input_error", rec, rc, ""); printf(",\n");
    }
    {
        MemInputter in(z, 15); MemOutputter out(16, x.size() / 2); Recorder rec;
        std::string what;
        int rc = 99;
        try { rc = baidu::zling::Decode(&in, &out, &rec); } catch (const std::exception& e) { what = e.what(); }
        print_events("decode_output_error", rec, rc, what.c_str()); printf(",\n");
    }
    {
        MemInputter in(z, 17, z.size() / 2); MemOutputter out(18); Recorder rec;
        std::string what;
        int rc = 99;
        try { rc = baidu::zling::Decode(&in, &out, &rec); } catch (const std::exception& e) { what = e.what(); }
        print_events("decode_input_error", rec, rc, what.c_str()); printf(",\n");
    }
    // 5. no handler at all (the batched Decode path) with short reads and writes
    {
        MemInputter in(z, 19); MemOutputter out(20);
        const int rc = baidu::zling::Decode(&in, &out, nullptr);
        printf("  \"decode_no_handler\": {\"rc\": %d, \"same_as_input\": %s}\n", rc, (out.dst == x) ? "true" : "false");
    }
    // 6. per-call traits of the handler (libzling.h): an Encode spread over two contexts of device 0 with the four longest rank
    //    chains on host threads produces the same bytes; a Decode whose handler is a DecodeReadAhead takes the batched path
    //    (several blocks per call: the inputter stands far behind block 0 at the first OnProcess) and the same output
    {
        struct Placed : baidu::zling::ActionHandler, baidu::zling::EncodePlacement {};
        static const int devs[2] = {0, 0};
        Placed h; h.devices = devs; h.ndevices = 2; h.host_rank_contexts = 4;
        MemInputter in(x, 21); MemOutputter out(22);
        std::string threw;
        int rc = -2;
        try { rc = baidu::zling::Encode(&in, &out, &h, level); } catch (const std::exception& e) { threw = e.what(); }
        printf("  ,\"encode_placement\": {\"rc\": %d, \"threw\": \"%s\", \"same_bytes\": %s}\n", rc, threw.c_str(), (out.dst == z) ? "true" : "false");
    }
    {
        struct Ahead : baidu::zling::ActionHandler, baidu::zling::DecodeReadAhead {
            MemInputter* in; size_t first_pos = 0; int calls = 0;
            void OnProcess(unsigned char*, size_t) override { if (calls++ == 0) first_pos = in->pos; }
        };
        MemInputter in(z, 23); MemOutputter out(24);
        Ahead h; h.in = &in;
        std::string threw;
        int rc = -2;
        try { rc = baidu::zling::Decode(&in, &out, &h); } catch (const std::exception& e) { threw = e.what(); }
        printf("  ,\"decode_read_ahead_trait\": {\"rc\": %d, \"threw\": \"%s\", \"same_as_input\": %s, \"calls\": %d, \"inputter_pos_at_first_onprocess\": %zu}\n",
               rc, threw.c_str(), (out.dst == x) ? "true" : "false", h.calls, h.first_pos);
    }
    printf("}\n");
    return 0;
}
