// zling_demo -- command-line front end over the drop-in API, same usage as the reference's
// demo/zling.cpp:159-235:
//
//     zling_demo e[0-4] [input [output]]     compress   (default level 0; stdin/stdout when omitted)
//     zling_demo d      [input [output]]     decompress
//
// Progress and throughput go to stderr.  Unlike the reference (clock(), CPU time,
// demo/zling.cpp:94-113) the rate is wall-clock: the work happens on the GPU.
#include <chrono>
#include <cstdio>
#include <cstring>
#include <stdexcept>

#include "libzling.h"

namespace {

// (DecodeReadAhead: OnProcess below only reads the two byte counters, never the stream, so Decode may read ahead)
struct Progress : baidu::zling::ActionHandler, baidu::zling::DecodeReadAhead {
    std::chrono::steady_clock::time_point t0;
    baidu::zling::FileInputter* in;
    baidu::zling::FileOutputter* out;
    void OnInit() override { t0 = std::chrono::steady_clock::now(); }
    void OnProcess(unsigned char*, size_t) override {
        fprintf(stderr, "\r%zu => %zu", in->GetInputSize(), out->GetOutputSize());
        fflush(stderr);
    }
    void OnDone() override {
        const double s = std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count();
        const size_t a = in->GetInputSize(), b = out->GetOutputSize();
        const size_t raw = IsEncode() ? a : b, packed = IsEncode() ? b : a;
        fprintf(stderr, "\r%s: %zu => %zu, ratio %.4f, %.3f s wall, %.2f MB/s\n", IsEncode() ? "encode" : "decode", a, b,
                raw ? (double)packed / (double)raw : 0.0, s, s > 0 ? (double)raw / s / 1e6 : 0.0);
    }
};

int usage(const char* argv0) {
    fprintf(stderr, "usage: %s e[0-4]|d [input-file [output-file]]\n", argv0);
    return 2;
}

}  // namespace

int main(int argc, char** argv) {
    if (argc < 2 || argc > 4) return usage(argv[0]);
    int level = -1;
    bool decode = false;
    if (strcmp(argv[1], "d") == 0) decode = true;
    else if (strcmp(argv[1], "e") == 0) level = 0;
    else if (argv[1][0] == 'e' && argv[1][1] >= '0' && argv[1][1] <= '4' && argv[1][2] == 0) level = argv[1][1] - '0';
    else return usage(argv[0]);

    FILE* fi = argc > 2 ? fopen(argv[2], "rb") : stdin;
    FILE* fo = argc > 3 ? fopen(argv[3], "wb") : stdout;
    if (!fi || !fo) { perror("zling_demo"); return 1; }
    baidu::zling::FileInputter in(fi);
    baidu::zling::FileOutputter out(fo);
    Progress p;
    p.in = &in;
    p.out = &out;
    int rc;
    try {
        rc = decode ? baidu::zling::Decode(&in, &out, &p) : baidu::zling::Encode(&in, &out, &p, level);
    } catch (const std::exception& e) {
        fprintf(stderr, "\nzling_demo: %s\n", e.what());
        return 1;
    }
    if (fo != stdout) fclose(fo); else fflush(fo);
    if (fi != stdin) fclose(fi);
    if (rc != 0) { fprintf(stderr, "zling_demo: I/O error\n"); return 1; }
    return 0;
}
