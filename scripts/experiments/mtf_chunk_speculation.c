// Experiment behind DESIGN.md 3/K2 (research tool, not product, not test): can one context's rank chain
// (ZlingMTFEncoder::Encode, src/libzling_lz.cpp:112-117) be cut into chunks that start from a GUESSED table and be stitched
// afterwards?  Counts the literals whose speculative rank differs from the true one.  Result on 64 MiB of text (hot context
// ' '): 66-78 % differ, the late half of a chunk as often as the early half -- the neighbour-swap rule never forgets.
//   python scripts/experiments/gen_literal_streams.py 4   # writes /tmp/exp/lit_32.bin etc. from the oracle's parse
//   gcc -O2 -o /tmp/exp/spec scripts/experiments/mtf_chunk_speculation.c && /tmp/exp/spec /tmp/exp/lit_32.bin 65536 65536 4
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <stdint.h>
static uint8_t nxt[256];
static const uint8_t mtfinit[256] = {
     32, 101, 116,  97, 105, 111, 110, 114, 115, 108, 104, 100,  99, 117,  93,  91,
    109, 112, 103, 102,  10, 121,  98,  39, 119,  46,  44, 118,  59,  38, 124,  47,
     49, 107,  61,  48,  67,  65,  58,  45,  84,  83,  60,  62,  50, 113,  73,  57,
     42, 120,  41,  40,  66,  77,  80,  69,  68,  53,  51,  72,  70,  56,  52,  71,
     82,  54,  76,  55,  78,  87, 122, 125, 123,  79, 106,  85,  74,  75, 208,  95,
    195,  35,  86, 215,  90,  34,  89, 209, 128, 224, 184, 131,  92, 227,  37,  33,
    176, 169, 206, 226, 130,  63,  88,  81, 161, 153,  43, 129, 188, 179, 216, 164,
    181, 189, 148, 190, 173, 187, 186, 229, 225, 167, 217, 177, 178, 168, 149, 185,
    197, 144, 147, 196, 207, 194, 180, 156, 132, 170, 166, 136, 182, 191,   9, 230,
    141, 160, 175,  36, 152, 140, 165, 145,  94, 133, 163, 183, 171, 157, 137, 174,
    134, 135, 236, 151, 231, 155, 201, 158, 138, 143, 150, 162, 159, 139, 172, 154,
    126, 232, 235, 146, 233, 228, 202, 203, 142, 214, 237, 204, 219, 234, 213,  96,
    218, 199,  64, 210, 239, 198, 211, 205, 212, 240, 222, 220, 200,   0,   1,   2,
      3,   4,   5,   6,   7,   8,  11,  12,  13,  14,  15,  16,  17,  18,  19,  20,
     21,  22,  23,  24,  25,  26,  27,  28,  29,  30,  31, 127, 192, 193, 221, 223,
    238, 241, 242, 243, 244, 245, 246, 247, 248, 249, 250, 251, 252, 253, 254, 255,
};
typedef struct { uint8_t tab[256], idx[256]; } Mtf;
static void mtf_init(Mtf* m, const uint8_t* t) { for (int i = 0; i < 256; i++) { m->tab[i] = t[i]; m->idx[t[i]] = i; } }
static inline int mtf_enc(Mtf* m, uint8_t c) {
    int i = m->idx[c], n = nxt[i];
    uint8_t d = m->tab[n];
    m->tab[n] = c; m->tab[i] = d; m->idx[c] = n; m->idx[d] = i;
    return i;
}
int main(int argc, char** argv) {
    for (int i = 0; i < 256; i++) nxt[i] = i < 128 ? i * 95 / 100 : i * 55 / 100;
    const char* f = argv[1]; long chunk = atol(argv[2]); long warm = atol(argv[3]); int mode = argc > 4 ? atoi(argv[4]) : 0;
    FILE* fp = fopen(f, "rb"); fseek(fp, 0, SEEK_END); long n = ftell(fp); fseek(fp, 0, SEEK_SET);
    uint8_t* L = malloc(n); fread(L, 1, n, fp); fclose(fp);
    uint8_t* rt = malloc(n);
    Mtf T; mtf_init(&T, mtfinit);
    long nch = (n + chunk - 1) / chunk;
    Mtf* Tstart = malloc(sizeof(Mtf) * (nch + 1));
    for (long j = 0; j < n; j++) { if (j % chunk == 0) Tstart[j / chunk] = T; rt[j] = mtf_enc(&T, L[j]); }
    long tot_ev = 0, tot = 0, maxev = 0; long late_ev = 0;
    // mode 0: guess = mtfinit + warm-up over the `warm` literals before the chunk
    // mode 1: guess = true state at start of chunk k-1 is NOT available; use spec end of previous chunk's spec run (round 2), iterated `mode` rounds
    Mtf* G = malloc(sizeof(Mtf) * (nch + 1));
    for (long k = 0; k < nch; k++) mtf_init(&G[k], mtfinit);
    int rounds = mode > 0 ? mode : 1;
    for (int r = 0; r < rounds; r++) {
        Mtf* E = malloc(sizeof(Mtf) * (nch + 1));
        tot_ev = 0; tot = 0; maxev = 0; late_ev = 0;
        for (long k = 1; k < nch; k++) {
            Mtf S = G[k];
            long b = k * chunk, e = b + chunk > n ? n : b + chunk;
            long w0 = b - warm < 0 ? 0 : b - warm;
            if (mode > 0 && r > 0) w0 = b;        // later rounds: start from previous chunk's speculative end state, no warm-up
            for (long j = w0; j < b; j++) mtf_enc(&S, L[j]);
            long ev = 0;
            for (long j = b; j < e; j++) { int rk = mtf_enc(&S, L[j]); if (rk != rt[j]) { ev++; if (j - b > (e - b) / 2) late_ev++; } }
            E[k] = S;
            tot_ev += ev; tot += e - b; if (ev > maxev) maxev = ev;
        }
        printf("round %d: chunks %ld, literals %ld, events %ld (%.4f%%), late-half events %ld, max per chunk %ld\n", r, nch - 1, tot, tot_ev, 100.0 * tot_ev / tot, late_ev, maxev);
        for (long k = nch - 1; k >= 2; k--) G[k] = E[k - 1];
        G[1] = Tstart[1];   // chunk 0 is exact from the start
        free(E);
    }
    return 0;
}
