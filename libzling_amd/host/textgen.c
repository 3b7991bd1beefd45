/*
 * textgen.c -- deterministic enwik-shaped synthetic text (SURVEY.md 8(d) "Synthetic input spec").
 *
 * enwik8/enwik9 are not present on the build or GPU boxes and there is no network, so the
 * benchmark and the large parity tests run on this generator unless a real file is supplied.
 *
 *   PRNG        SplitMix64, seed = 0x5A4C4E47 ("ZLNG") + chunk index
 *   vocabulary  65,536 words; length 1 + Poisson(4.5) capped at 14; letters drawn from English
 *               unigram frequencies (vocabulary is independent of the chunk index)
 *   words       Zipf(s = 1.1) by inverse-CDF table
 *   separators  89 % " ", 6 % ". ", 4 % ", ", 1 % "\n"
 *
 * A stream is generated chunk by chunk (one chunk per 16 MiB block) so any block range can be
 * produced independently (multi-GPU block-range sharding) and in parallel.
 */
#include <math.h>
#include <stdint.h>
#include <stdlib.h>
#include <string.h>

#define ZT_VOCAB 65536
#define ZT_MAXW 14

static uint8_t  g_word[ZT_VOCAB][ZT_MAXW];
static uint8_t  g_wlen[ZT_VOCAB];
static uint64_t g_cdf[ZT_VOCAB];      /* fixed-point 2^53-scaled cumulative Zipf weights */
static int      g_ready;

static inline uint64_t splitmix64(uint64_t* s) {
    uint64_t z = (*s += 0x9E3779B97F4A7C15ull);
    z = (z ^ (z >> 30)) * 0xBF58476D1CE4E5B9ull;
    z = (z ^ (z >> 27)) * 0x94D049BB133111EBull;
    return z ^ (z >> 31);
}

/* English letter frequencies, per 10000 (a..z). */
static const uint16_t k_letter[26] = {817, 149, 278, 425, 1270, 223, 202, 609, 697, 15, 77, 403, 241,
                                      675, 751, 193, 10,  599, 633,  906, 276, 98,  236, 15, 197, 7};

static void zt_init(void) {
    if (g_ready) return;
    uint32_t lcdf[26], tot = 0;
    for (int i = 0; i < 26; i++) { tot += k_letter[i]; lcdf[i] = tot; }
    /* Poisson(4.5) CDF scaled to 2^32 */
    double pc[ZT_MAXW], p = exp(-4.5), acc = 0;
    for (int k = 0; k < ZT_MAXW; k++) { acc += p; pc[k] = acc; p = p * 4.5 / (k + 1); }
    for (int w = 0; w < ZT_VOCAB; w++) {
        uint64_t s = 0x766F636162ull + (uint64_t)w * 0x100000001B3ull;   /* "vocab" */
        double u = (double)(splitmix64(&s) >> 11) * (1.0 / 9007199254740992.0);
        int k = 0;
        while (k < ZT_MAXW - 1 && u > pc[k]) k++;
        int len = 1 + k;
        g_wlen[w] = (uint8_t)len;
        for (int j = 0; j < len; j++) {
            uint32_t r = (uint32_t)(splitmix64(&s) % tot);
            int c = 0;
            while (r >= lcdf[c]) c++;
            g_word[w][j] = (uint8_t)('a' + c);
        }
    }
    double z = 0;
    for (int k = 1; k <= ZT_VOCAB; k++) z += pow((double)k, -1.1);
    double run = 0;
    for (int k = 1; k <= ZT_VOCAB; k++) {
        run += pow((double)k, -1.1);
        g_cdf[k - 1] = (uint64_t)(run / z * 9007199254740992.0);
    }
    g_cdf[ZT_VOCAB - 1] = ~0ull;
    g_ready = 1;
}

/* Fill out[0..n) with chunk `chunk` of the synthetic stream. */
void zt_generate_chunk(uint8_t* out, size_t n, uint64_t chunk) {
    zt_init();
    uint64_t s = 0x5A4C4E47ull + chunk;
    size_t o = 0;
    while (o < n) {
        uint64_t r = splitmix64(&s);
        uint64_t u = r >> 11;
        int lo = 0, hi = ZT_VOCAB - 1;
        while (lo < hi) { int mid = (lo + hi) >> 1; if (g_cdf[mid] > u) hi = mid; else lo = mid + 1; }
        int len = g_wlen[lo];
        for (int j = 0; j < len && o < n; j++) out[o++] = g_word[lo][j];
        unsigned sep = (unsigned)(r & 0x7FF) % 100;      /* low bits, independent of the word draw */
        if (sep < 89) { if (o < n) out[o++] = ' '; }
        else if (sep < 95) { if (o < n) out[o++] = '.'; if (o < n) out[o++] = ' '; }
        else if (sep < 99) { if (o < n) out[o++] = ','; if (o < n) out[o++] = ' '; }
        else { if (o < n) out[o++] = '\n'; }
    }
}

/* Fill out[0..n) with the stream whose 16 MiB block b is chunk (first_chunk + b). */
void zt_generate(uint8_t* out, size_t n, uint64_t first_chunk) {
    const size_t blk = 16777216;
    zt_init();
    size_t nb = (n + blk - 1) / blk;
#pragma omp parallel for schedule(dynamic)
    for (long b = 0; b < (long)nb; b++) {
        size_t off = (size_t)b * blk, len = n - off < blk ? n - off : blk;
        zt_generate_chunk(out + off, len, first_chunk + (uint64_t)b);
    }
}

/* de Bruijn sequence B(256, 3) (Fredricksen-Kessler-Maiorana): 16,777,216 bytes in which every 3-byte window occurs once.
 * XOR 0x55 it is a block the reference parses into one token per byte -- the worst case of the token pools (tests). */
static uint8_t* g_db_out;
static size_t   g_db_n;
static int      g_db_a[4];
static void zt_db(int t, int p) {
    if (t > 3) {
        if (3 % p == 0) for (int i = 1; i <= p; i++) g_db_out[g_db_n++] = (uint8_t)g_db_a[i];
        return;
    }
    g_db_a[t] = g_db_a[t - p];
    zt_db(t + 1, p);
    for (int j = g_db_a[t - p] + 1; j < 256; j++) { g_db_a[t] = j; zt_db(t + 1, t); }
}
size_t zt_debruijn3(uint8_t* out) {
    g_db_out = out; g_db_n = 0;
    g_db_a[0] = g_db_a[1] = g_db_a[2] = g_db_a[3] = 0;
    zt_db(1, 1);
    return g_db_n;
}
