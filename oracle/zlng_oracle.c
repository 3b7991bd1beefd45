/*
 * zlng_oracle.c -- CPU restatement of libzling's Encode/Decode block pipeline.
 *
 * TEST INFRASTRUCTURE ONLY (see zlng_oracle.h).  Parity: pinned against the
 * reference compiled from /root/reference by oracle/Makefile into oracle/_ref/
 * (tests/test_oracle_vs_ref.py) and against tests/golden/ (made with that build).
 *
 * Written from the behavioural spec (SURVEY.md appendices A-D); organised as
 * explicit stages -- parse, rank, histogram, lengths, codes, pack, frame -- so
 * every HIP kernel has a stage-level checker.  Citations are file:line in the
 * upstream richox/libzling tree.
 */
#include "zlng_oracle.h"

#include <stdlib.h>
#include <string.h>

/* ------------------------------------------------------------------ tables */

/* src/tables/gen.py:33-48 (the 256 enwik8-tuned initial ranks). */
const uint8_t zo_mtfinit[256] = {
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

static uint8_t  g_mtfnext[256];
static uint8_t  g_idx_code[ZO_RING];
static uint16_t g_idx_base[ZO_NSYM2];
static uint8_t  g_idx_blen[ZO_NSYM2];
static int      g_tables_ready;

/* src/tables/gen.py:10-18 (matchidx_*), :52-56 (mtfnext = int(0.95 i) / int(0.55 i)). */
static void tables_init(void) {
    if (g_tables_ready) return;
    for (int i = 0; i < 256; i++) g_mtfnext[i] = (uint8_t)(i < 128 ? (i * 95) / 100 : (i * 55) / 100);
    int filled = 0, ncode = 0;
    while (filled < ZO_RING) {
        int bl = ncode < 4 ? 0 : (ncode < 18 ? (ncode - 2) / 2 : 8);
        g_idx_blen[ncode] = (uint8_t)bl;
        g_idx_base[ncode] = (uint16_t)filled;
        for (int k = 0; k < (1 << bl); k++) g_idx_code[filled++] = (uint8_t)ncode;
        ncode++;
    }
    g_tables_ready = 1;
}
const uint8_t*  zo_mtfnext(void)       { tables_init(); return g_mtfnext; }
const uint8_t*  zo_matchidx_code(void) { tables_init(); return g_idx_code; }
const uint16_t* zo_matchidx_base(void) { tables_init(); return g_idx_base; }
const uint8_t*  zo_matchidx_blen(void) { tables_init(); return g_idx_blen; }

/* ------------------------------------------------------------ stream state */

typedef struct {                /* src/libzling_lz.h:98-103 ZlingEncodeBucket */
    uint16_t suffix[ZO_RING];
    uint32_t offset[ZO_RING];
    uint16_t head;
    uint16_t hash[ZO_HASH];
} zo_bucket;

typedef struct {                /* src/libzling_lz.h:50-57 ZlingMTFEncoder */
    uint8_t table[256];
    uint8_t index[256];
} zo_mtf;

struct zo_stream {
    zo_bucket bucket[256];
    zo_mtf    mtf[256];
    int       level;            /* requested level            */
    int       current_level;    /* src/libzling.cpp:185, 261-266 (carried across blocks, H3) */
};

void zo_reset_buckets(zo_stream* s) {                 /* src/libzling_lz.cpp:197-209 */
    for (int c = 0; c < 256; c++) {
        zo_bucket* b = &s->bucket[c];
        memset(b->offset, 0, sizeof b->offset);
        memset(b->suffix, 0xFF, sizeof b->suffix);
        memset(b->hash, 0xFF, sizeof b->hash);
        b->head = 0;
    }
}

zo_stream* zo_stream_new(int level) {
    if (level < 0 || level > 4) return NULL;
    tables_init();
    zo_stream* s = (zo_stream*)malloc(sizeof *s);
    if (!s) return NULL;
    zo_reset_buckets(s);
    for (int c = 0; c < 256; c++) {                   /* src/libzling_lz.cpp:106-111 */
        memcpy(s->mtf[c].table, zo_mtfinit, 256);
        for (int i = 0; i < 256; i++) s->mtf[c].index[zo_mtfinit[i]] = (uint8_t)i;
    }
    s->level = level;
    s->current_level = level;
    return s;
}
void zo_stream_free(zo_stream* s) { free(s); }

/* current_level (src/libzling.cpp:185): part of the stream state a block-range hand-off carries */
int  zo_stream_get_level(const zo_stream* s) { return s->current_level; }
void zo_stream_set_level(zo_stream* s, int level) { s->current_level = level; }
void zo_stream_get_mtf(const zo_stream* s, uint8_t t[256 * 256]) {
    for (int c = 0; c < 256; c++) memcpy(t + 256 * c, s->mtf[c].table, 256);
}
void zo_stream_set_mtf(zo_stream* s, const uint8_t t[256 * 256]) {
    for (int c = 0; c < 256; c++) {
        memcpy(s->mtf[c].table, t + 256 * c, 256);
        for (int i = 0; i < 256; i++) s->mtf[c].index[s->mtf[c].table[i]] = (uint8_t)i;
    }
}

/* -------------------------------------------------------------- ROLZ parse */

static inline uint32_t le32(const uint8_t* p) { uint32_t v; memcpy(&v, p, 4); return v; }

/* src/libzling_lz.cpp:55-57 HashContext */
static inline uint32_t hash4(const uint8_t* p) { return le32(p) + p[2] * 137u + p[3] * 13337u; }

/* src/libzling_lz.cpp:66-89 GetCommonLength: 0 unless the first four bytes agree,
 * else the byte-wise common prefix capped at 259. */
static inline int common_len(const uint8_t* a, const uint8_t* b) {
    if (le32(a) != le32(b)) return 0;
    int n = 4;
    while (n + 4 <= ZO_MATCH_MAX && le32(a + n) == le32(b + n)) n += 4;
    while (n < ZO_MATCH_MAX && a[n] == b[n]) n++;
    return n;
}

/* src/libzling_lz.cpp:291-316 MatchLazy */
static inline int lazy_probe(const zo_stream* s, const uint8_t* buf, int pos, int maxlen, int depth) {
    const zo_bucket* b = &s->bucket[buf[pos - 1]];
    int node = b->hash[hash4(buf + pos) % ZO_HASH];
    if (node == 65535) return 0;
    int m = maxlen - 3;
    for (int i = 0; i < depth; i++) {
        uint32_t off = b->offset[node] & 0xFFFFFF;
        if (le32(buf + pos + m) == le32(buf + off + m)) return 1;
        node = b->suffix[node];
        if (node == 65535 || off <= (b->offset[node] & 0xFFFFFF)) break;
    }
    return 0;
}

/* src/libzling_lz.cpp:211-289 MatchAndUpdate: insert first, then walk <= depth nodes. */
static inline int match_and_update(zo_stream* s, const uint8_t* buf, int pos, int depth, int lazy1, int lazy2,
                                   int* match_idx, int* match_len) {
    uint32_t h = hash4(buf + pos);
    uint32_t chk = (h / ZO_HASH) % 256;
    uint32_t hc = h % ZO_HASH;
    zo_bucket* b = &s->bucket[buf[pos - 1]];
    int node = b->hash[hc];

    b->head = (uint16_t)((b->head + 1) & (ZO_RING - 1));
    b->suffix[b->head] = (uint16_t)node;
    b->offset[b->head] = (uint32_t)pos | chk << 24;
    b->hash[hc] = b->head;

    if (node == 65535 || node == b->head) return 0;

    int maxlen = ZO_MATCH_MIN - 1, maxnode = 0;
    for (int i = 0; i < depth; i++) {
        uint32_t off = b->offset[node] & 0xFFFFFF;
        if ((b->offset[node] >> 24) == chk && buf[pos + maxlen] == buf[off + maxlen]) {
            int len = common_len(buf + pos, buf + off);
            if (len > maxlen) {
                maxnode = node;
                maxlen = len;
                if (maxlen == ZO_MATCH_MAX) break;
            }
        }
        node = b->suffix[node];
        if (node == 65535 || off <= (b->offset[node] & 0xFFFFFF)) break;
    }
    if (maxlen < ZO_MATCH_MIN) return 0;
    if (maxlen < ZO_LAZY_LIMIT) {
        if (lazy1 > 0 && lazy_probe(s, buf, pos + 1, maxlen, lazy1)) return 0;
        if (lazy2 > 0 && lazy_probe(s, buf, pos + 2, maxlen, lazy2)) return 0;
    }
    *match_len = maxlen;
    *match_idx = (b->head - maxnode) & (ZO_RING - 1);
    return 1;
}

/* src/libzling_lz.cpp:112-117 ZlingMTFEncoder::Encode */
static inline int mtf_encode(zo_mtf* m, int c) {
    int i = m->index[c];
    int n = g_mtfnext[i];
    uint8_t d = m->table[n];
    m->index[c] = (uint8_t)n;
    m->index[d] = (uint8_t)i;
    m->table[i] = d;
    m->table[n] = (uint8_t)c;
    return i;
}

/* src/libzling_lz.cpp:128-137: level -> (depth, lazy1 depth, lazy2 depth) */
static const int k_level_cfg[5][3] = {{2, 1, 0}, {4, 1, 0}, {6, 2, 0}, {8, 3, 1}, {16, 4, 2}};

/* src/libzling_lz.cpp:139-195 EncodeImpl */
int zo_parse_subblock(zo_stream* s, int level, const uint8_t* ibuf, int ilen, int* encpos, uint32_t* tok,
                      int* rlen, int apply_mtf) {
    const int depth = k_level_cfg[level][0], lazy1 = k_level_cfg[level][1], lazy2 = k_level_cfg[level][2];
    const int olen = ZO_SUBBLOCK_SYMS;
    int ipos = *encpos, opos = 0, nt = 0;
    uint16_t mru[256][2];
    memset(mru, 0, sizeof mru);

    if (ipos == 0 && opos < olen && ipos < ilen) { tok[nt++] = ibuf[ipos++] | ZO_TOK_RAWCTX << 16; opos++; }
    if (ipos == 1 && opos < olen && ipos < ilen) { tok[nt++] = ibuf[ipos++] | ZO_TOK_RAWCTX << 16; opos++; }

    while (opos + 1 < olen && ipos < ilen) {
        int midx, mlen;
        if (ipos + ZO_SENTINEL < ilen && match_and_update(s, ibuf, ipos, depth, lazy1, lazy2, &midx, &mlen)) {
            tok[nt++] = (uint32_t)(258 + mlen - ZO_MATCH_MIN) | (uint32_t)midx << 16;
            opos += 2;
            ipos += mlen;
            uint16_t w = (uint16_t)(ibuf[ipos - 2] << 8 | ibuf[ipos - 1]);
            uint16_t* m = mru[ibuf[ipos - 3]];
            if (m[0] != w) { m[1] = m[0]; m[0] = w; }
            continue;
        }
        if (ipos + 1 < ilen) {
            uint16_t w = (uint16_t)(ibuf[ipos] << 8 | ibuf[ipos + 1]);
            uint16_t* m = mru[ibuf[ipos - 1]];
            if (m[0] == w) { tok[nt++] = 256; opos++; ipos += 2; continue; }
            if (m[1] == w) { tok[nt++] = 257; opos++; ipos += 2; m[1] = m[0]; m[0] = w; continue; }
        }
        {
            int ctx = ibuf[ipos - 1], c = ibuf[ipos];
            int sym = apply_mtf ? mtf_encode(&s->mtf[ctx], c) : c;
            tok[nt++] = (uint32_t)sym | (uint32_t)ctx << 16;
            opos++;
            ipos++;
            uint16_t* m = mru[ibuf[ipos - 3]];
            m[1] = m[0];
            m[0] = (uint16_t)(ibuf[ipos - 2] << 8 | ibuf[ipos - 1]);
        }
    }
    *encpos = ipos;
    *rlen = opos;
    return nt;
}

void zo_mtf_rank(zo_stream* s, uint32_t* tok, size_t ntok) {
    for (size_t i = 0; i < ntok; i++) {
        uint32_t t = tok[i], sym = t & 0xFFFF, aux = t >> 16;
        if (sym < 256 && aux != ZO_TOK_RAWCTX) tok[i] = (uint32_t)mtf_encode(&s->mtf[aux], (int)sym) | aux << 16;
    }
}

/* ----------------------------------------------------------------- Huffman */

/* src/libzling.cpp:219-224 */
void zo_histogram(const uint32_t* tok, size_t ntok, uint32_t freq1[ZO_NSYM1], uint32_t freq2[ZO_NSYM2]) {
    tables_init();
    memset(freq1, 0, sizeof(uint32_t) * ZO_NSYM1);
    memset(freq2, 0, sizeof(uint32_t) * ZO_NSYM2);
    for (size_t i = 0; i < ntok; i++) {
        uint32_t sym = tok[i] & 0xFFFF;
        freq1[sym]++;
        if (sym >= 258) freq2[g_idx_code[tok[i] >> 16]]++;
    }
}

/* libstdc++ binary-heap primitives (GCC 11 bits/stl_heap.h:134-147, 223-248, 253-265)
 * restated over an index array; comp(a,b) = weight[a] > weight[b], no tie-break key
 * (src/libzling_huffman.cpp:63-67). */
typedef struct { int w[2 * ZO_NSYM1]; int kid0[2 * ZO_NSYM1]; int kid1[2 * ZO_NSYM1]; int leaf[2 * ZO_NSYM1]; } hnodes;

static void heap_sift_up(int* h, const int* w, int hole, int top, int v) {
    int parent = (hole - 1) / 2;
    while (hole > top && w[h[parent]] > w[v]) { h[hole] = h[parent]; hole = parent; parent = (hole - 1) / 2; }
    h[hole] = v;
}
static void heap_adjust(int* h, const int* w, int hole, int len, int v) {
    const int top = hole;
    int kid = hole;
    while (kid < (len - 1) / 2) {
        kid = 2 * (kid + 1);
        if (w[h[kid]] > w[h[kid - 1]]) kid--;
        h[hole] = h[kid];
        hole = kid;
    }
    if ((len & 1) == 0 && kid == (len - 2) / 2) {
        kid = 2 * (kid + 1);
        h[hole] = h[kid - 1];
        hole = kid - 1;
    }
    heap_sift_up(h, w, hole, top, v);
}
static int heap_pop(int* h, const int* w, int* n) {     /* top(); pop() */
    int top = h[0];
    if (*n > 1) { int v = h[*n - 1]; h[*n - 1] = h[0]; heap_adjust(h, w, 0, *n - 1, v); }
    (*n)--;
    return top;
}

/* src/libzling_huffman.cpp:41-112 ZlingMakeLengthTable */
void zo_make_length_table(const uint32_t* freq, uint32_t* len, int n, int limit) {
    hnodes nd;
    int heap[ZO_NSYM1], stack[2 * ZO_NSYM1], depth[2 * ZO_NSYM1];
    memset(len, 0, sizeof(uint32_t) * (size_t)n);
    for (int scaling = 0;; scaling++) {
        int nn = 0, hn = 0;
        for (int i = 0; i < n; i++) {
            if (freq[i] > 0) {
                nd.w[nn] = (int)((freq[i] + ((1u << scaling) - 1)) >> scaling);
                nd.leaf[nn] = i;
                heap[hn++] = nn++;
            }
        }
        if (hn == 0) return;
        if (hn >= 2) for (int p = (hn - 2) / 2; p >= 0; p--) { int v = heap[p]; heap_adjust(heap, nd.w, p, hn, v); }
        while (hn > 1) {
            int a = heap_pop(heap, nd.w, &hn);
            int b = heap_pop(heap, nd.w, &hn);
            nd.w[nn] = nd.w[a] + nd.w[b];
            nd.leaf[nn] = -1;
            nd.kid0[nn] = a;
            nd.kid1[nn] = b;
            heap[hn] = nn;
            heap_sift_up(heap, nd.w, hn, 0, nn);
            hn++;
            nn++;
        }
        int sp = 0, maxlen = 0;
        stack[sp] = heap[0]; depth[sp++] = 0;
        while (sp) {
            int v = stack[--sp], d = depth[sp];
            if (nd.leaf[v] >= 0) {
                int l = d > 1 ? d : 1;
                len[nd.leaf[v]] = (uint32_t)l;
                if (l > maxlen) maxlen = l;
            } else {
                stack[sp] = nd.kid0[v]; depth[sp++] = d + 1;
                stack[sp] = nd.kid1[v]; depth[sp++] = d + 1;
            }
        }
        if (maxlen <= limit) return;
    }
}

/* src/libzling_huffman.cpp:114-138 ZlingMakeEncodeTable */
void zo_make_encode_table(const uint32_t* len, uint16_t* code, int n, int limit) {
    unsigned next = 0;
    for (int i = 0; i < n; i++) code[i] = 0;
    for (int l = 1; l <= limit; l++) {
        for (int i = 0; i < n; i++) if (len[i] == (uint32_t)l) code[i] = (uint16_t)next++;
        next *= 2;
    }
    for (int i = 0; i < n; i++) {
        unsigned v = code[i], r = 0;
        for (int b = 0; b < 16; b++) r |= ((v >> b) & 1u) << (15 - b);
        code[i] = (uint16_t)((r & 0xFFFF) >> (16 - len[i]));   /* len 0 -> shift 16 -> 0 */
    }
}

/* src/libzling.cpp:232-258 (nibble tables + LSB-first bit packing) */
size_t zo_pack_subblock(const uint32_t* tok, size_t ntok, const uint32_t* len1, const uint32_t* len2, uint8_t* out) {
    tables_init();
    uint16_t code1[ZO_NSYM1], code2[ZO_NSYM2];
    zo_make_encode_table(len1, code1, ZO_NSYM1, ZO_MAXLEN1);
    zo_make_encode_table(len2, code2, ZO_NSYM2, ZO_MAXLEN2);
    size_t o = 0;
    for (int i = 0; i < ZO_NSYM1; i += 2) out[o++] = (uint8_t)(len1[i] * 16 + len1[i + 1]);
    for (int i = 0; i < ZO_NSYM2; i += 2) out[o++] = (uint8_t)(len2[i] * 16 + len2[i + 1]);
    uint64_t acc = 0;
    int nb = 0;
    for (size_t i = 0; i < ntok; i++) {
        uint32_t sym = tok[i] & 0xFFFF;
        acc |= (uint64_t)code1[sym] << nb; nb += (int)len1[sym];
        if (sym >= 258) {
            uint32_t idx = tok[i] >> 16, c = g_idx_code[idx];
            acc |= (uint64_t)code2[c] << nb; nb += (int)len2[c];
            acc |= (uint64_t)(idx - g_idx_base[c]) << nb; nb += g_idx_blen[c];
        }
        while (nb >= 32) { out[o++] = (uint8_t)acc; out[o++] = (uint8_t)(acc >> 8); out[o++] = (uint8_t)(acc >> 16);
                           out[o++] = (uint8_t)(acc >> 24); acc >>= 32; nb -= 32; }
    }
    while (nb > 0) { out[o++] = (uint8_t)acc; acc >>= 8; nb -= 8; }
    return o;
}

/* ------------------------------------------------------------ block driver */

size_t zo_encode_bound(size_t n) {
    /* every u16 entry covers >= 1 input byte and every sub-block but the last of a block holds
     * >= 262143 entries, so a stream has <= n/262143 + nblk sub-blocks; an entry costs <= 2
     * payload bytes (15-bit literal code; 31 bits per 2-entry match). */
    size_t nblk = (n + ZO_BLOCK_IN - 1) / ZO_BLOCK_IN;
    size_t nsub = n / 262143 + nblk + 1;
    return nsub * (13 + 273 + 8) + 2 * n + nblk + 64;
}

static inline void put_be32(uint8_t* p, uint32_t v) { p[0] = (uint8_t)(v >> 24); p[1] = (uint8_t)(v >> 16); p[2] = (uint8_t)(v >> 8); p[3] = (uint8_t)v; }

/* src/libzling.cpp:187-284 (one iteration of the outer loop per 16 MiB block) */
int zo_encode_blocks(zo_stream* s, const uint8_t* in, size_t n, uint8_t* out, size_t cap, size_t* out_len) {
    size_t o = 0;
    uint32_t* tok = (uint32_t*)malloc(sizeof(uint32_t) * ZO_SUBBLOCK_SYMS);
    uint8_t* ibuf = (uint8_t*)malloc(ZO_BLOCK_IN + ZO_SENTINEL);
    uint8_t* payload = (uint8_t*)malloc(273 + (size_t)ZO_SUBBLOCK_SYMS * 4 + 16);
    if (!tok || !ibuf || !payload) { free(tok); free(ibuf); free(payload); return -1; }
    int rc = 0;
    for (size_t base = 0; base < n; base += ZO_BLOCK_IN) {
        int ilen = (int)(n - base < ZO_BLOCK_IN ? n - base : ZO_BLOCK_IN);
        memcpy(ibuf, in + base, (size_t)ilen);
        memset(ibuf + ilen, 0, ZO_SENTINEL);
        zo_reset_buckets(s);
        int encpos = 0;
        while (encpos < ilen) {
            int old = encpos, rlen;
            uint32_t f1[ZO_NSYM1], f2[ZO_NSYM2], l1[ZO_NSYM1], l2[ZO_NSYM2];
            int nt = zo_parse_subblock(s, s->current_level, ibuf, ilen, &encpos, tok, &rlen, 1);
            zo_histogram(tok, (size_t)nt, f1, f2);
            zo_make_length_table(f1, l1, ZO_NSYM1, ZO_MAXLEN1);
            zo_make_length_table(f2, l2, ZO_NSYM2, ZO_MAXLEN2);
            size_t olen = zo_pack_subblock(tok, (size_t)nt, l1, l2, payload);
            /* src/libzling.cpp:261-266 */
            s->current_level = (1.0 * (double)olen / (encpos - old + 1) > 0.95) ? 0 : s->level;
            if (o + 13 + olen + 1 > cap) { rc = -1; goto done; }
            out[o++] = 1;                                   /* src/libzling.cpp:200 */
            put_be32(out + o, (uint32_t)encpos); o += 4;    /* :269-271 */
            put_be32(out + o, (uint32_t)rlen); o += 4;
            put_be32(out + o, (uint32_t)olen); o += 4;
            memcpy(out + o, payload, olen); o += olen;
        }
        if (o + 1 > cap) { rc = -1; goto done; }
        out[o++] = 0;                                       /* :278 */
    }
done:
    free(tok); free(ibuf); free(payload);
    *out_len = o;
    return rc;
}

int zo_encode(const uint8_t* in, size_t n, int level, uint8_t* out, size_t cap, size_t* out_len) {
    zo_stream* s = zo_stream_new(level);
    if (!s) return -1;
    int rc = zo_encode_blocks(s, in, n, out, cap, out_len);
    zo_stream_free(s);
    return rc;
}

/* ------------------------------------------------------------------ decode */

/* src/libzling_huffman.cpp:140-153 ZlingMakeDecodeTable */
static void make_decode_table(const uint32_t* len, const uint16_t* code, uint16_t* lut, int n, int limit) {
    for (int i = 0; i < (1 << limit); i++) lut[i] = 0xFFFF;
    for (int c = 0; c < n; c++)
        if (len[c] > 0 && len[c] <= (uint32_t)limit)
            for (int i = code[c]; i < (1 << limit); i += 1 << len[c]) lut[i] = (uint16_t)c;
}

typedef struct { uint32_t offset[ZO_RING]; uint16_t head; } zo_dbucket;   /* src/libzling_lz.h:132-135 */

/* Does the block that starts at in[ip] lie complete in the input?  Walks flags and headers only, up to the block's 0x00 or the
 * first thing the decode loop below turns into an error by itself (a bad flag, sizes over the limits): those are met in stream
 * order there.  0 = the input ends inside the block. */
static int block_is_complete(const uint8_t* in, size_t n, size_t ip) {
    for (;;) {
        if (ip >= n) return 0;
        const int flag = in[ip++];
        if (flag != 1) return 1;                                        /* 0x00 closes it; anything else is an error, not a cut */
        if (ip + 12 > n) return 0;
        const uint32_t rlen = (uint32_t)in[ip + 4] << 24 | in[ip + 5] << 16 | in[ip + 6] << 8 | in[ip + 7];
        const uint32_t olen = (uint32_t)in[ip + 8] << 24 | in[ip + 9] << 16 | in[ip + 10] << 8 | in[ip + 11];
        ip += 12;
        if (rlen > ZO_SUBBLOCK_SYMS || olen > ZO_PAYLOAD_MAX || olen < ZO_TABLE_BYTES) return 1;
        if (ip + olen > n) return 0;
        ip += olen;
    }
}

/* Decoder state that outlives a block: the 256 literal tables (src/libzling_lz.cpp:119-121: set up once per decoder object, never
 * reset).  Everything else -- ring buckets, word MRU, the block buffer -- is per block or per sub-block and lives in the scratch. */
struct zo_dstream {
    uint8_t mtf[256][256];
    zo_dbucket bk[256];
    uint16_t tb[ZO_SUBBLOCK_SYMS + ZO_SENTINEL];
    uint8_t blk[ZO_BLOCK_IN + ZO_SENTINEL + 512];
    uint8_t pay[ZO_PAYLOAD_MAX + ZO_SENTINEL + 16];
    uint16_t lut1[1 << ZO_MAXLEN1];
};

zo_dstream* zo_dstream_new(void) {
    tables_init();
    zo_dstream* d = (zo_dstream*)calloc(1, sizeof(zo_dstream));
    if (d) for (int c = 0; c < 256; c++) memcpy(d->mtf[c], zo_mtfinit, 256);     /* src/libzling_lz.cpp:119-121 */
    return d;
}
void zo_dstream_free(zo_dstream* d) { free(d); }
void zo_dstream_get_mtf(const zo_dstream* d, uint8_t t[256 * 256]) { memcpy(t, d->mtf, 256 * 256); }
void zo_dstream_set_mtf(zo_dstream* d, const uint8_t t[256 * 256]) { memcpy(d->mtf, t, 256 * 256); }

/* src/libzling.cpp:293-427 Decode + src/libzling_lz.cpp:318-399 ZlingRolzDecoder.
 *
 * On a VALID stream this is the reference, statement by statement.  On a hostile one the reference has behaviour that no
 * restatement can share -- it reads uninitialised or stale heap bytes, or writes past its buffers -- and there this decoder
 * REJECTS instead (each rule sets a bit of *flags, so that the differential tests know which verdicts can be compared with the
 * real reference and which cannot):
 *   ZO_DEV_TRUNC     the input ends inside a block (the reference's GetChar/GetUInt32 return EOF garbage and it decodes on;
 *                    src/libzling.cpp:312-334): ZO_E_TRUNC, before any sub-block of that block is looked at
 *   ZO_DEV_SIGNED    rlen or olen >= 2^31 (the reference compares them as int, :326: negative values pass): ZO_E_BLOCKSIZE
 *   ZO_DEV_SHORT     olen < 273 (the reference takes the missing table bytes from whatever obuf held, :347-356): ZO_E_LZ
 *   ZO_DEV_ENCPOS    encpos > 16 MiB (the reference's replay would write past ibuf before its size test, lz.cpp:363-373):
 *                    ZO_E_LZ, after the sub-block's Huffman stream has been checked like the reference checks it
 *   ZO_DEV_OPENING   one of the two block-opening u16 entries is a match symbol (the reference copies entries as raw bytes
 *                    there, lz.cpp:327-328, which for a two-entry match splits the pair and re-reads the index as a symbol --
 *                    lengths up to 3,841 past the buffer's sentinel): ZO_E_LZ
 *   ZO_DEV_SELF      a match with ring index 0 (names the slot the token itself has just written, lz.cpp:388-399: the copy's
 *                    source is the destination, i.e. bytes no one has written): ZO_E_LZ
 *   ZO_DEV_ENTRIES   the sub-blocks of one block hold more than 16 Mi + 64 u16 entries (every entry stands for at least one output
 *                    byte, so such a block cannot land on its encpos: the reference, which sizes nothing by the block's entries,
 *                    still decodes the Huffman stream of the sub-block that crosses the limit before its replay fails, and a damaged
 *                    table there is "bad code" to it): ZO_E_LZ at that sub-block's header, its Huffman stream unread -- the HIP
 *                    decoder sizes its token pool by this bound (csrc/decode.hip, k_frame_walk `too_many`)
 *   ZO_DEV_OVERREAD  (no rejection) the bit reader consumed bits behind the payload's end: zeros here, stale obuf bytes in
 *                    the reference (:369-374) -- the symbols decoded from there on may differ
 * Everything else -- bad flags, sizes over the limits, codes without a symbol in either alphabet, over-subscribed length sets
 * (last symbol in table order wins, with the 10-bit fast table of :361, 376-379 in front of the 15-bit one), indices >= 4096,
 * lengths that miss encpos -- is the reference's own behaviour and is restated exactly.
 * (Test infrastructure, like everything in this file: never linked into or called by the product.) */

/* What a decode reached (tests/hostile.py's "big structures" streams are checked to reach them): set by zo_decode_stats around a
 * plain decode, never read by the decoder itself. */
static __thread zo_dstats* g_stats = NULL;

/* One pass of the reference's outer loop (src/libzling.cpp:306-420): the block that starts at in[*ipp].  On success the block's
 * bytes are d->blk[0 .. *size) and *ipp stands behind its 0x00 (or at n).  On an error the tables are NOT rolled back (the
 * reference throws and its decoder object dies with the call); zo_dstream_decode_block below does that for callers that go on. */
static int decode_one_block(zo_dstream* d, const uint8_t* in, size_t n, size_t* ipp, int* size, uint32_t* flagsp) {
    zo_dbucket* bk = d->bk;
    uint8_t (*mtf)[256] = d->mtf;
    uint16_t* tb = d->tb;
    uint8_t* blk = d->blk;
    uint8_t* pay = d->pay;
    uint16_t* lut1 = d->lut1;
    uint16_t lut1f[1 << ZO_MAXLEN1_FAST];
    int rc = ZO_E_OK;
    uint32_t flags = *flagsp;
    size_t ip = *ipp;
    {
        int decpos = 0;
        uint64_t blk_entries = 0;                                       /* u16 entries of the sub-blocks of this block so far */
        if (!block_is_complete(in, n, ip)) { flags |= ZO_DEV_TRUNC; rc = ZO_E_TRUNC; goto done; }
        for (int c = 0; c < 256; c++) { memset(bk[c].offset, 0, sizeof bk[c].offset); bk[c].head = 0; }
        if (g_stats) memset(g_stats->inserts_scratch, 0, sizeof g_stats->inserts_scratch);
        while (ip < n) {
            int flag = in[ip++];
            if (flag != 0 && flag != 1) { rc = ZO_E_FLAG; goto done; }  /* :315-317 */
            if (flag == 0) break;
            uint32_t encpos = (uint32_t)in[ip] << 24 | in[ip + 1] << 16 | in[ip + 2] << 8 | in[ip + 3];
            uint32_t rlen = (uint32_t)in[ip + 4] << 24 | in[ip + 5] << 16 | in[ip + 6] << 8 | in[ip + 7];
            uint32_t olen = (uint32_t)in[ip + 8] << 24 | in[ip + 9] << 16 | in[ip + 10] << 8 | in[ip + 11];
            ip += 12;
            if (rlen > ZO_SUBBLOCK_SYMS || olen > ZO_PAYLOAD_MAX) {     /* :326-328 */
                if ((rlen | olen) & 0x80000000u) flags |= ZO_DEV_SIGNED;
                rc = ZO_E_BLOCKSIZE; goto done;
            }
            if (olen < ZO_TABLE_BYTES) { flags |= ZO_DEV_SHORT; rc = ZO_E_LZ; goto done; }
            if (blk_entries + rlen > (uint64_t)ZO_BLOCK_IN + 64) { flags |= ZO_DEV_ENTRIES; rc = ZO_E_LZ; goto done; }
            blk_entries += rlen;
            memcpy(pay, in + ip, olen);
            memset(pay + olen, 0, 16);
            ip += olen;

            uint32_t l1[ZO_NSYM1], l2[ZO_NSYM2];
            uint16_t c1[ZO_NSYM1], c2[ZO_NSYM2], lut2[1 << ZO_MAXLEN2];
            size_t pp = 0;
            for (int i = 0; i < ZO_NSYM1; i += 2) { l1[i] = pay[pp] / 16; l1[i + 1] = pay[pp] % 16; pp++; }
            for (int i = 0; i < ZO_NSYM2; i += 2) { l2[i] = pay[pp] / 16; l2[i + 1] = pay[pp] % 16; pp++; }
            zo_make_encode_table(l1, c1, ZO_NSYM1, ZO_MAXLEN1);
            zo_make_encode_table(l2, c2, ZO_NSYM2, ZO_MAXLEN2);
            make_decode_table(l1, c1, lut1, ZO_NSYM1, ZO_MAXLEN1);      /* :358-365: two levels for alphabet 1 */
            make_decode_table(l1, c1, lut1f, ZO_NSYM1, ZO_MAXLEN1_FAST);
            make_decode_table(l2, c2, lut2, ZO_NSYM2, ZO_MAXLEN2);

            uint64_t acc = 0, loaded = 0;
            int nb = 0;
            for (uint32_t i = 0; i < rlen; i++) {                       /* :368-402 */
                if (nb < 32) {
                    uint32_t w = 0;
                    for (int k = 0; k < 4; k++) w |= (uint32_t)(pp < olen ? pay[pp] : 0) << (8 * k), pp++;
                    acc |= (uint64_t)w << nb; nb += 32; loaded += 32;
                }
                uint32_t sym = lut1f[acc & ((1u << ZO_MAXLEN1_FAST) - 1)];     /* :376-379 */
                if (sym == 0xFFFF) sym = lut1[acc & 0x7FFF];
                if (sym >= ZO_NSYM1) { rc = ZO_E_CODE1; goto done; }
                acc >>= l1[sym]; nb -= (int)l1[sym];
                tb[i] = (uint16_t)sym;
                if (sym >= 258) {
                    uint32_t c = lut2[acc & 0xFF];
                    if (c >= ZO_NSYM2) { rc = ZO_E_CODE2; goto done; }
                    acc >>= l2[c]; nb -= (int)l2[c];
                    uint32_t bits = (uint32_t)(acc & ((1u << g_idx_blen[c]) - 1));
                    acc >>= g_idx_blen[c]; nb -= g_idx_blen[c];
                    uint32_t idx = g_idx_base[c] + bits;
                    if (idx >= ZO_RING) { rc = ZO_E_EXBITS; goto done; }      /* :396-399 (the largest index the 32 codes can name is
                                                                                 * 3840 + 255: the reference's test never fires either) */
                    tb[++i] = (uint16_t)idx;                                  /* may be entry `rlen` itself: the reference stores it behind
                                                                                 * the counted entries too, and its replay reads it there */
                }
            }
            if (loaded - (uint64_t)nb > 8ull * (olen - ZO_TABLE_BYTES)) flags |= ZO_DEV_OVERREAD;
            if (encpos > ZO_BLOCK_IN) { flags |= ZO_DEV_ENCPOS; rc = ZO_E_LZ; goto done; }

            /* ZlingRolzDecoder::Decode, src/libzling_lz.cpp:318-376 */
            uint16_t mru[256][2];
            memset(mru, 0, sizeof mru);
            int opos = decpos;
            uint32_t ti = 0;
            while (opos < 2 && ti < rlen) {                              /* :327-328: the block's two opening entries, raw */
                if (tb[ti] >= 258) { flags |= ZO_DEV_OPENING; rc = ZO_E_LZ; goto done; }
                blk[opos++] = (uint8_t)tb[ti++];
            }
            while (ti < rlen) {
                uint32_t v = tb[ti];
                if (opos + ZO_MATCH_MAX + 4 > ZO_BLOCK_IN + ZO_SENTINEL) { rc = ZO_E_LZ; goto done; }   /* (opos <= encpos <= 16 MiB: never) */
                zo_dbucket* b = &bk[blk[opos - 1]];
                b->head = (uint16_t)((b->head + 1) & (ZO_RING - 1));   /* GetMatchAndUpdate :388-399 */
                b->offset[b->head] = (uint32_t)opos;
                if (g_stats) {
                    uint32_t* cnt = g_stats->inserts_scratch;
                    if (++cnt[blk[opos - 1]] > g_stats->max_inserts_one_context) g_stats->max_inserts_one_context = cnt[blk[opos - 1]];
                }
                if (v < 256) {
                    uint8_t* t = mtf[blk[opos - 1]];                     /* ZlingMTFDecoder::Decode :122-126 */
                    uint8_t c = t[v], nx = g_mtfnext[v];
                    t[v] = t[nx]; t[nx] = c;
                    blk[opos++] = c; ti++;
                    if (g_stats) { g_stats->tokens++; g_stats->literals++; if (g_stats->writer) g_stats->writer[opos - 1] = (uint32_t)g_stats->tokens; }
                    mru[blk[opos - 3]][1] = mru[blk[opos - 3]][0];
                    mru[blk[opos - 3]][0] = (uint16_t)(blk[opos - 2] << 8 | blk[opos - 1]);
                } else if (v == 256) {
                    uint16_t w = mru[blk[opos - 1]][0];
                    blk[opos++] = (uint8_t)(w >> 8); blk[opos++] = (uint8_t)w; ti++;
                    if (g_stats) { g_stats->tokens++; g_stats->words++; if (g_stats->writer) g_stats->writer[opos - 2] = g_stats->writer[opos - 1] = (uint32_t)g_stats->tokens; }
                } else if (v == 257) {
                    uint16_t w = mru[blk[opos - 1]][1];
                    blk[opos++] = (uint8_t)(w >> 8); blk[opos++] = (uint8_t)w; ti++;
                    if (g_stats) { g_stats->tokens++; g_stats->words++; if (g_stats->writer) g_stats->writer[opos - 2] = g_stats->writer[opos - 1] = (uint32_t)g_stats->tokens; }
                    mru[blk[opos - 3]][1] = mru[blk[opos - 3]][0];
                    mru[blk[opos - 3]][0] = (uint16_t)(blk[opos - 2] << 8 | blk[opos - 1]);
                } else {
                    int mlen = (int)v - 258 + ZO_MATCH_MIN;
                    uint32_t idx = tb[ti + 1];
                    ti += 2;
                    if (idx == 0) { flags |= ZO_DEV_SELF; rc = ZO_E_LZ; goto done; }
                    uint32_t src = b->offset[(b->head - idx) & (ZO_RING - 1)];
                    if (g_stats) {
                        const uint32_t dist = (uint32_t)opos - src;
                        g_stats->matches++;
                        if (dist > g_stats->max_distance) g_stats->max_distance = dist;
                        if (dist > 131072) g_stats->far_matches++;
                        if (dist > 65536) g_stats->beyond_window++;
                        if (((uint32_t)opos >> 16) != (((uint32_t)opos + (uint32_t)mlen - 1) >> 16)) g_stats->dst_straddles_64k++;
                        if ((src >> 16) != ((src + (uint32_t)mlen - 1) >> 16)) g_stats->src_straddles_64k++;
                        if (g_stats->inserts_scratch[blk[opos - 1]] > ZO_RING && idx > 0) g_stats->matches_in_wrapped_ring++;
                    }
                    if (g_stats) {
                        g_stats->tokens++; g_stats->match_bytes += (uint64_t)mlen;
                        if (g_stats->writer && opos >= 2) {
                            uint32_t* wr = g_stats->writer;
                            /* the copy's last source byte; for a self-overlapping copy it is a byte this very copy writes: no wait */
                            const uint32_t last = src + (uint32_t)mlen - 1;
                            if (last < (uint32_t)opos) {
                                uint64_t lag = g_stats->tokens - (src + (uint32_t)mlen - 1 < 2 ? 0 : wr[last]);
                                int k = 0; while (lag > 1 && k < 23) { lag >>= 1; k++; }
                                g_stats->lag_hist[k]++;
                            } else g_stats->lag_hist[23]++;               /* [23]: self-overlap */
                            for (int k = 0; k < mlen; k++) wr[opos + k] = (uint32_t)g_stats->tokens;
                        }
                    }
                    for (int k = 0; k < mlen; k++) blk[opos + k] = blk[src + k];   /* :91-104 forward copy */
                    opos += mlen;
                    uint16_t w = (uint16_t)(blk[opos - 2] << 8 | blk[opos - 1]);
                    if (mru[blk[opos - 3]][0] != w) { mru[blk[opos - 3]][1] = mru[blk[opos - 3]][0]; mru[blk[opos - 3]][0] = w; }
                }
                if ((uint32_t)opos > encpos) { rc = ZO_E_LZ; goto done; }
            }
            if ((uint32_t)opos != encpos) { rc = ZO_E_LZ; goto done; }
            decpos = opos;
        }
        *size = decpos;
    }
done:
    *ipp = ip;
    *flagsp = flags;
    return rc;
}

int zo_dstream_decode_block(zo_dstream* d, const uint8_t* in, size_t n, size_t* ip, uint8_t* out, size_t cap, size_t* out_len, uint32_t* flags_out) {
    static uint8_t saved[256][256];                                    /* (the checker is single-threaded per process in the tests that use this) */
    uint32_t flags = 0;
    size_t p = *ip;
    int size = 0;
    *out_len = 0;
    memcpy(saved, d->mtf, sizeof saved);
    int rc = decode_one_block(d, in, n, &p, &size, &flags);
    if (rc == ZO_E_OK && (size_t)size > cap) rc = ZO_E_CAP;
    if (flags_out) *flags_out = flags;
    if (rc != ZO_E_OK) { memcpy(d->mtf, saved, sizeof saved); return rc; }
    memcpy(out, d->blk, (size_t)size);
    *out_len = (size_t)size;
    *ip = p;
    return ZO_E_OK;
}

int zo_decode_ex(const uint8_t* in, size_t n, uint8_t* out, size_t cap, size_t* out_len, uint32_t* flags_out) {
    zo_dstream* d = zo_dstream_new();
    int rc = ZO_E_OK;
    uint32_t flags = 0;
    size_t ip = 0, op = 0;
    if (!d) { rc = ZO_E_CAP; goto done; }
    while (ip < n) {                                                    /* one output block */
        int decpos = 0;
        rc = decode_one_block(d, in, n, &ip, &decpos, &flags);
        if (rc != ZO_E_OK) goto done;
        if (op + (size_t)decpos > cap) { rc = ZO_E_CAP; goto done; }
        memcpy(out + op, d->blk, (size_t)decpos);
        op += (size_t)decpos;
    }
done:
    zo_dstream_free(d);
    *out_len = op;
    if (flags_out) *flags_out = flags;
    return rc;
}

int zo_decode_stats(const uint8_t* in, size_t n, uint8_t* out, size_t cap, size_t* out_len, zo_dstats* st) {
    uint32_t* writer = st->writer;
    memset(st, 0, sizeof *st);
    st->writer = writer;
    g_stats = st;
    const int rc = zo_decode_ex(in, n, out, cap, out_len, NULL);
    g_stats = NULL;
    return rc;
}

int zo_decode(const uint8_t* in, size_t n, uint8_t* out, size_t cap, size_t* out_len) {
    return zo_decode_ex(in, n, out, cap, out_len, NULL);
}
