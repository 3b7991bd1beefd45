/* zlng_stub.c -- TEST INFRASTRUCTURE, never shipped and never loaded by the product: a stand-in for libzlng_hip.so's context-level
 * C-ABI (include/zlng.h) on top of the CPU checker (oracle/zlng_oracle.c), so that the HOST logic above the ABI -- the C++ shim
 * (libzling_amd/cxx/libzling_shim.cpp), the group driver (libzling_amd/csrc/zlng_group.hip, host code only), tools/zling_demo and
 * tests/cxx/protocol_test -- can be compiled against it and exercised by `pytest -m "not gpu"` in a container without a GPU
 * (tests/test_shim_on_stub.py builds everything into tests/cxx/_stub/).  It restates the ABI's documented CONTRACT, not the
 * kernels: what a call reports (bytes, per-block ends, error codes and their order, the state it carries, what a failed call
 * leaves behind).  The GPU suite (-m gpu) holds the real library to the same expectations through the same programs.
 *
 * Contract restated (include/zlng.h, libzling_amd/csrc/zlng_api.hip):
 *   zlng_encode_parse    takes a copy of the range (the real call stages an H2D copy); zlng_encode_finish produces the framed bytes of
 *                        that range from the context's stream state (tables + current_level), advances the state, reports every
 *                        block's end offset; a failed finish (ZLNG_E_CAP) leaves the state as it was
 *   zlng_decode_blocks   decodes the whole blocks found in the prefix, at most max_blocks; complete good blocks in front of a bad
 *                        (or incomplete) one are reported with ZLNG_OK and the error is met at the head of the next call; errors
 *                        inside a block come in stream order (the checker's decoder is the reference's one loop); a buffer that
 *                        is too small reports nothing and leaves the state (ZLNG_E_CAP)
 */
#define _POSIX_C_SOURCE 200809L
#include <stdint.h>
#include <stdlib.h>
#include <string.h>
#include <time.h>

#include "../../include/zlng.h"
#include "../../oracle/zlng_oracle.h"

struct zlng_ctx {
    int is_encode, level, max_blocks;
    zo_stream* es;
    zo_dstream* ds;
    uint8_t* pending;
    size_t pending_len;
    uint8_t* staged;                                     /* zlng_encode_finish_staged -> zlng_encode_copy_out */
    size_t staged_len;
    uint8_t* last_in;                                    /* the range of the last finished encode call (zlng_debug_fetch) */
    size_t last_len;
    double last_ms;                                      /* wall time of the last finish / decode call (zlng_last_timings) */
};

static double now_ms(void) { struct timespec t; clock_gettime(CLOCK_MONOTONIC, &t); return t.tv_sec * 1e3 + t.tv_nsec * 1e-6; }

int zlng_device_count(void) { return 1; }

zlng_ctx* zlng_create(int device, int level, int is_encode, int max_blocks, int* err) {
    int dummy;
    if (!err) err = &dummy;
    *err = ZLNG_OK;
    if (device != 0) { *err = ZLNG_E_DEVICE; return NULL; }
    if ((is_encode && (level < 0 || level > 4)) || max_blocks < 1 || max_blocks > 240) { *err = ZLNG_E_ARG; return NULL; }
    zlng_ctx* c = (zlng_ctx*)calloc(1, sizeof *c);
    if (!c) { *err = ZLNG_E_NOMEM; return NULL; }
    c->is_encode = is_encode; c->level = is_encode ? level : 0; c->max_blocks = max_blocks;
    if (is_encode) c->es = zo_stream_new(level); else c->ds = zo_dstream_new();
    if (!c->es && !c->ds) { free(c); *err = ZLNG_E_NOMEM; return NULL; }
    return c;
}

void zlng_destroy(zlng_ctx* c) {
    if (!c) return;
    if (c->es) zo_stream_free(c->es);
    if (c->ds) zo_dstream_free(c->ds);
    free(c->pending);
    free(c->staged);
    free(c->last_in);
    free(c);
}

size_t zlng_encode_bound(size_t n) {                    /* zlng_api.hip: 1.5 B per input byte + headers (>= the checker's own bound for text) */
    const size_t nblk = (n + ZLNG_BLOCK_SIZE - 1) / ZLNG_BLOCK_SIZE;
    size_t a = n + n / 2 + nblk * (80 * 13 + 1) + 4096, b = zo_encode_bound(n);
    return a > b ? a : b;
}

int zlng_encode_parse(zlng_ctx* c, const uint8_t* in, size_t in_len) {
    if (!c || !c->is_encode || !in || in_len == 0) return ZLNG_E_ARG;
    if ((in_len + ZLNG_BLOCK_SIZE - 1) / ZLNG_BLOCK_SIZE > (size_t)c->max_blocks) return ZLNG_E_ARG;
    free(c->pending);
    c->pending = (uint8_t*)malloc(in_len);
    if (!c->pending) { c->pending_len = 0; return ZLNG_E_NOMEM; }
    memcpy(c->pending, in, in_len);
    c->pending_len = in_len;
    return ZLNG_OK;
}

int zlng_encode_finish(zlng_ctx* c, uint8_t* out, size_t out_cap, size_t* out_len, size_t* per_block_out_end) {
    if (!c || !c->pending || !c->pending_len || !out || !out_len) return ZLNG_E_ARG;
    *out_len = 0;
    static uint8_t saved[ZLNG_MTF_STATE];
    zo_stream_get_mtf(c->es, saved);
    const int lv = zo_stream_get_level(c->es);
    size_t n = 0;
    const double t0 = now_ms();
    const int zrc = zo_encode_blocks(c->es, c->pending, c->pending_len, out, out_cap, &n);
    c->last_ms = now_ms() - t0;
    if (zrc != 0) {
        zo_stream_set_mtf(c->es, saved);
        zo_stream_set_level(c->es, lv);
        return ZLNG_E_CAP;                               /* the range stays pending: the call can be repeated with a larger buffer */
    }
    if (per_block_out_end) {
        size_t p = 0, b = 0;
        while (p < n) {
            if (out[p] == 0) { p++; per_block_out_end[b++] = p; continue; }
            p += 13 + ((size_t)out[p + 9] << 24 | (size_t)out[p + 10] << 16 | (size_t)out[p + 11] << 8 | out[p + 12]);
        }
    }
    free(c->last_in); c->last_in = c->pending; c->last_len = c->pending_len;      /* kept for zlng_debug_fetch(8) */
    c->pending = NULL; c->pending_len = 0;
    *out_len = n;
    return ZLNG_OK;
}

/* Test hook, `what` = 8 only (literals per context of the last encode call: what bench.py's rank_chain line asks for): the range is
 * parsed again with the checker's stage API and the literal tokens counted by their context byte.  The other hooks look into the
 * kernels' buffers and have no counterpart here. */
int zlng_debug_fetch(zlng_ctx* c, int what, int blk, void* dst, size_t bytes) {
    (void)blk;
    if (!c || !c->is_encode || what != 8 || !dst || bytes > 256 * 4 || !c->last_in) return ZLNG_E_ARG;
    uint32_t cnt[256] = {0};
    zo_stream* s = zo_stream_new(c->level);
    uint32_t* tok = (uint32_t*)malloc(sizeof(uint32_t) * ZO_SUBBLOCK_SYMS);
    uint8_t* ibuf = (uint8_t*)malloc(ZO_BLOCK_IN + ZO_SENTINEL);
    if (!s || !tok || !ibuf) { free(tok); free(ibuf); if (s) zo_stream_free(s); return ZLNG_E_NOMEM; }
    for (size_t base = 0; base < c->last_len; base += ZO_BLOCK_IN) {
        const int ilen = (int)(c->last_len - base < ZO_BLOCK_IN ? c->last_len - base : ZO_BLOCK_IN);
        memcpy(ibuf, c->last_in + base, (size_t)ilen);
        memset(ibuf + ilen, 0, ZO_SENTINEL);
        zo_reset_buckets(s);
        int encpos = 0, rlen = 0;
        while (encpos < ilen) {
            const int nt = zo_parse_subblock(s, c->level, ibuf, ilen, &encpos, tok, &rlen, 0);
            for (int i = 0; i < nt; i++) { const uint32_t sym = tok[i] & 0xFFFF, aux = tok[i] >> 16; if (sym < 256 && aux < 256) cnt[aux]++; }
        }
    }
    free(tok); free(ibuf); zo_stream_free(s);
    memcpy(dst, cnt, bytes);
    return ZLNG_OK;
}

int zlng_encode_finish_staged(zlng_ctx* c, size_t out_cap, size_t* out_len, size_t* per_block_out_end) {
    if (!c || !c->pending || !out_len) return ZLNG_E_ARG;
    *out_len = 0;
    const size_t bound = zlng_encode_bound(c->pending_len);
    free(c->staged);
    c->staged_len = 0;
    c->staged = (uint8_t*)malloc(bound);
    if (!c->staged) return ZLNG_E_NOMEM;
    const int rc = zlng_encode_finish(c, c->staged, bound < out_cap ? bound : out_cap, &c->staged_len, per_block_out_end);
    if (rc == ZLNG_OK) *out_len = c->staged_len;
    return rc;
}

int zlng_encode_copy_out(zlng_ctx* c, uint8_t* out, size_t n) {
    if (!c || !out || n != c->staged_len) return ZLNG_E_ARG;
    if (n) memcpy(out, c->staged, n);
    c->staged_len = 0;
    return ZLNG_OK;
}

int zlng_encode_blocks(zlng_ctx* c, const uint8_t* in, size_t in_len, uint8_t* out, size_t out_cap, size_t* out_len, size_t* ends) {
    if (!c || !out_len) return ZLNG_E_ARG;
    *out_len = 0;
    if (in_len == 0) return ZLNG_OK;
    const int rc = zlng_encode_parse(c, in, in_len);
    return rc != ZLNG_OK ? rc : zlng_encode_finish(c, out, out_cap, out_len, ends);
}

/* The device-pointer forms.  There is no device here: in the stand-in a "device pointer" is a host address (the CPU tests hand numpy /
 * torch-CPU buffers' addresses where the GPU tests hand tensor.data_ptr()), and the calls are their host-buffer twins.  They exist so
 * that the Python view of the ABI (libzling_amd/__init__.py: Stream, Group) and sharding.RangeEncoder with REAL Stream objects can be
 * driven on the CPU (tests/test_python_binding_on_stub.py); what they say about the kernels is nothing. */
int zlng_stub_marker(void) { return 1; }                /* only the stand-in exports this */
int zlng_encode_parse_device(zlng_ctx* c, const void* d_in, size_t in_len) { return zlng_encode_parse(c, (const uint8_t*)d_in, in_len); }
int zlng_encode_finish_device(zlng_ctx* c, void* d_out, size_t out_cap, size_t* out_len, size_t* ends) {
    if (((uintptr_t)d_out & 3) != 0) return ZLNG_E_ARG; /* zlng_api.hip: the output starts on a 4-byte boundary */
    return zlng_encode_finish(c, (uint8_t*)d_out, out_cap, out_len, ends);
}
int zlng_encode_blocks_device(zlng_ctx* c, const void* d_in, size_t in_len, void* d_out, size_t out_cap, size_t* out_len, size_t* ends) {
    if (!c || !out_len) return ZLNG_E_ARG;
    *out_len = 0;
    if (in_len == 0) return ZLNG_OK;
    const int rc = zlng_encode_parse_device(c, d_in, in_len);
    return rc != ZLNG_OK ? rc : zlng_encode_finish_device(c, d_out, out_cap, out_len, ends);
}
int zlng_encode_parse_after(zlng_ctx* c, zlng_ctx* first) { return (!c || !first || !c->is_encode || !first->is_encode) ? ZLNG_E_ARG : ZLNG_OK; }
/* Stage names as the real library reports them, with the wall time of the stand-in's last call split in made-up shares: enough for the
 * callers' arithmetic (dominant stage, roofline object, schedule model) to run; the numbers mean nothing. */
int zlng_last_timings(zlng_ctx* c, const char** names, float* ms, int cap) {
    static const char* enc[] = {"dict_reset", "rolz_parse", "lit_partition", "mtf_chain", "rank_replay", "histogram", "huff_lengths", "layout_scan", "huff_pack"};
    static const float encw[] = {0.01f, 0.44f, 0.01f, 0.50f, 0.01f, 0.01f, 0.01f, 0.005f, 0.005f};
    static const char* dec[] = {"frame_walk", "huff_decode", "rolz_decode"};
    static const float decw[] = {0.01f, 0.09f, 0.90f};
    if (!c || !names || !ms) return 0;
    const int n = c->is_encode ? 9 : 3;
    int k = 0;
    for (; k < n && k < cap; k++) { names[k] = c->is_encode ? enc[k] : dec[k]; ms[k] = (float)c->last_ms * (c->is_encode ? encw[k] : decw[k]); }
    return k;
}
void* zlng_stream(zlng_ctx* c) { (void)c; return NULL; }

int zlng_set_host_rank_contexts(zlng_ctx* c, int k) { return (!c || !c->is_encode || k < 0) ? ZLNG_E_ARG : ZLNG_OK; }   /* same bytes by definition */

int zlng_get_state(zlng_ctx* c, uint8_t mtf[ZLNG_MTF_STATE], int* current_level) {
    if (!c || !mtf) return ZLNG_E_ARG;
    if (c->is_encode) zo_stream_get_mtf(c->es, mtf); else zo_dstream_get_mtf(c->ds, mtf);
    if (current_level) *current_level = c->is_encode ? zo_stream_get_level(c->es) : 0;
    return ZLNG_OK;
}

int zlng_set_state(zlng_ctx* c, const uint8_t mtf[ZLNG_MTF_STATE], int current_level) {
    if (!c || !mtf || (current_level != 0 && current_level != c->level)) return ZLNG_E_ARG;
    for (int ctx = 0; ctx < 256; ctx++) {                /* every table must be a permutation of 0..255 */
        unsigned char seen[256] = {0};
        for (int i = 0; i < 256; i++) { const uint8_t v = mtf[256 * ctx + i]; if (seen[v]) return ZLNG_E_ARG; seen[v] = 1; }
    }
    if (c->is_encode) { zo_stream_set_mtf(c->es, mtf); zo_stream_set_level(c->es, current_level); }
    else zo_dstream_set_mtf(c->ds, mtf);
    return ZLNG_OK;
}

int zlng_get_state_device(zlng_ctx* c, void* d_mtf, int* current_level) { return zlng_get_state(c, (uint8_t*)d_mtf, current_level); }
int zlng_set_state_device(zlng_ctx* c, const void* d_mtf, int current_level) { return zlng_set_state(c, (const uint8_t*)d_mtf, current_level); }

static int map_err(int zo) {
    switch (zo) {
        case ZO_E_OK: return ZLNG_OK;
        case ZO_E_CAP: return ZLNG_E_CAP;
        case ZO_E_FLAG: return ZLNG_E_FLAG;
        case ZO_E_BLOCKSIZE: return ZLNG_E_BLOCKSIZE;
        case ZO_E_CODE1: return ZLNG_E_CODE1;
        case ZO_E_CODE2: return ZLNG_E_CODE2;
        case ZO_E_EXBITS: return ZLNG_E_EXBITS;
        case ZO_E_LZ: return ZLNG_E_LZ;
        case ZO_E_TRUNC: return ZLNG_E_TRUNC;
        default: return ZLNG_E_DEVICE;
    }
}

int zlng_decode_blocks(zlng_ctx* c, const uint8_t* in, size_t in_len, size_t* in_used, uint8_t* out, size_t out_cap,
                       size_t* out_len, size_t* per_block_out_end) {
    if (!c || c->is_encode || !in_used || !out_len) return ZLNG_E_ARG;
    *in_used = 0;
    *out_len = 0;
    if (in_len == 0) return ZLNG_OK;
    if (!in || !out) return ZLNG_E_ARG;
    static uint8_t entry[ZLNG_MTF_STATE];
    zo_dstream_get_mtf(c->ds, entry);
    uint8_t* blk = (uint8_t*)malloc(ZLNG_BLOCK_SIZE);
    if (!blk) return ZLNG_E_NOMEM;
    size_t ip = 0, op = 0;
    int nblk = 0, rc = ZLNG_OK;
    const double t0 = now_ms();
    while (ip < in_len && nblk < c->max_blocks) {
        size_t n = 0;
        const int zrc = zo_dstream_decode_block(c->ds, in, in_len, &ip, blk, ZLNG_BLOCK_SIZE, &n, NULL);
        if (zrc != ZO_E_OK) {                            /* good blocks in front are reported; the error is met at the head of the next call */
            if (nblk == 0) rc = map_err(zrc);
            break;
        }
        if (op + n > out_cap) {                          /* nothing is reported: put the stream state back */
            zo_dstream_set_mtf(c->ds, entry);
            free(blk);
            return ZLNG_E_CAP;
        }
        memcpy(out + op, blk, n);
        op += n;
        if (per_block_out_end) per_block_out_end[nblk] = op;
        nblk++;
        *in_used = ip;
    }
    free(blk);
    c->last_ms = now_ms() - t0;
    if (rc != ZLNG_OK) return rc;
    *out_len = op;
    return ZLNG_OK;
}

int zlng_decode_blocks_device(zlng_ctx* c, const void* d_in, size_t in_len, size_t* in_used, void* d_out, size_t out_cap,
                              size_t* out_len, size_t* per_block_out_end) {
    return zlng_decode_blocks(c, (const uint8_t*)d_in, in_len, in_used, (uint8_t*)d_out, out_cap, out_len, per_block_out_end);
}

const char* zlng_strerror(int code) {                    /* the product's table (zlng_api.hip); the decode messages are the reference's */
    switch (code) {
        case ZLNG_OK: return "ok";
        case ZLNG_E_ARG: return "invalid argument";
        case ZLNG_E_NOMEM: return "out of memory";
        case ZLNG_E_CAP: return "output capacity too small";
        case ZLNG_E_DEVICE: return "HIP device error or no gfx950 device";
        case ZLNG_E_PAYLOAD: return "sub-block payload exceeds 393216 bytes";
        case ZLNG_E_FLAG: return "baidu::zling::Decode(): invalid encflag.";
        case ZLNG_E_BLOCKSIZE: return "baidu::zling::Decode(): invalid block size.";
        case ZLNG_E_CODE1: return "baidu::zling::Decode(): invalid huffman stream. (bad code1)";
        case ZLNG_E_CODE2: return "baidu::zling::Decode(): invalid huffman stream. (bad code2)";
        case ZLNG_E_EXBITS: return "baidu::zling::Decode(): invalid huffman stream. (bad ex-bits)";
        case ZLNG_E_LZ: return "baidu::zling::Decode(): lzdecode failed.";
        case ZLNG_E_TRUNC: return "truncated stream";
        default: return "unknown error";
    }
}
