// Model of the round-4 rank chain (csrc/mtf_rank.hip, k_mtf_chain) in plain C: the table front of a context in CHAIN LAYOUT in one
// 64-lane register, the four-instruction neighbour step with its test on the shifted mask, the out-of-line head repairs and the
// slow step for literals outside the front -- executed lane by lane exactly as the instructions do it (EXEC lane 0 off, DPP write
// suppression, s_ashr_i64 of the ne-mask) and compared, literal by literal, with ZlingMTFEncoder::Encode
// (src/libzling_lz.cpp:112-117).  Research tool: written BEFORE the kernel; not product, not test.
//   python scripts/experiments/gen_literal_streams.py 4      # ctx.npy / lit.npy from the oracle's parse (in the cwd)
//   gcc -O2 -o /tmp/exp/chain_model scripts/experiments/mtf_chain_model.c && /tmp/exp/chain_model mtfinit.bin ctx.npy lit.npy
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <stdint.h>
static uint8_t nxt[256], mtfinit[256];
typedef struct { uint8_t tab[256], idx[256]; } Mtf;
static inline int enc(Mtf* m, uint8_t c) { int i = m->idx[c], n = nxt[i]; uint8_t d = m->tab[n]; m->tab[n] = c; m->tab[i] = d; m->idx[c] = n; m->idx[d] = i; return i; }

#define NFRONT 60
static int lane_of_pos[NFRONT], pos_of_lane[64];
static void layout(void) {
    for (int l = 0; l < 64; l++) pos_of_lane[l] = -1;
    for (int p = 0; p < NFRONT; p++) {
        int l;
        if (p <= 19) l = p + 1;
        else if (p <= 40 && (p & 1) == 0) l = 21 + (p - 20) / 2;
        else if (p >= 43 && (p - 43) % 3 == 0) l = 32 + (p - 43) / 3;
        else if (p <= 39 && (p & 1)) l = 39 + (p - 21) / 2;
        else if (p >= 42 && (p - 42) % 3 == 0) l = 49 + (p - 42) / 3;
        else l = 56 + (p - 41) / 3;
        lane_of_pos[p] = l; pos_of_lane[l] = p;
    }
    // every front position but the two heads has its swap partner one lane down
    for (int p = 1; p < NFRONT; p++) {
        if (p == 21 || p == 41) continue;
        if (lane_of_pos[nxt[p]] != lane_of_pos[p] - 1) { printf("layout: pos %d\n", p); exit(1); }
    }
    if (nxt[21] != 19 || nxt[41] != 38 || lane_of_pos[21] != 39 || lane_of_pos[41] != 56 || lane_of_pos[19] != 20 || lane_of_pos[38] != 30) { printf("layout heads\n"); exit(1); }
    int used = 0; for (int l = 0; l < 64; l++) used += pos_of_lane[l] >= 0;
    if (used != NFRONT || pos_of_lane[0] >= 0 || pos_of_lane[38] >= 0 || pos_of_lane[55] >= 0 || pos_of_lane[63] >= 0) { printf("layout pads\n"); exit(1); }
}
typedef struct { uint32_t tf[64], tx[64], t1[64], t2[64], t3[64]; } Regs;
static int xpos(int l) { return l == 0 ? 60 : l == 38 ? 61 : l == 55 ? 62 : 63; }
static int xlane(int p) { return p == 60 ? 0 : p == 61 ? 38 : p == 62 ? 55 : 63; }
static uint64_t ALLOWX;
static long n_head, n_back, n_couple;
static void load(Regs* r, const uint8_t* tab) {
    // positions 60..63 sit in the four lanes the front leaves free (0, 38, 55, 63) of a register of their own
    for (int l = 0; l < 64; l++) { r->tf[l] = pos_of_lane[l] >= 0 ? tab[pos_of_lane[l]] : 0x100u | l; r->tx[l] = pos_of_lane[l] >= 0 ? 0x100u | l : tab[xpos(l)]; r->t1[l] = tab[64 + l]; r->t2[l] = tab[128 + l]; r->t3[l] = tab[192 + l]; }
}
static void store(const Regs* r, uint8_t* tab) {
    for (int l = 0; l < 64; l++) { if (pos_of_lane[l] >= 0) tab[pos_of_lane[l]] = r->tf[l]; else tab[xpos(l)] = r->tx[l]; tab[64 + l] = r->t1[l]; tab[128 + l] = r->t2[l]; tab[192 + l] = r->t3[l]; }
}
// one literal; returns the rank
static int step(Regs* r, uint32_t c) {
    const uint64_t exec = ~1ull;
    uint64_t ne = 0;
    for (int l = 0; l < 64; l++) if ((exec >> l & 1) && r->tf[l] != c) ne |= 1ull << l;       // v_cmp_ne (inactive lanes write 0)
    uint32_t old[64]; memcpy(old, r->tf, sizeof old);
    for (int l = 1; l < 64; l++) {                                                            // v_cndmask_b32_dpp wave_shr:1 (lane 1's source is off: suppressed)
        if (!(exec >> l & 1)) continue;
        if (l - 1 < 0 || !(exec >> (l - 1) & 1)) continue;
        r->tf[l] = (ne >> l & 1) ? old[l] : old[l - 1];
    }
    const uint64_t x = (uint64_t)((int64_t)ne >> 1);                                          // s_ashr_i64
    for (int l = 0; l < 64; l++) if ((exec >> l & 1) && !(x >> l & 1)) r->tf[l] = c;          // v_cndmask_b32_sdwa
    if (ALLOWX & ~x) {                                                                        // s_andn2: hit in an allowed lane
        int hl = __builtin_ctzll(~x) + 1;
        return pos_of_lane[hl];
    }
    // out of line
    uint64_t eq = 0;
    for (int l = 1; l < 64; l++) if (r->tf[l] == c) eq |= 1ull << l;
    if (eq) {
        n_head++;
        if (eq == 1ull << 38) { uint32_t da = r->tf[20]; r->tf[39] = da; r->tf[20] = c; r->tf[38] = 0x100u | 38; return 21; }
        if (eq == 1ull << 55) { uint32_t da = r->tf[30]; r->tf[56] = da; r->tf[30] = c; r->tf[55] = 0x100u | 55; return 41; }
        printf("unexpected eq mask %016llx\n", (unsigned long long)eq); exit(1);
    }
    // slow step: c is at a position >= 60
    n_back++;
    int i = -1;
    for (int l = 0; l < 64; l++) { if (r->tx[l] == c) i = xpos(l); if (r->t1[l] == c) i = 64 + l; if (r->t2[l] == c) i = 128 + l; if (r->t3[l] == c) i = 192 + l; }
    if (i < 0) { printf("lost symbol %u\n", c); exit(1); }
    const int n = nxt[i];
    uint32_t* ri = i < 64 ? r->tx : i < 128 ? r->t1 : i < 192 ? r->t2 : r->t3;
    const int li = i < 64 ? xlane(i) : i & 63;
    uint32_t* rn; int ln;
    if (n < NFRONT) { rn = r->tf; ln = lane_of_pos[n]; n_couple++; }
    else if (n < 64) { rn = r->tx; ln = xlane(n); }
    else { rn = n < 128 ? r->t1 : n < 192 ? r->t2 : r->t3; ln = n & 63; }
    const uint32_t d = rn[ln];
    ri[li] = d; rn[ln] = c;
    return i;
}
int main(int argc, char** argv) {
    for (int i = 0; i < 256; i++) nxt[i] = i < 128 ? i * 95 / 100 : i * 55 / 100;
    layout();
    for (int l = 1; l < 64; l++) if (pos_of_lane[l] >= 0 && pos_of_lane[l] != 21 && pos_of_lane[l] != 41) ALLOWX |= 1ull << (l - 1);
    printf("ALLOWX = 0x%016llx\n", (unsigned long long)ALLOWX);
    FILE* f = fopen(argv[1], "rb"); if (fread(mtfinit, 1, 256, f) != 256) return 1; fclose(f);
    f = fopen(argv[2], "rb"); fseek(f, 0, SEEK_END); long n = ftell(f) - 128; fseek(f, 128, SEEK_SET); uint8_t* ctx = malloc(n); if (fread(ctx, 1, n, f) != (size_t)n) return 1; fclose(f);
    f = fopen(argv[3], "rb"); fseek(f, 128, SEEK_SET); uint8_t* lit = malloc(n); if (fread(lit, 1, n, f) != (size_t)n) return 1; fclose(f);
    static Mtf T[256]; static Regs R[256];
    for (int c = 0; c < 256; c++) { for (int i = 0; i < 256; i++) { T[c].tab[i] = mtfinit[i]; T[c].idx[mtfinit[i]] = i; } load(&R[c], T[c].tab); }
    long bad = 0;
    for (long j = 0; j < n; j++) {
        const int c = ctx[j];
        const int want = enc(&T[c], lit[j]), got = step(&R[c], lit[j]);
        if (want != got && bad++ < 10) printf("literal %ld ctx %d byte %d: rank %d, model %d\n", j, c, lit[j], want, got);
    }
    for (int c = 0; c < 256; c++) { uint8_t tab[256]; store(&R[c], tab); if (memcmp(tab, T[c].tab, 256)) { bad++; printf("table of ctx %d differs\n", c); } }
    printf("%ld literals: %ld mismatches; head repairs %ld (%.2f%%), slow steps %ld (%.2f%%), of which couplings %ld (%.2f%%)\n", n, bad, n_head, 100.0 * n_head / n,
           n_back, 100.0 * n_back / n, n_couple, 100.0 * n_couple / n);
    return bad != 0;
}
