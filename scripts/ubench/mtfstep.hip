// Variants of the k_mtf_dense fast step, timed on one lone wavefront (test tool, not product).
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#define ITER 200000
#define COMMON_OUT [t0] "+v"(t0), [ranks] "+v"(ranks), [m0] "=&s"(m0), [m1] "=&s"(m1), [i] "=&s"(idx), [c] "=&s"(c), [cv] "=&v"(cv), [up] "+v"(up)
#define DECL unsigned t0 = threadIdx.x, v = (threadIdx.x * 7 + 3) & 15, ranks = 0, idx, c, cv, up = 0xffffffffu; unsigned long long m0, m1;
#define FIN out[threadIdx.x] = t0 * 31 + ranks;

// P0: the shipped order
#define P0(K) asm volatile("v_readlane_b32 %[c], %[v], " #K "\n\tv_mov_b32 %[cv], %[c]\n\tv_cmp_ne_u32_e32 vcc, %[cv], %[t0]\n\t" \
    "s_not_b64 %[m0], vcc\n\ts_lshr_b64 %[m1], %[m0], 1\n\t" \
    "v_cndmask_b32_dpp %[t0], %[t0], %[t0], vcc wave_shr:1 row_mask:0xf bank_mask:0xf\n\t" \
    "v_cndmask_b32_e64 %[t0], %[t0], %[cv], %[m1]\n\ts_ff1_i32_b64 %[i], %[m0]\n\t" \
    "s_cmp_lt_u32 %[i], 21\n\ts_cbranch_scc0 1f\n\t1: v_writelane_b32 %[ranks], %[i], " #K "\n\t" \
    : COMMON_OUT : [v] "v"(v) : "vcc", "scc")
__global__ void k_p0(unsigned* out) { DECL for (int it = 0; it < ITER; it++) { P0(0); P0(1); P0(2); P0(3); } FIN }

// P0 without the compare+branch
#define P0NB(K) asm volatile("v_readlane_b32 %[c], %[v], " #K "\n\tv_mov_b32 %[cv], %[c]\n\tv_cmp_ne_u32_e32 vcc, %[cv], %[t0]\n\t" \
    "s_not_b64 %[m0], vcc\n\ts_lshr_b64 %[m1], %[m0], 1\n\t" \
    "v_cndmask_b32_dpp %[t0], %[t0], %[t0], vcc wave_shr:1 row_mask:0xf bank_mask:0xf\n\t" \
    "v_cndmask_b32_e64 %[t0], %[t0], %[cv], %[m1]\n\ts_ff1_i32_b64 %[i], %[m0]\n\t" \
    "v_writelane_b32 %[ranks], %[i], " #K "\n\t" \
    : COMMON_OUT : [v] "v"(v) : "vcc", "scc")
__global__ void k_p0nb(unsigned* out) { DECL for (int it = 0; it < ITER; it++) { P0NB(0); P0NB(1); P0NB(2); P0NB(3); } FIN }

// P0 without the rank store
#define P0NS(K) asm volatile("v_readlane_b32 %[c], %[v], " #K "\n\tv_mov_b32 %[cv], %[c]\n\tv_cmp_ne_u32_e32 vcc, %[cv], %[t0]\n\t" \
    "s_not_b64 %[m0], vcc\n\ts_lshr_b64 %[m1], %[m0], 1\n\t" \
    "v_cndmask_b32_dpp %[t0], %[t0], %[t0], vcc wave_shr:1 row_mask:0xf bank_mask:0xf\n\t" \
    "v_cndmask_b32_e64 %[t0], %[t0], %[cv], %[m1]\n\ts_ff1_i32_b64 %[i], %[m0]\n\t" \
    "s_cmp_lt_u32 %[i], 21\n\ts_cbranch_scc0 1f\n\t1:\n\t" \
    : COMMON_OUT : [v] "v"(v) : "vcc", "scc")
__global__ void k_p0ns(unsigned* out) { DECL for (int it = 0; it < ITER; it++) { P0NS(0); P0NS(1); P0NS(2); P0NS(3); } FIN }

// P2: all-VALU table chain: up = t0[l+1]; m1 = (up == c); rank from ~vcc off the chain
#define P2(K) asm volatile("v_readlane_b32 %[c], %[v], " #K "\n\t" \
    "v_mov_b32_dpp %[up], %[t0] wave_shl:1 row_mask:0xf bank_mask:0xf\n\t" \
    "v_cmp_ne_u32_e32 vcc, %[c], %[t0]\n\t" \
    "v_cmp_eq_u32_e64 %[m1], %[c], %[up]\n\t" \
    "v_cndmask_b32_dpp %[t0], %[t0], %[t0], vcc wave_shr:1 row_mask:0xf bank_mask:0xf\n\t" \
    "v_cndmask_b32_e64 %[t0], %[t0], %[up], %[m1]\n\t" \
    "s_not_b64 %[m0], vcc\n\ts_ff1_i32_b64 %[i], %[m0]\n\t" \
    "s_cmp_lt_u32 %[i], 21\n\ts_cbranch_scc0 1f\n\t1: v_writelane_b32 %[ranks], %[i], " #K "\n\t" \
    : COMMON_OUT : [v] "v"(v) : "vcc", "scc")
__global__ void k_p2(unsigned* out) { DECL for (int it = 0; it < ITER; it++) { P2(0); P2(1); P2(2); P2(3); } FIN }

// P3: P2 with the branch taken from one s_andn2 on vcc (SCC = any hit in lanes 0..20), rank from that mask
#define P3(K) asm volatile("v_readlane_b32 %[c], %[v], " #K "\n\t" \
    "v_mov_b32_dpp %[up], %[t0] wave_shl:1 row_mask:0xf bank_mask:0xf\n\t" \
    "v_cmp_ne_u32_e32 vcc, %[c], %[t0]\n\t" \
    "v_cmp_eq_u32_e64 %[m1], %[c], %[up]\n\t" \
    "v_cndmask_b32_dpp %[t0], %[t0], %[t0], vcc wave_shr:1 row_mask:0xf bank_mask:0xf\n\t" \
    "v_cndmask_b32_e64 %[t0], %[t0], %[up], %[m1]\n\t" \
    "s_andn2_b64 %[m0], 0x1fffff, vcc\n\ts_cbranch_scc0 1f\n\t1: s_ff1_i32_b64 %[i], %[m0]\n\t" \
    "v_writelane_b32 %[ranks], %[i], " #K "\n\t" \
    : COMMON_OUT : [v] "v"(v) : "vcc", "scc")
__global__ void k_p3(unsigned* out) { DECL for (int it = 0; it < ITER; it++) { P3(0); P3(1); P3(2); P3(3); } FIN }

// P4: P3 with the four literal fetches hoisted in front of the four steps (c0..c3)
#define P4BODY(C, K) \
    "v_mov_b32_dpp %[up], %[t0] wave_shl:1 row_mask:0xf bank_mask:0xf\n\t" \
    "v_cmp_ne_u32_e32 vcc, %[" #C "], %[t0]\n\t" \
    "v_cmp_eq_u32_e64 %[m1], %[" #C "], %[up]\n\t" \
    "v_cndmask_b32_dpp %[t0], %[t0], %[t0], vcc wave_shr:1 row_mask:0xf bank_mask:0xf\n\t" \
    "v_cndmask_b32_e64 %[t0], %[t0], %[up], %[m1]\n\t" \
    "s_andn2_b64 %[m0], 0x1fffff, vcc\n\ts_cbranch_scc0 1f\n\t1: s_ff1_i32_b64 %[i], %[m0]\n\t" \
    "v_writelane_b32 %[ranks], %[i], " #K "\n\t"
__global__ void k_p4(unsigned* out) {
    DECL unsigned c1, c2, c3;
    for (int it = 0; it < ITER; it++) {
        asm volatile("v_readlane_b32 %[c], %[v], 0\n\tv_readlane_b32 %[c1], %[v], 1\n\tv_readlane_b32 %[c2], %[v], 2\n\tv_readlane_b32 %[c3], %[v], 3\n\t"
                     P4BODY(c, 0) P4BODY(c1, 1) P4BODY(c2, 2) P4BODY(c3, 3)
                     : COMMON_OUT, [c1] "=&s"(c1), [c2] "=&s"(c2), [c3] "=&s"(c3) : [v] "v"(v) : "vcc", "scc");
    }
    FIN
}
// P5: P4 with the rank store deferred: ranks packed four to an SGPR (s_lshl_b32 + s_or_b32), none per step
#define P5BODY(C) \
    "v_mov_b32_dpp %[up], %[t0] wave_shl:1 row_mask:0xf bank_mask:0xf\n\t" \
    "v_cmp_ne_u32_e32 vcc, %[" #C "], %[t0]\n\t" \
    "v_cmp_eq_u32_e64 %[m1], %[" #C "], %[up]\n\t" \
    "v_cndmask_b32_dpp %[t0], %[t0], %[t0], vcc wave_shr:1 row_mask:0xf bank_mask:0xf\n\t" \
    "v_cndmask_b32_e64 %[t0], %[t0], %[up], %[m1]\n\t" \
    "s_andn2_b64 %[m0], 0x1fffff, vcc\n\ts_cbranch_scc0 1f\n\t1: s_ff1_i32_b64 %[i], %[m0]\n\t" \
    "s_lshl_b32 %[pk], %[pk], 8\n\ts_or_b32 %[pk], %[pk], %[i]\n\t"
__global__ void k_p5(unsigned* out) {
    DECL unsigned c1, c2, c3, pk = 0;
    for (int it = 0; it < ITER; it++) {
        asm volatile("v_readlane_b32 %[c], %[v], 0\n\tv_readlane_b32 %[c1], %[v], 1\n\tv_readlane_b32 %[c2], %[v], 2\n\tv_readlane_b32 %[c3], %[v], 3\n\t"
                     P5BODY(c) P5BODY(c1) P5BODY(c2) P5BODY(c3) "v_writelane_b32 %[ranks], %[pk], 0\n\t"
                     : COMMON_OUT, [c1] "=&s"(c1), [c2] "=&s"(c2), [c3] "=&s"(c3), [pk] "+s"(pk) : [v] "v"(v) : "vcc", "scc");
    }
    FIN
}
// P6: only the table chain (4 VALU + dpp), nothing else: the floor for this formulation
#define P6BODY(C) \
    "v_mov_b32_dpp %[up], %[t0] wave_shl:1 row_mask:0xf bank_mask:0xf\n\t" \
    "v_cmp_ne_u32_e32 vcc, %[" #C "], %[t0]\n\t" \
    "v_cmp_eq_u32_e64 %[m1], %[" #C "], %[up]\n\t" \
    "v_cndmask_b32_dpp %[t0], %[t0], %[t0], vcc wave_shr:1 row_mask:0xf bank_mask:0xf\n\t" \
    "v_cndmask_b32_e64 %[t0], %[t0], %[up], %[m1]\n\t"
__global__ void k_p6(unsigned* out) {
    DECL unsigned c1, c2, c3;
    for (int it = 0; it < ITER; it++) {
        asm volatile("v_readlane_b32 %[c], %[v], 0\n\tv_readlane_b32 %[c1], %[v], 1\n\tv_readlane_b32 %[c2], %[v], 2\n\tv_readlane_b32 %[c3], %[v], 3\n\t"
                     P6BODY(c) P6BODY(c1) P6BODY(c2) P6BODY(c3)
                     : COMMON_OUT, [c1] "=&s"(c1), [c2] "=&s"(c2), [c3] "=&s"(c3) : [v] "v"(v) : "vcc", "scc");
    }
    FIN
}


// P4NB: P4 without the branch (bound for any cleverer slow-path detection)
#define P4NBBODY(C, K) \
    "v_mov_b32_dpp %[up], %[t0] wave_shl:1 row_mask:0xf bank_mask:0xf\n\t" \
    "v_cmp_ne_u32_e32 vcc, %[" #C "], %[t0]\n\t" \
    "v_cmp_eq_u32_e64 %[m1], %[" #C "], %[up]\n\t" \
    "v_cndmask_b32_dpp %[t0], %[t0], %[t0], vcc wave_shr:1 row_mask:0xf bank_mask:0xf\n\t" \
    "v_cndmask_b32_e64 %[t0], %[t0], %[up], %[m1]\n\t" \
    "s_andn2_b64 %[m0], 0x1fffff, vcc\n\ts_ff1_i32_b64 %[i], %[m0]\n\t" \
    "v_writelane_b32 %[ranks], %[i], " #K "\n\t"
__global__ void k_p4nb(unsigned* out) {
    DECL unsigned c1, c2, c3;
    for (int it = 0; it < ITER; it++) {
        asm volatile("v_readlane_b32 %[c], %[v], 0\n\tv_readlane_b32 %[c1], %[v], 1\n\tv_readlane_b32 %[c2], %[v], 2\n\tv_readlane_b32 %[c3], %[v], 3\n\t"
                     P4NBBODY(c, 0) P4NBBODY(c1, 1) P4NBBODY(c2, 2) P4NBBODY(c3, 3)
                     : COMMON_OUT, [c1] "=&s"(c1), [c2] "=&s"(c2), [c3] "=&s"(c3) : [v] "v"(v) : "vcc", "scc");
    }
    FIN
}
// P7: scalar side work interleaved into the VALU chain
#define P7BODY(C, K) \
    "v_mov_b32_dpp %[up], %[t0] wave_shl:1 row_mask:0xf bank_mask:0xf\n\t" \
    "v_cmp_ne_u32_e32 vcc, %[" #C "], %[t0]\n\t" \
    "v_cmp_eq_u32_e64 %[m1], %[" #C "], %[up]\n\t" \
    "s_andn2_b64 %[m0], 0x1fffff, vcc\n\t" \
    "v_cndmask_b32_dpp %[t0], %[t0], %[t0], vcc wave_shr:1 row_mask:0xf bank_mask:0xf\n\t" \
    "s_ff1_i32_b64 %[i], %[m0]\n\t" \
    "v_cndmask_b32_e64 %[t0], %[t0], %[up], %[m1]\n\t" \
    "s_cbranch_scc0 1f\n\t" \
    "1: v_writelane_b32 %[ranks], %[i], " #K "\n\t"
__global__ void k_p7(unsigned* out) {
    DECL unsigned c1, c2, c3;
    for (int it = 0; it < ITER; it++) {
        asm volatile("v_readlane_b32 %[c], %[v], 0\n\tv_readlane_b32 %[c1], %[v], 1\n\tv_readlane_b32 %[c2], %[v], 2\n\tv_readlane_b32 %[c3], %[v], 3\n\t"
                     P7BODY(c, 0) P7BODY(c1, 1) P7BODY(c2, 2) P7BODY(c3, 3)
                     : COMMON_OUT, [c1] "=&s"(c1), [c2] "=&s"(c2), [c3] "=&s"(c3) : [v] "v"(v) : "vcc", "scc");
    }
    FIN
}
// P8: P7 with the rank store of step k-1 moved into step k (i alternates between two SGPRs)
#define P8BODY(C, IP, IN, KPREV) \
    "v_mov_b32_dpp %[up], %[t0] wave_shl:1 row_mask:0xf bank_mask:0xf\n\t" \
    "v_cmp_ne_u32_e32 vcc, %[" #C "], %[t0]\n\t" \
    "v_writelane_b32 %[ranks], %[" #IP "], " #KPREV "\n\t" \
    "v_cmp_eq_u32_e64 %[m1], %[" #C "], %[up]\n\t" \
    "s_andn2_b64 %[m0], 0x1fffff, vcc\n\t" \
    "v_cndmask_b32_dpp %[t0], %[t0], %[t0], vcc wave_shr:1 row_mask:0xf bank_mask:0xf\n\t" \
    "s_ff1_i32_b64 %[" #IN "], %[m0]\n\t" \
    "v_cndmask_b32_e64 %[t0], %[t0], %[up], %[m1]\n\t" \
    "s_cbranch_scc0 1f\n\t1:\n\t"
__global__ void k_p8(unsigned* out) {
    DECL unsigned c1, c2, c3, i2 = 0; idx = 0;
    for (int it = 0; it < ITER; it++) {
        asm volatile("v_readlane_b32 %[c], %[v], 0\n\tv_readlane_b32 %[c1], %[v], 1\n\tv_readlane_b32 %[c2], %[v], 2\n\tv_readlane_b32 %[c3], %[v], 3\n\t"
                     P8BODY(c, i2, i, 3) P8BODY(c1, i, i2, 0) P8BODY(c2, i2, i, 1) P8BODY(c3, i, i2, 2)
                     : [t0] "+v"(t0), [ranks] "+v"(ranks), [m0] "=&s"(m0), [m1] "=&s"(m1), [i] "+s"(idx), [i2] "+s"(i2), [c] "=&s"(c), [cv] "=&v"(cv), [up] "+v"(up),
                       [c1] "=&s"(c1), [c2] "=&s"(c2), [c3] "=&s"(c3) : [v] "v"(v) : "vcc", "scc");
    }
    FIN
}
// P9: P8 with the branch decided on the previous step's mask (one step late: s_cbranch far from its s_andn2)
//     -- only a timing probe for "how much does distance to the branch matter"
#define P9BODY(C, IP, IN, KPREV) \
    "v_mov_b32_dpp %[up], %[t0] wave_shl:1 row_mask:0xf bank_mask:0xf\n\t" \
    "v_cmp_ne_u32_e32 vcc, %[" #C "], %[t0]\n\t" \
    "v_writelane_b32 %[ranks], %[" #IP "], " #KPREV "\n\t" \
    "v_cmp_eq_u32_e64 %[m1], %[" #C "], %[up]\n\t" \
    "s_cbranch_scc0 1f\n\t1:\n\t" \
    "s_andn2_b64 %[m0], 0x1fffff, vcc\n\t" \
    "v_cndmask_b32_dpp %[t0], %[t0], %[t0], vcc wave_shr:1 row_mask:0xf bank_mask:0xf\n\t" \
    "s_ff1_i32_b64 %[" #IN "], %[m0]\n\t" \
    "v_cndmask_b32_e64 %[t0], %[t0], %[up], %[m1]\n\t"
__global__ void k_p9(unsigned* out) {
    DECL unsigned c1, c2, c3, i2 = 0; idx = 0;
    for (int it = 0; it < ITER; it++) {
        asm volatile("s_cmp_eq_u32 0, 0\n\tv_readlane_b32 %[c], %[v], 0\n\tv_readlane_b32 %[c1], %[v], 1\n\tv_readlane_b32 %[c2], %[v], 2\n\tv_readlane_b32 %[c3], %[v], 3\n\t"
                     P9BODY(c, i2, i, 3) P9BODY(c1, i, i2, 0) P9BODY(c2, i2, i, 1) P9BODY(c3, i, i2, 2)
                     : [t0] "+v"(t0), [ranks] "+v"(ranks), [m0] "=&s"(m0), [m1] "=&s"(m1), [i] "+s"(idx), [i2] "+s"(i2), [c] "=&s"(c), [cv] "=&v"(cv), [up] "+v"(up),
                       [c1] "=&s"(c1), [c2] "=&s"(c2), [c3] "=&s"(c3) : [v] "v"(v) : "vcc", "scc");
    }
    FIN
}


// P10: P4 with the rank recorded as the one-hot fast-lane mask (v_writelane of hm_lo), decoded per tile later
#define P10BODY(C, K) \
    "v_mov_b32_dpp %[up], %[t0] wave_shl:1 row_mask:0xf bank_mask:0xf\n\t" \
    "v_cmp_ne_u32_e32 vcc, %[" #C "], %[t0]\n\t" \
    "v_cmp_eq_u32_e64 %[m1], %[" #C "], %[up]\n\t" \
    "v_cndmask_b32_dpp %[t0], %[t0], %[t0], vcc wave_shr:1 row_mask:0xf bank_mask:0xf\n\t" \
    "v_cndmask_b32_e64 %[t0], %[t0], %[up], %[m1]\n\t" \
    "s_andn2_b64 s[90:91], 0x1fffff, vcc\n\ts_cbranch_scc0 1f\n\t1: " \
    "v_writelane_b32 %[ranks], s90, " #K "\n\t"
__global__ void k_p10(unsigned* out) {
    DECL unsigned c1, c2, c3;
    for (int it = 0; it < ITER; it++) {
        asm volatile("v_readlane_b32 %[c], %[v], 0\n\tv_readlane_b32 %[c1], %[v], 1\n\tv_readlane_b32 %[c2], %[v], 2\n\tv_readlane_b32 %[c3], %[v], 3\n\t"
                     P10BODY(c, 0) P10BODY(c1, 1) P10BODY(c2, 2) P10BODY(c3, 3)
                     : [t0] "+v"(t0), [ranks] "+v"(ranks), [m0] "=&s"(m0), [m1] "=&s"(m1), [c] "=&s"(c), [cv] "=&v"(cv), [up] "+v"(up),
                       [c1] "=&s"(c1), [c2] "=&s"(c2), [c3] "=&s"(c3) : [v] "v"(v) : "vcc", "scc", "s90", "s91");
    }
    FIN
}
// P11: P10 with the branch of step k-1 taken inside step k, after its compares (SCC carried across the VALU ops)
#define P11BODY(C, K) \
    "v_mov_b32_dpp %[up], %[t0] wave_shl:1 row_mask:0xf bank_mask:0xf\n\t" \
    "v_cmp_ne_u32_e32 vcc, %[" #C "], %[t0]\n\t" \
    "v_cmp_eq_u32_e64 %[m1], %[" #C "], %[up]\n\t" \
    "s_cbranch_scc0 1f\n\t1: " \
    "v_cndmask_b32_dpp %[t0], %[t0], %[t0], vcc wave_shr:1 row_mask:0xf bank_mask:0xf\n\t" \
    "v_cndmask_b32_e64 %[t0], %[t0], %[up], %[m1]\n\t" \
    "s_andn2_b64 s[90:91], 0x1fffff, vcc\n\t" \
    "v_writelane_b32 %[ranks], s90, " #K "\n\t"
__global__ void k_p11(unsigned* out) {
    DECL unsigned c1, c2, c3;
    for (int it = 0; it < ITER; it++) {
        asm volatile("s_cmp_eq_u32 0, 0\n\tv_readlane_b32 %[c], %[v], 0\n\tv_readlane_b32 %[c1], %[v], 1\n\tv_readlane_b32 %[c2], %[v], 2\n\tv_readlane_b32 %[c3], %[v], 3\n\t"
                     P11BODY(c, 0) P11BODY(c1, 1) P11BODY(c2, 2) P11BODY(c3, 3)
                     : [t0] "+v"(t0), [ranks] "+v"(ranks), [m0] "=&s"(m0), [m1] "=&s"(m1), [c] "=&s"(c), [cv] "=&v"(cv), [up] "+v"(up),
                       [c1] "=&s"(c1), [c2] "=&s"(c2), [c3] "=&s"(c3) : [v] "v"(v) : "vcc", "scc", "s90", "s91");
    }
    FIN
}


// P12: P11 with the literal taken as a byte of a packed SGPR (SDWA scalar source): no per-literal fetch at all
#define P12BODY(B, K) \
    "v_mov_b32_dpp %[up], %[t0] wave_shl:1 row_mask:0xf bank_mask:0xf\n\t" \
    "v_cmp_ne_u32_sdwa vcc, %[pk], %[t0] src0_sel:BYTE_" #B " src1_sel:DWORD\n\t" \
    "v_cmp_eq_u32_sdwa %[m1], %[pk], %[up] src0_sel:BYTE_" #B " src1_sel:DWORD\n\t" \
    "s_cbranch_scc0 1f\n\t1: " \
    "v_cndmask_b32_dpp %[t0], %[t0], %[t0], vcc wave_shr:1 row_mask:0xf bank_mask:0xf\n\t" \
    "v_cndmask_b32_e64 %[t0], %[t0], %[up], %[m1]\n\t" \
    "s_andn2_b64 s[90:91], 0x1fffff, vcc\n\t" \
    "v_writelane_b32 %[ranks], s90, " #K "\n\t"
__global__ void k_p12(unsigned* out) {
    DECL unsigned pk = 0x0d060a03u;
    for (int it = 0; it < ITER; it++) {
        asm volatile("s_cmp_eq_u32 0, 0\n\t"
                     P12BODY(0, 0) P12BODY(1, 1) P12BODY(2, 2) P12BODY(3, 3)
                     : [t0] "+v"(t0), [ranks] "+v"(ranks), [m1] "=&s"(m1), [up] "+v"(up) : [pk] "s"(pk) : "vcc", "scc", "s90", "s91");
    }
    FIN
}

// P13: P12 without the rank record (state-only chain; ranks recomputed per tile in parallel from table snapshots)
#define P13BODY(B, K) \
    "v_cmp_ne_u32_sdwa vcc, %[pk], %[t0] src0_sel:BYTE_" #B " src1_sel:DWORD\n\t" \
    "v_mov_b32_dpp %[up], %[t0] wave_shl:1 row_mask:0xf bank_mask:0xf\n\t" \
    "v_cmp_eq_u32_sdwa %[m1], %[pk], %[up] src0_sel:BYTE_" #B " src1_sel:DWORD\n\t" \
    "s_cbranch_scc0 1f\n\t1: " \
    "v_cndmask_b32_dpp %[t0], %[t0], %[t0], vcc wave_shr:1 row_mask:0xf bank_mask:0xf\n\t" \
    "v_cndmask_b32_e64 %[t0], %[t0], %[up], %[m1]\n\t" \
    "s_andn2_b64 s[90:91], 0x1fffff, vcc\n\t"
__global__ void k_p13(unsigned* out) {
    DECL unsigned pk = 0x0d060a03u;
    for (int it = 0; it < ITER; it++) {
        asm volatile("s_cmp_eq_u32 0, 0\n\t"
                     P13BODY(0, 0) P13BODY(1, 1) P13BODY(2, 2) P13BODY(3, 3)
                     : [t0] "+v"(t0), [m1] "=&s"(m1), [up] "+v"(up) : [pk] "s"(pk) : "vcc", "scc", "s90", "s91");
    }
    FIN
}
// P14: P13 with the slow-path test on the vcc of a compare restricted to lanes 0..20 by EXEC (s_cbranch_vccz, no s_andn2):
//      the restricted compare is an extra v_cmp_eq under a narrowed exec -- does swapping SALU for VALU + exec writes pay?  (6 + 2 exec writes)
#define P14BODY(B, K) \
    "v_mov_b32_dpp %[up], %[t0] wave_shl:1 row_mask:0xf bank_mask:0xf\n\t" \
    "v_cmp_ne_u32_sdwa vcc, %[pk], %[t0] src0_sel:BYTE_" #B " src1_sel:DWORD\n\t" \
    "v_cmp_eq_u32_sdwa %[m1], %[pk], %[up] src0_sel:BYTE_" #B " src1_sel:DWORD\n\t" \
    "v_cndmask_b32_dpp %[t0], %[t0], %[t0], vcc wave_shr:1 row_mask:0xf bank_mask:0xf\n\t" \
    "v_cndmask_b32_e64 %[t0], %[t0], %[up], %[m1]\n\t" \
    "s_andn2_b64 s[90:91], 0x1fffff, vcc\n\t" \
    "s_cbranch_scc0 1f\n\t1: "
__global__ void k_p14(unsigned* out) {
    DECL unsigned pk = 0x0d060a03u;
    for (int it = 0; it < ITER; it++) {
        asm volatile(P14BODY(0, 0) P14BODY(1, 1) P14BODY(2, 2) P14BODY(3, 3)
                     : [t0] "+v"(t0), [m1] "=&s"(m1), [up] "+v"(up) : [pk] "s"(pk) : "vcc", "scc", "s90", "s91");
    }
    FIN
}
// P15: P13 but one slow-path test per TWO literals: both one-hot words must be non-zero (s_min_u32 of the two, then s_cmp)
#define P15PAIR(B0, B1) \
    "v_mov_b32_dpp %[up], %[t0] wave_shl:1 row_mask:0xf bank_mask:0xf\n\t" \
    "v_cmp_ne_u32_sdwa vcc, %[pk], %[t0] src0_sel:BYTE_" #B0 " src1_sel:DWORD\n\t" \
    "v_cmp_eq_u32_sdwa %[m1], %[pk], %[up] src0_sel:BYTE_" #B0 " src1_sel:DWORD\n\t" \
    "s_cbranch_scc0 1f\n\t1: " \
    "v_cndmask_b32_dpp %[t0], %[t0], %[t0], vcc wave_shr:1 row_mask:0xf bank_mask:0xf\n\t" \
    "v_cndmask_b32_e64 %[t0], %[t0], %[up], %[m1]\n\t" \
    "s_andn2_b64 s[90:91], 0x1fffff, vcc\n\t" \
    "v_mov_b32_dpp %[up], %[t0] wave_shl:1 row_mask:0xf bank_mask:0xf\n\t" \
    "v_cmp_ne_u32_sdwa vcc, %[pk], %[t0] src0_sel:BYTE_" #B1 " src1_sel:DWORD\n\t" \
    "v_cmp_eq_u32_sdwa %[m1], %[pk], %[up] src0_sel:BYTE_" #B1 " src1_sel:DWORD\n\t" \
    "v_cndmask_b32_dpp %[t0], %[t0], %[t0], vcc wave_shr:1 row_mask:0xf bank_mask:0xf\n\t" \
    "v_cndmask_b32_e64 %[t0], %[t0], %[up], %[m1]\n\t" \
    "s_andn2_b64 s[92:93], 0x1fffff, vcc\n\t" \
    "s_min_u32 s90, s90, s92\n\t" \
    "s_cmp_lg_u32 s90, 0\n\t"
__global__ void k_p15(unsigned* out) {
    DECL unsigned pk = 0x0d060a03u;
    for (int it = 0; it < ITER; it++) {
        asm volatile("s_cmp_eq_u32 0, 0\n\t"
                     P15PAIR(0, 1) P15PAIR(2, 3)
                     : [t0] "+v"(t0), [m1] "=&s"(m1), [up] "+v"(up) : [pk] "s"(pk) : "vcc", "scc", "s90", "s91", "s92", "s93");
    }
    FIN
}

// P16: state-only chain whose slow-path test stays on the vector unit: after the speculative neighbour swap c sits at lane
//      rank-1 (or 0), so "c is within lanes 0..19 of the NEW table" <=> rank <= 20.  tl = table with lanes >= 20 blanked
//      (one v_cndmask with a constant lane mask), v_cmp_eq -> VCC, s_cbranch_vccz taken two instructions late.  No SALU ALU op.
#define P16BODY(B, K) \
    "v_mov_b32_dpp %[up], %[t0] wave_shl:1 row_mask:0xf bank_mask:0xf\n\t" \
    "v_cmp_eq_u32_sdwa %[m1], %[pk], %[up] src0_sel:BYTE_" #B " src1_sel:DWORD\n\t" \
    "s_cbranch_vccz 1f\n\t1: " \
    "v_cmp_ne_u32_sdwa vcc, %[pk], %[t0] src0_sel:BYTE_" #B " src1_sel:DWORD\n\t" \
    "v_cndmask_b32_dpp %[t0], %[t0], %[t0], vcc wave_shr:1 row_mask:0xf bank_mask:0xf\n\t" \
    "v_cndmask_b32_e64 %[t0], %[t0], %[up], %[m1]\n\t" \
    "v_cndmask_b32_e64 %[tl], %[t0], %[ff], %[hi]\n\t" \
    "v_cmp_eq_u32_sdwa vcc, %[pk], %[tl] src0_sel:BYTE_" #B " src1_sel:DWORD\n\t"
__global__ void k_p16(unsigned* out) {
    DECL unsigned pk = 0x0d060a03u, tl, ff = 0xffffffffu; unsigned long long hi = ~0xfffffull;
    for (int it = 0; it < ITER; it++) {
        asm volatile("v_cmp_eq_u32_e32 vcc, %[t0], %[t0]\n\t"
                     P16BODY(0, 0) P16BODY(1, 1) P16BODY(2, 2) P16BODY(3, 3)
                     : [t0] "+v"(t0), [m1] "=&s"(m1), [up] "+v"(up), [tl] "=&v"(tl) : [pk] "s"(pk), [ff] "v"(ff), [hi] "s"(hi) : "vcc", "scc");
    }
    FIN
}
template <class F> static double run(F launch) {
    hipEvent_t a, b; (void)hipEventCreate(&a); (void)hipEventCreate(&b);
    float best = 1e30f;
    for (int r = 0; r < 3; r++) {
        (void)hipEventRecord(a); launch(); (void)hipEventRecord(b); (void)hipEventSynchronize(b);
        float ms; (void)hipEventElapsedTime(&ms, a, b); if (ms < best) best = ms;
    }
    return best;
}
int main(int argc, char** argv) {
    setvbuf(stdout, NULL, _IONBF, 0);
    int sel = argc > 1 ? atoi(argv[1]) : -1, kid = 0;
    unsigned* out; (void)hipMalloc(&out, 4096);
    unsigned h[64];
    auto rep = [&](const char* name, double ms) {
        (void)hipMemcpy(h, out, 256, hipMemcpyDeviceToHost);
        unsigned cs = 0; for (int i = 0; i < 64; i++) cs = cs * 1000003u + h[i];
        printf("%-44s %8.3f ms  %6.2f ns/step  checksum %08x\n", name, ms, ms * 1e6 / (4.0 * ITER), cs);
    };
#define RUN(NAME, K) if (sel < 0 || sel == kid) { rep(NAME, run([&] { K<<<1, 64>>>(out); })); } kid++;
    RUN("P0 shipped (11)", k_p0)
    RUN("P0 no cmp/branch (9)", k_p0nb)
    RUN("P0 no rank store (10)", k_p0ns)
    RUN("P2 all-VALU chain (11)", k_p2)
    RUN("P3 P2 + andn2 branch (10)", k_p3)
    RUN("P4 P3 + hoisted readlanes (10)", k_p4)
    RUN("P5 P4 + packed ranks (11.25)", k_p5)
    RUN("P6 chain only (6)", k_p6)
    RUN("P4NB P4 without branch (9)", k_p4nb)
    RUN("P7 interleaved scalar side (10)", k_p7)
    RUN("P8 P7 + late rank store (10)", k_p8)
    RUN("P9 P8 + branch one step late (10)", k_p9)
    RUN("P10 P4 + one-hot rank record (9)", k_p10)
    RUN("P11 P10 + late branch (9)", k_p11)
    RUN("P12 P11 + SDWA packed literals (8)", k_p12)
    RUN("P13 P12 without rank record (7)", k_p13)
    RUN("P14 P13, branch right after its andn2 (7)", k_p14)
    RUN("P15 P13, one test per two literals (7.5)", k_p15)
    RUN("P16 VALU-only slow test, blanked table (8)", k_p16)
    return 0;
}
