// Issue-rate microbenchmark for one lone wavefront on gfx950 (test tool, not product).
// Each kernel runs ITER iterations of a 16-instruction unrolled body; host prints cycles/instruction
// from s_memrealtime-free wall timing (hipEvent) at the reported clock.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#define ITER 200000
#define REP4(x) x x x x
#define REP16(x) REP4(x) REP4(x) REP4(x) REP4(x)

__global__ void k_valu_dep(unsigned* out) {
    unsigned a = threadIdx.x, b = 3;
    for (int i = 0; i < ITER; i++) { REP16(asm volatile("v_add_u32 %0, %0, %1" : "+v"(a) : "v"(b));) }
    out[threadIdx.x] = a;
}
__global__ void k_valu_ind(unsigned* out) {
    unsigned a = threadIdx.x, b = 3, c = 5, d = 7, e = 9;
    for (int i = 0; i < ITER; i++) {
        REP4(asm volatile("v_add_u32 %0, %0, %4\n\tv_add_u32 %1, %1, %4\n\tv_add_u32 %2, %2, %4\n\tv_add_u32 %3, %3, %4"
                          : "+v"(a), "+v"(c), "+v"(d), "+v"(e) : "v"(b));)
    }
    out[threadIdx.x] = a + c + d + e;
}
__global__ void k_salu_dep(unsigned* out) {
    unsigned a = 1;
    for (int i = 0; i < ITER; i++) { REP16(asm volatile("s_add_u32 %0, %0, 3" : "+s"(a) : : "scc");) }
    out[threadIdx.x] = a;
}
__global__ void k_salu_ind(unsigned* out) {
    unsigned a = 1, c = 2, d = 3, e = 4;
    for (int i = 0; i < ITER; i++) {
        REP4(asm volatile("s_add_u32 %0, %0, 3\n\ts_add_u32 %1, %1, 3\n\ts_add_u32 %2, %2, 3\n\ts_add_u32 %3, %3, 3"
                          : "+s"(a), "+s"(c), "+s"(d), "+s"(e) : : "scc");)
    }
    out[threadIdx.x] = a + c + d + e;
}
// VALU -> vcc -> SALU -> VALU round trip: 4 instructions per link, 4 links per body (=16 instrs)
__global__ void k_mixed_chain(unsigned* out) {
    unsigned t = threadIdx.x, c = 17; unsigned long long m;
    for (int i = 0; i < ITER; i++) {
        REP4(asm volatile("v_cmp_ne_u32_e32 vcc, %2, %0\n\ts_not_b64 %1, vcc\n\ts_lshr_b64 %1, %1, 1\n\tv_cndmask_b32_e64 %0, %0, %2, %1"
                          : "+v"(t), "=&s"(m) : "v"(c) : "vcc", "scc");)
    }
    out[threadIdx.x] = t;
}
// dependent DPP chain
__global__ void k_dpp_dep(unsigned* out) {
    unsigned t = threadIdx.x;
    for (int i = 0; i < ITER; i++) { REP16(asm volatile("v_mov_b32_dpp %0, %0 wave_shr:1 row_mask:0xf bank_mask:0xf" : "+v"(t));) }
    out[threadIdx.x] = t;
}
// readlane -> salu -> writelane chain (3 instrs + nop-free), 5 links + 1 = 16
__global__ void k_lane_chain(unsigned* out) {
    unsigned t = threadIdx.x; unsigned s;
    for (int i = 0; i < ITER; i++) {
        REP4(asm volatile("v_readlane_b32 %1, %0, 5\n\ts_add_u32 %1, %1, 1\n\tv_writelane_b32 %0, %1, 6\n\tv_add_u32 %0, %0, %0"
                          : "+v"(t), "=&s"(s) : : "scc");)
    }
    out[threadIdx.x] = t;
}
// the k_mtf_dense fast step as shipped (7 asm instrs) + scalar compare/branch-free consume (dependent through t0)
__global__ void k_mtf_step(unsigned* out) {
    unsigned t0 = threadIdx.x, cv; unsigned c = 9, idx, acc = 0; unsigned long long m0, m1;
    for (int i = 0; i < ITER; i++) {
        REP4(asm volatile("v_mov_b32 %[cv], %[c]\n\tv_cmp_ne_u32_e32 vcc, %[cv], %[t0]\n\ts_not_b64 %[m0], vcc\n\t"
                          "s_lshr_b64 %[m1], %[m0], 1\n\t"
                          "v_cndmask_b32_dpp %[t0], %[t0], %[t0], vcc wave_shr:1 row_mask:0xf bank_mask:0xf\n\t"
                          "v_cndmask_b32_e64 %[t0], %[t0], %[cv], %[m1]\n\ts_ff1_i32_b64 %[i], %[m0]\n\t"
                          "s_add_u32 %[acc], %[acc], %[i]\n\ts_add_u32 %[c], %[c], 1\n\ts_and_b32 %[c], %[c], 15"
                          : [t0] "+v"(t0), [cv] "=&v"(cv), [m0] "=&s"(m0), [m1] "=&s"(m1), [i] "=&s"(idx), [acc] "+s"(acc), [c] "+s"(c)
                          : : "vcc", "scc");)
    }
    out[threadIdx.x] = t0 + acc;
}
// two independent mixed chains interleaved (does ILP hide the VALU<->SALU latency?)
__global__ void k_mixed_2chains(unsigned* out) {
    unsigned t = threadIdx.x, u = threadIdx.x ^ 5, c = 17; unsigned long long m, n;
    for (int i = 0; i < ITER; i++) {
        REP4(asm volatile("v_cmp_ne_u32_e64 %1, %4, %0\n\tv_cmp_ne_u32_e64 %3, %4, %2\n\t"
                          "s_not_b64 %1, %1\n\ts_not_b64 %3, %3\n\t"
                          "v_cndmask_b32_e64 %0, %0, %4, %1\n\tv_cndmask_b32_e64 %2, %2, %4, %3\n\t"
                          "v_add_u32 %0, %0, 1\n\tv_add_u32 %2, %2, 1"
                          : "+v"(t), "=&s"(m), "+v"(u), "=&s"(n) : "v"(c) : "scc");)
    }
    out[threadIdx.x] = t + u;
}
// same but one chain (8 instrs per REP -> use 2 links)
__global__ void k_mixed_1chain(unsigned* out) {
    unsigned t = threadIdx.x, c = 17; unsigned long long m;
    for (int i = 0; i < ITER; i++) {
        REP4(asm volatile("v_cmp_ne_u32_e64 %1, %2, %0\n\ts_not_b64 %1, %1\n\tv_cndmask_b32_e64 %0, %0, %2, %1\n\tv_add_u32 %0, %0, 1"
                          : "+v"(t), "=&s"(m) : "v"(c) : "scc");)
    }
    out[threadIdx.x] = t;
}
// LDS dependent load chain (pointer chase in LDS)
__global__ void k_lds_chase(unsigned* out) {
    __shared__ unsigned tab[1024];
    for (int i = threadIdx.x; i < 1024; i += 64) tab[i] = ((i * 7 + 13) & 1023) * 4;
    __syncthreads();
    unsigned p = threadIdx.x * 4;
    for (int i = 0; i < ITER; i++) { REP16(asm volatile("ds_read_b32 %0, %0\n\ts_waitcnt lgkmcnt(0)" : "+v"(p));) }
    out[threadIdx.x] = p;
}
// global (L2/L1-resident) dependent load chain
__global__ void k_glb_chase(unsigned* out, const unsigned* tab) {
    unsigned p = threadIdx.x & 3;
    for (int i = 0; i < ITER / 8; i++) { REP16(p = __builtin_nontemporal_load(tab + p) & 1023;) }
    out[threadIdx.x] = p;
}
__global__ void k_glb_chase_plain(unsigned* out, const unsigned* tab) {
    unsigned p = threadIdx.x & 3;
    for (int i = 0; i < ITER / 8; i++) { REP16(p = *(volatile const unsigned*)(tab + p) & 1023;) }
    out[threadIdx.x] = p;
}
// scalar (SMEM) dependent load chain
__global__ void k_smem_chase(unsigned* out, const unsigned* tab) {
    unsigned p = 0;
    for (int i = 0; i < ITER / 8; i++) {
        REP16(asm volatile("s_lshl_b32 %0, %0, 2\n\ts_load_dword %0, %1, %0\n\ts_waitcnt lgkmcnt(0)\n\ts_and_b32 %0, %0, 1023" : "+s"(p) : "s"(tab) : "scc");)
    }
    out[threadIdx.x] = p;
}


// candidate re-ordered MTF step: no instruction depends on its immediate predecessor
#define PIPE_STEP(CVK, CVN, K) \
    asm volatile("v_cmp_ne_u32_e32 vcc, %[" #CVK "], %[t0]\n\t" \
                 "v_readlane_b32 %[sc], %[v], 5\n\t" \
                 "s_not_b64 %[m0], vcc\n\t" \
                 "v_cndmask_b32_dpp %[t0], %[t0], %[t0], vcc wave_shr:1 row_mask:0xf bank_mask:0xf\n\t" \
                 "s_lshr_b64 %[m1], %[m0], 1\n\t" \
                 "s_and_b64 %[tmp], %[m0], 0x1fffff\n\t" \
                 "v_cndmask_b32_e64 %[t0], %[t0], %[" #CVK "], %[m1]\n\t" \
                 "s_ff1_i32_b64 %[i], %[m0]\n\t" \
                 "v_mov_b32 %[" #CVN "], %[sc]\n\t" \
                 "s_cbranch_scc0 1f\n\t" \
                 "1: v_writelane_b32 %[ranks], %[i], " #K "\n\t" \
                 : [t0] "+v"(t0), [a] "+v"(cva), [b] "+v"(cvb), [m0] "=&s"(m0), [m1] "=&s"(m1), [tmp] "=&s"(tmp), [i] "=&s"(idx), \
                   [sc] "=&s"(sc), [ranks] "+v"(ranks) : [v] "v"(v) : "vcc", "scc")
__global__ void k_mtf_pipe(unsigned* out) {
    unsigned t0 = threadIdx.x, cva = 5, cvb = 5, v = threadIdx.x & 15, ranks = 0, idx, sc; unsigned long long m0, m1, tmp;
    for (int i = 0; i < ITER; i++) { PIPE_STEP(a, b, 0); PIPE_STEP(b, a, 1); PIPE_STEP(a, b, 2); PIPE_STEP(b, a, 3); }
    out[threadIdx.x] = t0 + ranks;
}

template <class F> static double run(F launch, int reps = 3) {
    hipEvent_t a, b; hipEventCreate(&a); hipEventCreate(&b);
    float best = 1e30f;
    for (int r = 0; r < reps; r++) {
        hipEventRecord(a); launch(); hipEventRecord(b); hipEventSynchronize(b);
        float ms; hipEventElapsedTime(&ms, a, b); if (ms < best) best = ms;
    }
    return best;
}
int main(int argc, char** argv) {
    setvbuf(stdout, NULL, _IONBF, 0);
    int sel = argc > 1 ? atoi(argv[1]) : -1; int kid = 0;
    unsigned* out; hipMalloc(&out, 4096);
    unsigned* tab; hipMalloc(&tab, 4096);
    unsigned h[1024]; for (int i = 0; i < 1024; i++) h[i] = (i * 7 + 13) & 1023;
    hipMemcpy(tab, h, 4096, hipMemcpyHostToDevice);
    int clk = 0; hipDeviceGetAttribute(&clk, hipDeviceAttributeClockRate, 0);   // kHz
    printf("clock %d kHz\n", clk);
    auto rep = [&](const char* name, double ms, double instrs) {
        printf("%-28s %8.3f ms  %6.2f cycles/instr (at max clock)  %6.1f ns/instr\n", name, ms, ms * 1e-3 * clk * 1e3 / instrs, ms * 1e6 / instrs);
    };
    double n16 = 16.0 * ITER;
    if (sel < 0 || sel == kid) {rep("valu dependent", run([&] { k_valu_dep<<<1, 64>>>(out); }), n16);} kid++;
    if (sel < 0 || sel == kid) {rep("valu independent x4", run([&] { k_valu_ind<<<1, 64>>>(out); }), n16);} kid++;
    if (sel < 0 || sel == kid) {rep("salu dependent", run([&] { k_salu_dep<<<1, 64>>>(out); }), n16);} kid++;
    if (sel < 0 || sel == kid) {rep("salu independent x4", run([&] { k_salu_ind<<<1, 64>>>(out); }), n16);} kid++;
    if (sel < 0 || sel == kid) {rep("v_cmp>s_not>s_lshr>v_cndmask", run([&] { k_mixed_chain<<<1, 64>>>(out); }), n16);} kid++;
    if (sel < 0 || sel == kid) {rep("dpp dependent", run([&] { k_dpp_dep<<<1, 64>>>(out); }), n16);} kid++;
    if (sel < 0 || sel == kid) {rep("readlane>salu>writelane>valu", run([&] { k_lane_chain<<<1, 64>>>(out); }), n16);} kid++;
    if (sel < 0 || sel == kid) {rep("mtf fast step (10 instr)", run([&] { k_mtf_step<<<1, 64>>>(out); }), 40.0 * ITER);} kid++;
    if (sel < 0 || sel == kid) {rep("mixed 1 chain (4/link)", run([&] { k_mixed_1chain<<<1, 64>>>(out); }), n16);} kid++;
    if (sel < 0 || sel == kid) {rep("mixed 2 chains interleaved", run([&] { k_mixed_2chains<<<1, 64>>>(out); }), 32.0 * ITER);} kid++;
    if (sel < 0 || sel == kid) {rep("lds chase (per load)", run([&] { k_lds_chase<<<1, 64>>>(out); }), n16);} kid++;
    if (sel < 0 || sel == kid) {rep("global chase nt (per load)", run([&] { k_glb_chase<<<1, 64>>>(out, tab); }), 16.0 * (ITER / 8));} kid++;
    if (sel < 0 || sel == kid) {rep("global chase (per load)", run([&] { k_glb_chase_plain<<<1, 64>>>(out, tab); }), 16.0 * (ITER / 8));} kid++;
    if (sel < 0 || sel == kid) {rep("smem chase (per load)", run([&] { k_smem_chase<<<1, 64>>>(out, tab); }), 16.0 * (ITER / 8));} kid++;
    // same kernels with 4 waves on one CU (do co-resident waves slow each other?)
    if (sel < 0 || sel == kid) {rep("valu dependent, 4 waves/WG", run([&] { k_valu_dep<<<1, 256>>>(out); }), n16);} kid++;
    if (sel < 0 || sel == kid) {rep("mtf fast step, 256 WGs", run([&] { k_mtf_step<<<256, 64>>>(out); }), 40.0 * ITER);} kid++;
    if (sel < 0 || sel == kid) {rep("mtf pipelined step (11 instr)", run([&] { k_mtf_pipe<<<1, 64>>>(out); }), 44.0 * ITER);} kid++;
    return 0;
}
