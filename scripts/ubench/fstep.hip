// Test-free "front" step of the rank chain (ranks 0..20 only: swap with the left neighbour), timed on one lone wavefront AND
// checked against a host emulation of the same steps: does a DPP read of a register written by the previous instruction
// need the gfx9 wait states on gfx950, and what does the s_nop cost?  (test tool, not product)
//   hipcc --offload-arch=gfx950 -O2 -o scripts/ubench/fstep scripts/ubench/fstep.hip
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>
#define NLIT (1 << 22)

// one 64-literal tile: sixteen dwords of literals in SGPRs, five VALU per literal
#define F5(PK, B) \
    "v_cmp_ne_u32_sdwa vcc, %[" #PK "], %[tf] src0_sel:BYTE_" #B " src1_sel:DWORD\n\t" \
    "v_mov_b32_dpp %[up], %[tf] wave_shl:1 row_mask:0xf bank_mask:0xf\n\t" \
    "v_cmp_eq_u32_sdwa %[m1], %[" #PK "], %[up] src0_sel:BYTE_" #B " src1_sel:DWORD\n\t" \
    "v_cndmask_b32_dpp %[tf], %[tf], %[tf], vcc wave_shr:1 row_mask:0xf bank_mask:0xf\n\t" \
    "v_cndmask_b32_e64 %[tf], %[tf], %[up], %[m1]\n\t"
#define F5N(PK, B) \
    "v_cmp_ne_u32_sdwa vcc, %[" #PK "], %[tf] src0_sel:BYTE_" #B " src1_sel:DWORD\n\t" \
    "s_nop 0\n\t" \
    "v_mov_b32_dpp %[up], %[tf] wave_shl:1 row_mask:0xf bank_mask:0xf\n\t" \
    "v_cmp_eq_u32_sdwa %[m1], %[" #PK "], %[up] src0_sel:BYTE_" #B " src1_sel:DWORD\n\t" \
    "v_cndmask_b32_dpp %[tf], %[tf], %[tf], vcc wave_shr:1 row_mask:0xf bank_mask:0xf\n\t" \
    "v_cndmask_b32_e64 %[tf], %[tf], %[up], %[m1]\n\t"
// dpp mov first (directly behind the previous step's last write of tf)
#define F5D(PK, B) \
    "v_mov_b32_dpp %[up], %[tf] wave_shl:1 row_mask:0xf bank_mask:0xf\n\t" \
    "v_cmp_ne_u32_sdwa vcc, %[" #PK "], %[tf] src0_sel:BYTE_" #B " src1_sel:DWORD\n\t" \
    "v_cmp_eq_u32_sdwa %[m1], %[" #PK "], %[up] src0_sel:BYTE_" #B " src1_sel:DWORD\n\t" \
    "v_cndmask_b32_dpp %[tf], %[tf], %[tf], vcc wave_shr:1 row_mask:0xf bank_mask:0xf\n\t" \
    "v_cndmask_b32_e64 %[tf], %[tf], %[up], %[m1]\n\t"
// the shipped state-only step (with the slow-path test and its late branch), for comparison
#define F7(PK, B) \
    "v_cmp_ne_u32_sdwa vcc, %[" #PK "], %[tf] src0_sel:BYTE_" #B " src1_sel:DWORD\n\t" \
    "v_mov_b32_dpp %[up], %[tf] wave_shl:1 row_mask:0xf bank_mask:0xf\n\t" \
    "v_cmp_eq_u32_sdwa %[m1], %[" #PK "], %[up] src0_sel:BYTE_" #B " src1_sel:DWORD\n\t" \
    "s_cbranch_scc0 9f\n\t" \
    "v_cndmask_b32_dpp %[tf], %[tf], %[tf], vcc wave_shr:1 row_mask:0xf bank_mask:0xf\n\t" \
    "v_cndmask_b32_e64 %[tf], %[tf], %[up], %[m1]\n\t" \
    "s_andn2_b64 s[90:91], 0x1fffff, vcc\n\t"
#define W4(S, PK) S(PK, 0) S(PK, 1) S(PK, 2) S(PK, 3)
#define TILE(S) W4(S, p0) W4(S, p1) W4(S, p2) W4(S, p3) W4(S, p4) W4(S, p5) W4(S, p6) W4(S, p7) W4(S, p8) W4(S, p9) W4(S, p10) W4(S, p11) W4(S, p12) W4(S, p13) W4(S, p14) W4(S, p15)
typedef unsigned Tile16 __attribute__((ext_vector_type(16)));
#define KERNEL(NAME, S, PRE)                                                                                              \
    __global__ void NAME(const unsigned char* lit, unsigned* out, unsigned long long* cyc) {                               \
        unsigned tf = threadIdx.x <= 20 ? 65 + threadIdx.x : 0x100 + threadIdx.x, up = 0xffffffffu;                          \
        unsigned long long m1;                                                                                             \
        Tile16 pk;                                                                                                         \
        const unsigned long long t0 = __builtin_readcyclecounter();                                                        \
        for (unsigned base = 0; base < NLIT; base += 64) {                                                                 \
            const unsigned char* p = lit + base;                                                                           \
            asm volatile("s_load_dwordx16 %0, %1, 0x0\n\ts_waitcnt lgkmcnt(0)" : "=s"(pk) : "s"(p));                        \
            asm volatile(PRE TILE(S) "9:\n\t"                                                                              \
                         : [tf] "+v"(tf), [up] "+v"(up), [m1] "=&s"(m1)                                                    \
                         : [p0] "s"(pk[0]), [p1] "s"(pk[1]), [p2] "s"(pk[2]), [p3] "s"(pk[3]), [p4] "s"(pk[4]), [p5] "s"(pk[5]), [p6] "s"(pk[6]), [p7] "s"(pk[7]), \
                           [p8] "s"(pk[8]), [p9] "s"(pk[9]), [p10] "s"(pk[10]), [p11] "s"(pk[11]), [p12] "s"(pk[12]), [p13] "s"(pk[13]), [p14] "s"(pk[14]), [p15] "s"(pk[15]) \
                         : "vcc", "scc", "s90", "s91");                                                                    \
        }                                                                                                                  \
        if (threadIdx.x == 0) *cyc = __builtin_readcyclecounter() - t0;                                                    \
        out[threadIdx.x] = tf;                                                                                             \
    }
KERNEL(k_f5, F5, "")
KERNEL(k_f5n, F5N, "")
KERNEL(k_f5d, F5D, "")
KERNEL(k_f7, F7, "s_cmp_eq_u32 0, 0\n\t")

int main() {
    std::vector<unsigned char> lit(NLIT);
    unsigned s = 12345;
    for (int i = 0; i < NLIT; i++) {                 // symbols 65..85 are the front; ~3 % others (no-ops for the front), skewed ranks
        s = s * 1664525u + 1013904223u;
        const unsigned r = (s >> 8) % 1000;
        unsigned v = 65 + (r < 300 ? 0 : r < 500 ? 1 : r < 640 ? 2 : r < 740 ? 3 : 4 + (s >> 20) % 17);
        if (r >= 970) v = 200 + (s >> 20) % 40;
        lit[i] = (unsigned char)v;
    }
    unsigned ref[21];
    for (int i = 0; i < 21; i++) ref[i] = 65 + i;
    for (int i = 0; i < NLIT; i++) for (int k = 1; k < 21; k++) if (ref[k] == lit[i]) { unsigned t = ref[k]; ref[k] = ref[k - 1]; ref[k - 1] = t; break; }
    unsigned char* d_lit; unsigned* d_out; unsigned long long* d_cyc;
    hipMalloc(&d_lit, NLIT + 256); hipMalloc(&d_out, 256); hipMalloc(&d_cyc, 8);
    hipMemcpy(d_lit, lit.data(), NLIT, hipMemcpyHostToDevice);
    struct { const char* name; void (*k)(const unsigned char*, unsigned*, unsigned long long*); } ks[] = {
        {"F5  (5 VALU, cmp first, no wait state)", k_f5}, {"F5N (5 VALU + s_nop 0 before the dpp mov)", k_f5n}, {"F5D (5 VALU, dpp mov first)", k_f5d}, {"F7  (shipped state-only step)", k_f7}};
    for (auto& kk : ks) {
        unsigned out[64]; unsigned long long cyc = 0;
        hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
        hipLaunchKernelGGL(kk.k, dim3(1), dim3(64), 0, 0, d_lit, d_out, d_cyc);
        hipDeviceSynchronize();
        hipEventRecord(e0); hipLaunchKernelGGL(kk.k, dim3(1), dim3(64), 0, 0, d_lit, d_out, d_cyc); hipEventRecord(e1);
        hipDeviceSynchronize();
        float ms = 0; hipEventElapsedTime(&ms, e0, e1);
        hipMemcpy(out, d_out, 256, hipMemcpyDeviceToHost); hipMemcpy(&cyc, d_cyc, 8, hipMemcpyDeviceToHost);
        int bad = 0; for (int i = 0; i < 21; i++) bad += out[i] != ref[i];
        printf("%-46s %6.2f ns per literal  %5.1f cycles  table %s\n", kk.name, ms * 1e6 / NLIT, (double)cyc / NLIT, bad ? "WRONG" : "ok");
    }
    return 0;
}
