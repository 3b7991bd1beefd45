// Round 4: candidate rank-chain steps with FOUR table instructions, timed on one lone wavefront and checked against a host
// emulation of the same steps (test tool, not product).
//   hipcc --offload-arch=gfx950 -O2 -o scripts/ubench/gstep scripts/ubench/gstep.hip
// The shipped step (F7 in fstep.hip) is five VALU + test + late branch.  Two of the five VALU only produce "the hit mask one lane
// down" (v_mov_dpp up + v_cmp_eq on it) and the value that lands there (up).  Here:
//   * the landing mask is the compare's own mask shifted by the scalar unit (s_ashr_i64 vcc, vcc, 1: the ne-mask keeps its sign,
//     so lane 63 -- a pad -- never lands);
//   * the landing value is the literal itself, taken as an SDWA byte of a VECTOR register that holds the same four literals in
//     every lane (one global_load_dwordx4 with a wave-uniform address per sixteen literals) -- a scalar source would be a second
//     constant-bus read next to VCC, which gfx9 does not allow.
// Variants:
//   F5    fstep.hip's test-free five-VALU step (baseline)
//   G4N   cmp, s_nop, dpp select (hit lane takes its left neighbour), s_ashr, sdwa select (lane below takes the literal)
//   G4    the same without the s_nop (DPP read of tf two instructions behind its VALU write?)
//   G6A   G4 + s_andn2 test behind the compare + branch in the same step (no s_nop: the branch sits between cmp and dpp select)
//   G6B   table moved up one lane, lane 0 disabled by EXEC; test on the SHIFTED mask behind s_ashr; branch one step late
//   G4X   G4N with EXEC = low 32 lanes only (does a half-empty wavefront issue faster?)
//   DPPX  semantics probe: is a DPP source lane that EXEC disables "invalid" (write suppressed with bound_ctrl:0)?
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>
#define NLIT (1 << 22)

#define F5(PK, B) \
    "v_cmp_ne_u32_sdwa vcc, %[" #PK "], %[tf] src0_sel:BYTE_" #B " src1_sel:DWORD\n\t" \
    "s_nop 0\n\t" \
    "v_mov_b32_dpp %[up], %[tf] wave_shl:1 row_mask:0xf bank_mask:0xf\n\t" \
    "v_cmp_eq_u32_sdwa %[m1], %[" #PK "], %[up] src0_sel:BYTE_" #B " src1_sel:DWORD\n\t" \
    "v_cndmask_b32_dpp %[tf], %[tf], %[tf], vcc wave_shr:1 row_mask:0xf bank_mask:0xf\n\t" \
    "v_cndmask_b32_e64 %[tf], %[tf], %[up], %[m1]\n\t"
#define G4N(PK, B) \
    "v_cmp_ne_u32_sdwa vcc, %[" #PK "], %[tf] src0_sel:BYTE_" #B " src1_sel:DWORD\n\t" \
    "s_nop 0\n\t" \
    "v_cndmask_b32_dpp %[tf], %[tf], %[tf], vcc wave_shr:1 row_mask:0xf bank_mask:0xf\n\t" \
    "s_ashr_i64 vcc, vcc, 1\n\t" \
    "v_cndmask_b32_sdwa %[tf], %[" #PK "], %[tf], vcc dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:BYTE_" #B " src1_sel:DWORD\n\t"
#define G4(PK, B) \
    "v_cmp_ne_u32_sdwa vcc, %[" #PK "], %[tf] src0_sel:BYTE_" #B " src1_sel:DWORD\n\t" \
    "v_cndmask_b32_dpp %[tf], %[tf], %[tf], vcc wave_shr:1 row_mask:0xf bank_mask:0xf\n\t" \
    "s_ashr_i64 vcc, vcc, 1\n\t" \
    "v_cndmask_b32_sdwa %[tf], %[" #PK "], %[tf], vcc dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:BYTE_" #B " src1_sel:DWORD\n\t"
// tested, branch in the step: SCC = "hit in lanes 0..20" from the ne-mask
#define G6A(PK, B) \
    "v_cmp_ne_u32_sdwa vcc, %[" #PK "], %[tf] src0_sel:BYTE_" #B " src1_sel:DWORD\n\t" \
    "s_andn2_b64 s[90:91], 0x1fffff, vcc\n\t" \
    "s_cbranch_scc0 9f\n\t" \
    "v_cndmask_b32_dpp %[tf], %[tf], %[tf], vcc wave_shr:1 row_mask:0xf bank_mask:0xf\n\t" \
    "s_ashr_i64 vcc, vcc, 1\n\t" \
    "v_cndmask_b32_sdwa %[tf], %[" #PK "], %[tf], vcc dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:BYTE_" #B " src1_sel:DWORD\n\t"
// tested, table in lanes 1..21 (lane 0 off), test on the shifted mask, branch one step late (behind the next compare)
#define G6B(PK, B) \
    "v_cmp_ne_u32_sdwa vcc, %[" #PK "], %[tf] src0_sel:BYTE_" #B " src1_sel:DWORD\n\t" \
    "s_cbranch_scc0 9f\n\t" \
    "v_cndmask_b32_dpp %[tf], %[tf], %[tf], vcc wave_shr:1 row_mask:0xf bank_mask:0xf\n\t" \
    "s_ashr_i64 vcc, vcc, 1\n\t" \
    "s_andn2_b64 s[90:91], 0x1fffff, vcc\n\t" \
    "v_cndmask_b32_sdwa %[tf], %[" #PK "], %[tf], vcc dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:BYTE_" #B " src1_sel:DWORD\n\t"
// G6C: the same six instructions, one filler per dependent hop: cmp, dpp select, s_ashr, test, sdwa select, branch (one slot behind its test,
// in front of the next compare: the out-of-line part finds step K complete and nothing of step K + 1)
#define G6C(PK, B) \
    "v_cmp_ne_u32_sdwa vcc, %[" #PK "], %[tf] src0_sel:BYTE_" #B " src1_sel:DWORD\n\t" \
    "v_cndmask_b32_dpp %[tf], %[tf], %[tf], vcc wave_shr:1 row_mask:0xf bank_mask:0xf\n\t" \
    "s_ashr_i64 vcc, vcc, 1\n\t" \
    "s_andn2_b64 s[90:91], 0x1fffff, vcc\n\t" \
    "v_cndmask_b32_sdwa %[tf], %[" #PK "], %[tf], vcc dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:BYTE_" #B " src1_sel:DWORD\n\t" \
    "s_cbranch_scc0 9f\n\t"
// G6D: test behind the sdwa select, branch behind the next compare
#define G6D(PK, B) \
    "v_cmp_ne_u32_sdwa vcc, %[" #PK "], %[tf] src0_sel:BYTE_" #B " src1_sel:DWORD\n\t" \
    "s_cbranch_scc0 9f\n\t" \
    "v_cndmask_b32_dpp %[tf], %[tf], %[tf], vcc wave_shr:1 row_mask:0xf bank_mask:0xf\n\t" \
    "s_ashr_i64 vcc, vcc, 1\n\t" \
    "v_cndmask_b32_sdwa %[tf], %[" #PK "], %[tf], vcc dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:BYTE_" #B " src1_sel:DWORD\n\t" \
    "s_andn2_b64 s[90:91], 0x1fffff, vcc\n\t"
// G6E: G6B with the freeze instead of a branch: an exception switches every lane off (s_cselect of EXEC), one branch per tile would follow
#define G6E(PK, B) \
    "v_cmp_ne_u32_sdwa vcc, %[" #PK "], %[tf] src0_sel:BYTE_" #B " src1_sel:DWORD\n\t" \
    "v_cndmask_b32_dpp %[tf], %[tf], %[tf], vcc wave_shr:1 row_mask:0xf bank_mask:0xf\n\t" \
    "s_ashr_i64 vcc, vcc, 1\n\t" \
    "s_andn2_b64 s[90:91], 0x1fffff, vcc\n\t" \
    "v_cndmask_b32_sdwa %[tf], %[" #PK "], %[tf], vcc dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:BYTE_" #B " src1_sel:DWORD\n\t" \
    "s_cselect_b64 exec, exec, 0\n\t"
// Round 5 (gfx940/950 owe TWO wait states between a VALU write of VCC and a VALU read of it -- LLVM's hazard recognizer
// pads a plain v_cmp; v_cndmask with s_nop 0 on gfx950 -- and G6D has one, the late branch):
// G7N: G6D with an s_nop 0 behind the late branch (seven slots)
#define G7N(PK, B) \
    "v_cmp_ne_u32_sdwa vcc, %[" #PK "], %[tf] src0_sel:BYTE_" #B " src1_sel:DWORD\n\t" \
    "s_cbranch_scc0 9f\n\t" \
    "s_nop 0\n\t" \
    "v_cndmask_b32_dpp %[tf], %[tf], %[tf], vcc wave_shr:1 row_mask:0xf bank_mask:0xf\n\t" \
    "s_ashr_i64 vcc, vcc, 1\n\t" \
    "v_cndmask_b32_sdwa %[tf], %[" #PK "], %[tf], vcc dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:BYTE_" #B " src1_sel:DWORD\n\t" \
    "s_andn2_b64 s[90:91], 0x1fffff, vcc\n\t"
// G7M: the s_nop in front of the branch
#define G7M(PK, B) \
    "v_cmp_ne_u32_sdwa vcc, %[" #PK "], %[tf] src0_sel:BYTE_" #B " src1_sel:DWORD\n\t" \
    "s_nop 0\n\t" \
    "s_cbranch_scc0 9f\n\t" \
    "v_cndmask_b32_dpp %[tf], %[tf], %[tf], vcc wave_shr:1 row_mask:0xf bank_mask:0xf\n\t" \
    "s_ashr_i64 vcc, vcc, 1\n\t" \
    "v_cndmask_b32_sdwa %[tf], %[" #PK "], %[tf], vcc dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:BYTE_" #B " src1_sel:DWORD\n\t" \
    "s_andn2_b64 s[90:91], 0x1fffff, vcc\n\t"
// H6: six slots WITH two instructions between the compare and the DPP select.  The shifted mask goes to its own SGPR pair
// (s_ashr_i64 s[92:93], vcc, 1), so the next compare does not overwrite it and the test of step K - 1 moves BEHIND the compare
// of step K, next to its branch: cmp(K), test(K-1), branch(K-1), dpp(K), ashr(K), select(K).  The second select is then a VOP3
// v_cndmask with an explicit mask, which cannot take an SDWA byte: the literal must be a whole dword of a vector register
// (sixteen wave-uniform dwordx4 loads per tile instead of four).
#define H6(L) \
    "v_cmp_ne_u32_e32 vcc, %[" #L "], %[tf]\n\t" \
    "s_andn2_b64 s[90:91], 0x1fffff, s[92:93]\n\t" \
    "s_cbranch_scc0 9f\n\t" \
    "v_cndmask_b32_dpp %[tf], %[tf], %[tf], vcc wave_shr:1 row_mask:0xf bank_mask:0xf\n\t" \
    "s_ashr_i64 s[92:93], vcc, 1\n\t" \
    "v_cndmask_b32_e64 %[tf], %[" #L "], %[tf], s[92:93]\n\t"
// H6B: the test one slot earlier in the NEXT step is not possible (it needs ashr(K)); variant with branch first, then test of the
// step before it is what G6D does.  H6N: H6 with the test and the branch swapped against each other's step (branch(K-2) ahead of test(K-1))
#define H6N(L) \
    "v_cmp_ne_u32_e32 vcc, %[" #L "], %[tf]\n\t" \
    "s_cbranch_scc0 9f\n\t" \
    "s_andn2_b64 s[90:91], 0x1fffff, s[92:93]\n\t" \
    "v_cndmask_b32_dpp %[tf], %[tf], %[tf], vcc wave_shr:1 row_mask:0xf bank_mask:0xf\n\t" \
    "s_ashr_i64 s[92:93], vcc, 1\n\t" \
    "v_cndmask_b32_e64 %[tf], %[" #L "], %[tf], s[92:93]\n\t"
#define W4(S, PK) S(PK, 0) S(PK, 1) S(PK, 2) S(PK, 3)
#define TILE(S) W4(S, p0) W4(S, p1) W4(S, p2) W4(S, p3) W4(S, p4) W4(S, p5) W4(S, p6) W4(S, p7) W4(S, p8) W4(S, p9) W4(S, p10) W4(S, p11) W4(S, p12) W4(S, p13) W4(S, p14) W4(S, p15)
typedef unsigned Tile16 __attribute__((ext_vector_type(16)));
typedef unsigned V4 __attribute__((ext_vector_type(4)));

// scalar-literal form (F5 only)
#define KERNEL_S(NAME, S, PRE)                                                                                            \
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
// vector-literal form: every lane loads the same 64 bytes (four dwordx4 with a wave-uniform address), the NEXT tile's while this
// one runs.  SHIFT = 1 puts the table into lanes 1..21 and switches lane 0 off (EXECMASK).
#define KERNEL_V(NAME, S, PRE, SHIFT, EXECMASK)                                                                            \
    __global__ void NAME(const unsigned char* lit, unsigned* out, unsigned long long* cyc) {                               \
        const unsigned ln = threadIdx.x - (SHIFT);                                                                          \
        unsigned tf = ln <= 20u ? 65 + ln : 0x100 + threadIdx.x;                                                            \
        V4 a0, a1, a2, a3, b0, b1, b2, b3;                                                                                 \
        const V4* q = (const V4*)lit;                                                                                      \
        a0 = q[0]; a1 = q[1]; a2 = q[2]; a3 = q[3];                                                                        \
        const unsigned long long t0 = __builtin_readcyclecounter();                                                        \
        const unsigned long long ex = (EXECMASK);                                                                          \
        asm volatile("s_mov_b64 exec, %0" ::"s"(ex));                                                                      \
        for (unsigned base = 0; base < NLIT; base += 64) {                                                                 \
            const V4* nq = (const V4*)(lit + base + 64);                                                                   \
            b0 = nq[0]; b1 = nq[1]; b2 = nq[2]; b3 = nq[3];                                                                \
            asm volatile(PRE TILE(S) "9:\n\t"                                                                              \
                         : [tf] "+v"(tf)                                                                                   \
                         : [p0] "v"(a0[0]), [p1] "v"(a0[1]), [p2] "v"(a0[2]), [p3] "v"(a0[3]), [p4] "v"(a1[0]), [p5] "v"(a1[1]), [p6] "v"(a1[2]), [p7] "v"(a1[3]), \
                           [p8] "v"(a2[0]), [p9] "v"(a2[1]), [p10] "v"(a2[2]), [p11] "v"(a2[3]), [p12] "v"(a3[0]), [p13] "v"(a3[1]), [p14] "v"(a3[2]), [p15] "v"(a3[3]) \
                         : "vcc", "scc", "s90", "s91");                                                                    \
            a0 = b0; a1 = b1; a2 = b2; a3 = b3;                                                                            \
        }                                                                                                                  \
        asm volatile("s_mov_b64 exec, -1");                                                                                \
        if (threadIdx.x == 0) *cyc = __builtin_readcyclecounter() - t0;                                                    \
        out[threadIdx.x] = tf;                                                                                             \
    }
// dword-literal form (H6): sixteen wave-uniform dwordx4 loads per tile, the NEXT tile's while this one runs
#define D4(S, Q) S(Q##0) S(Q##1) S(Q##2) S(Q##3)
#define TILE_D(S) D4(S, a) D4(S, b) D4(S, c) D4(S, d) D4(S, e) D4(S, f) D4(S, g) D4(S, h) D4(S, i) D4(S, j) D4(S, k) D4(S, l) D4(S, m) D4(S, n) D4(S, o) D4(S, p)
#define DOP(Q, V) [Q##0] "v"(V[0]), [Q##1] "v"(V[1]), [Q##2] "v"(V[2]), [Q##3] "v"(V[3])
#define KERNEL_D(NAME, S)                                                                                                  \
    __global__ void NAME(const unsigned* lit, unsigned* out, unsigned long long* cyc) {                                    \
        const unsigned ln = threadIdx.x - 1;                                                                               \
        unsigned tf = ln <= 20u ? 65 + ln : 0x100 + threadIdx.x;                                                            \
        V4 A[16], B[16];                                                                                                   \
        const V4* q = (const V4*)lit;                                                                                      \
        for (int t = 0; t < 16; t++) A[t] = q[t];                                                                          \
        const unsigned long long t0 = __builtin_readcyclecounter();                                                        \
        asm volatile("s_mov_b64 exec, -2\n\ts_mov_b64 s[92:93], 0" ::: "s92", "s93");                                      \
        for (unsigned base = 0; base < NLIT; base += 64) {                                                                 \
            const V4* nq = (const V4*)(lit + base + 64);                                                                   \
            for (int t = 0; t < 16; t++) B[t] = nq[t];                                                                     \
            asm volatile("s_cmp_eq_u32 0, 0\n\t" TILE_D(S) "9:\n\t"                                                       \
                         : [tf] "+v"(tf)                                                                                   \
                         : DOP(a, A[0]), DOP(b, A[1]), DOP(c, A[2]), DOP(d, A[3]), DOP(e, A[4]), DOP(f, A[5]), DOP(g, A[6]), DOP(h, A[7]), \
                           DOP(i, A[8]), DOP(j, A[9]), DOP(k, A[10]), DOP(l, A[11]), DOP(m, A[12]), DOP(n, A[13]), DOP(o, A[14]), DOP(p, A[15]) \
                         : "vcc", "scc", "s90", "s91", "s92", "s93");                                                      \
            for (int t = 0; t < 16; t++) A[t] = B[t];                                                                      \
        }                                                                                                                  \
        asm volatile("s_mov_b64 exec, -1");                                                                                \
        if (threadIdx.x == 0) *cyc = __builtin_readcyclecounter() - t0;                                                    \
        out[threadIdx.x] = tf;                                                                                             \
    }
KERNEL_D(k_h6, H6)
KERNEL_D(k_h6n, H6N)
KERNEL_V(k_g7n, G7N, "s_cmp_eq_u32 0, 0\n\t", 1, ~1ull)
KERNEL_V(k_g7m, G7M, "s_cmp_eq_u32 0, 0\n\t", 1, ~1ull)
KERNEL_S(k_f5, F5, "")
KERNEL_V(k_g4n, G4N, "", 0, ~0ull)
KERNEL_V(k_g4, G4, "", 0, ~0ull)
KERNEL_V(k_g6a, G6A, "", 0, ~0ull)
KERNEL_V(k_g6b, G6B, "s_cmp_eq_u32 0, 0\n\t", 1, ~1ull)
KERNEL_V(k_g6c, G6C, "s_cmp_eq_u32 0, 0\n\t", 1, ~1ull)
KERNEL_V(k_g6d, G6D, "s_cmp_eq_u32 0, 0\n\t", 1, ~1ull)
KERNEL_V(k_g6e, G6E, "", 1, ~1ull)
KERNEL_V(k_g4x, G4N, "", 0, 0xffffffffull)
KERNEL_V(k_g4y, G4N, "", 0, 0x3fffffull)

// DPP source lane switched off by EXEC: lane 1 reads lane 0 (off) through wave_shr:1, bound_ctrl off.  out[1] keeps 111 if the write
// is suppressed, becomes lane 0's 500 if the disabled lane is read anyway.
__global__ void k_dppx(unsigned* out) {
    unsigned v = threadIdx.x == 0 ? 500u : 111u, w = 222u;
    asm volatile("s_mov_b64 exec, -2\n\t"
                 "s_nop 4\n\t"
                 "v_mov_b32_dpp %0, %1 wave_shr:1 row_mask:0xf bank_mask:0xf\n\t"
                 "s_mov_b64 exec, -1\n\t"
                 : "+v"(w) : "v"(v));
    out[threadIdx.x] = w;
}

int main() {
    std::vector<unsigned char> lit(NLIT + 256);
    unsigned s = 12345;
    for (int i = 0; i < NLIT + 256; i++) {           // symbols 65..85 are the front, skewed ranks, every literal in the front
        s = s * 1664525u + 1013904223u;
        const unsigned r = (s >> 8) % 1000;
        lit[i] = (unsigned char)(65 + (r < 300 ? 0 : r < 500 ? 1 : r < 640 ? 2 : r < 740 ? 3 : 4 + (s >> 20) % 17));
    }
    unsigned ref[21];
    for (int i = 0; i < 21; i++) ref[i] = 65 + i;
    for (int i = 0; i < NLIT; i++) for (int k = 1; k < 21; k++) if (ref[k] == lit[i]) { unsigned t = ref[k]; ref[k] = ref[k - 1]; ref[k - 1] = t; break; }
    unsigned char* d_lit; unsigned* d_out; unsigned long long* d_cyc;
    hipMalloc(&d_lit, NLIT + 512); hipMalloc(&d_out, 256); hipMalloc(&d_cyc, 8);
    hipMemcpy(d_lit, lit.data(), NLIT + 256, hipMemcpyHostToDevice);
    {
        unsigned out[64];
        hipLaunchKernelGGL(k_dppx, dim3(1), dim3(64), 0, 0, d_out);
        hipMemcpy(out, d_out, 256, hipMemcpyDeviceToHost);
        printf("DPPX: lane 1 reads an EXEC-disabled lane 0 through wave_shr:1 -> %u (111 = write suppressed, 500 = read anyway, 0 = zero)  lane 2 -> %u (expect 111)\n", out[1], out[2]);
    }
    std::vector<unsigned> lit32(NLIT + 256);
    for (int i = 0; i < NLIT + 256; i++) lit32[i] = lit[i];
    unsigned* d_lit32; hipMalloc(&d_lit32, (NLIT + 512) * 4);
    hipMemcpy(d_lit32, lit32.data(), (NLIT + 256) * 4, hipMemcpyHostToDevice);
    struct { const char* name; void (*k)(const unsigned*, unsigned*, unsigned long long*); } kd[] = {
        {"H6  (cmp, test K-1, branch K-1, dpp, ashr->s[92:93], vop3 select; dword literals)", k_h6},
        {"H6N (cmp, branch K-2, test K-1, dpp, ashr, vop3 select; dword literals)", k_h6n}};
    for (auto& kk : kd) {
        unsigned out[64]; unsigned long long cyc = 0;
        hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
        hipLaunchKernelGGL(kk.k, dim3(1), dim3(64), 0, 0, d_lit32, d_out, d_cyc);
        hipDeviceSynchronize();
        hipEventRecord(e0); hipLaunchKernelGGL(kk.k, dim3(1), dim3(64), 0, 0, d_lit32, d_out, d_cyc); hipEventRecord(e1);
        hipDeviceSynchronize();
        float ms = 0; hipEventElapsedTime(&ms, e0, e1);
        hipMemcpy(out, d_out, 256, hipMemcpyDeviceToHost); hipMemcpy(&cyc, d_cyc, 8, hipMemcpyDeviceToHost);
        int bad = 0; for (int i = 0; i < 21; i++) bad += out[i + 1] != ref[i];
        printf("%-84s %6.2f ns per literal  %5.1f cycles  table %s\n", kk.name, ms * 1e6 / NLIT, (double)cyc / NLIT, bad ? "WRONG" : "ok");
    }
    struct { const char* name; void (*k)(const unsigned char*, unsigned*, unsigned long long*); int shift; } ks[] = {
        {"G7N (G6D + s_nop 0 behind the late branch)", k_g7n, 1}, {"G7M (G6D + s_nop 0 in front of the late branch)", k_g7m, 1},
        {"F5  (5 VALU + s_nop, test-free)", k_f5, 0}, {"G4N (cmp, nop, dpp sel, s_ashr, sdwa sel)", k_g4n, 0}, {"G4  (the same, no s_nop)", k_g4, 0},
        {"G6A (G4 + s_andn2 + branch in the step)", k_g6a, 0}, {"G6B (lane 0 off, test on shifted mask, late branch)", k_g6b, 1},
        {"G6C (cmp, dpp, ashr, test, sdwa, branch)", k_g6c, 1}, {"G6D (cmp, branch, dpp, ashr, sdwa, test)", k_g6d, 1}, {"G6E (G6C with s_cselect exec instead of the branch)", k_g6e, 1},
        {"G4X (G4N, EXEC = lanes 0..31)", k_g4x, 0}, {"G4Y (G4N, EXEC = lanes 0..21)", k_g4y, 0}};
    for (auto& kk : ks) {
        unsigned out[64]; unsigned long long cyc = 0;
        hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
        hipLaunchKernelGGL(kk.k, dim3(1), dim3(64), 0, 0, d_lit, d_out, d_cyc);
        hipDeviceSynchronize();
        hipEventRecord(e0); hipLaunchKernelGGL(kk.k, dim3(1), dim3(64), 0, 0, d_lit, d_out, d_cyc); hipEventRecord(e1);
        hipDeviceSynchronize();
        float ms = 0; hipEventElapsedTime(&ms, e0, e1);
        hipMemcpy(out, d_out, 256, hipMemcpyDeviceToHost); hipMemcpy(&cyc, d_cyc, 8, hipMemcpyDeviceToHost);
        int bad = 0; for (int i = 0; i < 21; i++) bad += out[i + kk.shift] != ref[i];
        printf("%-52s %6.2f ns per literal  %5.1f cycles  table %s\n", kk.name, ms * 1e6 / NLIT, (double)cyc / NLIT, bad ? "WRONG" : "ok");
    }
    return 0;
}
