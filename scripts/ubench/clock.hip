// Shader clock seen by one lone wavefront on gfx950, alone and beside a background load (test tool, not product).
//   s_memtime = shader cycles, s_memrealtime = constant 100 MHz  ->  effective clock, cycles per instruction.
// build: hipcc --offload-arch=gfx950 -O2 -o scripts/ubench/clock scripts/ubench/clock.hip
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#define ITER 100000
#define REP4(x) x x x x
#define REP16(x) REP4(x) REP4(x) REP4(x) REP4(x)

__global__ void k_probe(unsigned long long* out, int kind) {
    unsigned a = threadIdx.x, b = 3, c = 5, d = 7, e = 9;
    unsigned long long t0, t1, r0, r1;
    asm volatile("s_memtime %0\n\ts_memrealtime %1\n\ts_waitcnt lgkmcnt(0)" : "=s"(t0), "=s"(r0));
    if (kind == 0) {
        for (int i = 0; i < ITER; i++) { REP16(asm volatile("v_add_u32 %0, %0, %1" : "+v"(a) : "v"(b));) }
    } else if (kind == 1) {
        for (int i = 0; i < ITER; i++) {
            REP4(asm volatile("v_add_u32 %0, %0, %4\n\tv_add_u32 %1, %1, %4\n\tv_add_u32 %2, %2, %4\n\tv_add_u32 %3, %3, %4"
                              : "+v"(a), "+v"(c), "+v"(d), "+v"(e) : "v"(b));)
        }
    } else {
        unsigned s = 1;
        for (int i = 0; i < ITER; i++) { REP16(asm volatile("s_add_u32 %0, %0, 3" : "+s"(s) : : "scc");) }
        a += s;
    }
    asm volatile("s_memtime %0\n\ts_memrealtime %1\n\ts_waitcnt lgkmcnt(0)" : "=s"(t1), "=s"(r1));
    if (threadIdx.x == 0) { out[0] = t1 - t0; out[1] = r1 - r0; out[2] = a + c + d + e; }
}

// background: `waves` wavefronts per workgroup spin until *stop; kind 0 = VALU busy, 1 = s_sleep
__global__ void k_bg(volatile int* stop, int kind, float* sink) {
    float x = threadIdx.x;
    while (!*stop) {
        if (kind == 0) { for (int i = 0; i < 256; i++) x = x * 1.0001f + 0.5f; }
        else __builtin_amdgcn_s_sleep(64);
    }
    if (x == 12345.f) sink[0] = x;
}

int main() {
    unsigned long long* d; hipMalloc(&d, 64);
    int* stop; hipHostMalloc(&stop, 4); float* sink; hipMalloc(&sink, 4);
    hipStream_t s1, s2; hipStreamCreateWithFlags(&s1, hipStreamNonBlocking); hipStreamCreateWithFlags(&s2, hipStreamNonBlocking);
    setvbuf(stdout, nullptr, _IOLBF, 0);
    unsigned long long* h; hipHostMalloc(&h, 64);
    const char* names[3] = {"valu_dep", "valu_ind", "salu_dep"};
    for (int bg = 0; bg < 4; bg++) {
        // bg 0: none; 1: 255 WGs x 256 threads VALU busy; 2: 255 WGs s_sleep; 3: 1020 WGs x 64 VALU busy
        *stop = 0;
        if (bg == 1) hipLaunchKernelGGL(k_bg, dim3(255), dim3(256), 0, s2, stop, 0, sink);
        if (bg == 2) hipLaunchKernelGGL(k_bg, dim3(255), dim3(256), 0, s2, stop, 1, sink);
        if (bg == 3) hipLaunchKernelGGL(k_bg, dim3(1020), dim3(64), 0, s2, stop, 0, sink);
        for (int kind = 0; kind < 3; kind++) {
            for (int rep = 0; rep < 3; rep++) {
                hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
                hipEventRecord(e0, s1);
                hipLaunchKernelGGL(k_probe, dim3(1), dim3(64), 0, s1, d, kind);
                hipEventRecord(e1, s1);
                hipStreamSynchronize(s1);
                float ms; hipEventElapsedTime(&ms, e0, e1);
                hipMemcpyAsync(h, d, 24, hipMemcpyDeviceToHost, s1); hipStreamSynchronize(s1);
                double n = 16.0 * ITER;
                printf("bg %d %-9s rep %d: %8.3f ms wall, %llu cyc, %llu ticks(100MHz) -> clock %.3f GHz, %.2f cyc/instr, %.2f ns/instr\n",
                       bg, names[kind], rep, ms, h[0], h[1], h[0] / (h[1] * 10.0), h[0] / n, h[1] * 10.0 / n);
            }
        }
        *stop = 1;
        hipStreamSynchronize(s2);
    }
    return 0;
}
