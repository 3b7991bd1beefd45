// Dependent-load latency of one lone wavefront on gfx950 by working-set size (test tool, not product).
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>
#include <numeric>
#include <algorithm>
#include <random>
__global__ void k_chase(const unsigned* tab, unsigned* out, int n, unsigned start) {
    unsigned p = start;
    for (int i = 0; i < n; i++) p = tab[p];                       // all lanes the same address
    out[threadIdx.x] = p;
}
__global__ void k_chase_lanes(const unsigned* tab, unsigned* out, int n, unsigned start, int lanes_active) {
    unsigned p = start;                                           // vector loads (address in VGPRs), one line per hop
    if ((int)threadIdx.x < lanes_active)
        for (int i = 0; i < n; i++) { p = __builtin_nontemporal_load(tab + p + (threadIdx.x & 15)) ; p -= (threadIdx.x & 15) * 0; }
    out[threadIdx.x] = p;
}
// store then dependent load of ANOTHER address: does the load's wait include the store's round trip?
__global__ void k_store_then_load(const unsigned* tab, unsigned* scratch, unsigned* out, int n) {
    unsigned p = 0;
    for (int i = 0; i < n; i++) { scratch[threadIdx.x + 64 * (i & 1023)] = p; p = tab[p]; }
    out[threadIdx.x] = p;
}
int main() {
    setvbuf(stdout, NULL, _IONBF, 0);
    unsigned* out; (void)hipMalloc(&out, 4096);
    unsigned* scratch; (void)hipMalloc(&scratch, 1 << 20);
    std::mt19937 rng(1);
    hipEvent_t a, b; (void)hipEventCreate(&a); (void)hipEventCreate(&b);
    for (size_t bytes : {4096ul, 16384ul, 65536ul, 1ul << 20, 3ul << 20, 16ul << 20, 128ul << 20, 1ul << 30}) {
        const size_t stride = 64;                                  // one entry per 256 B: no two hops share a line
        const size_t ents = bytes / 4, slots = ents / stride;
        std::vector<unsigned> perm(slots); std::iota(perm.begin(), perm.end(), 0u); std::shuffle(perm.begin(), perm.end(), rng);
        std::vector<unsigned> h(ents, 0);
        for (size_t i = 0; i < slots; i++) for (int k = 0; k < 16; k++) h[(size_t)perm[i] * stride + k] = perm[(i + 1) % slots] * stride;   // one cycle through all slots
        unsigned* tab; (void)hipMalloc(&tab, bytes);
        (void)hipMemcpy(tab, h.data(), bytes, hipMemcpyHostToDevice);
        const int n = 20000;
        float best = 1e30f, bestl = 1e30f, bests = 1e30f;
        for (int r = 0; r < 3; r++) {
            float ms;
            (void)hipEventRecord(a); k_chase<<<1, 64>>>(tab, out, n, perm[0] * stride); (void)hipEventRecord(b); (void)hipEventSynchronize(b);
            (void)hipEventElapsedTime(&ms, a, b); best = std::min(best, ms);
            (void)hipEventRecord(a); k_store_then_load<<<1, 64>>>(tab, scratch, out, n); (void)hipEventRecord(b); (void)hipEventSynchronize(b);
            (void)hipEventElapsedTime(&ms, a, b); bests = std::min(bests, ms);
        }
        float bestv = 1e30f;
        for (int r = 0; r < 3; r++) {
            float ms;
            (void)hipEventRecord(a); k_chase_lanes<<<1, 64>>>(tab, out, n, perm[0] * stride, 64); (void)hipEventRecord(b); (void)hipEventSynchronize(b);
            (void)hipEventElapsedTime(&ms, a, b); bestv = std::min(bestv, ms);
        }
        printf("[vector load %7.1f ns] ", bestv * 1e6 / n);
        printf("working set %8zu KiB (%6zu lines): dependent load %7.1f ns   with a store before each load %7.1f ns\n",
               bytes >> 10, slots, best * 1e6 / n, bests * 1e6 / n);
        (void)hipFree(tab);
    }
    return 0;
}
