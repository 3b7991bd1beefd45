#include <hip/hip_runtime.h>
#include <cstdio>
__global__ void k(unsigned* out, unsigned start) {
    unsigned r = 0;
    asm volatile(
        "s_getpc_b64 s[98:99]\n\t"
        "7:\n\t"
        "s_lshl_b32 s97, %[st], 2\n\t"
        "s_add_u32 s98, s98, s97\n\t"
        "s_addc_u32 s99, s99, 0\n\t"
        "s_add_u32 s98, s98, 6f-7b\n\t"
        "s_addc_u32 s99, s99, 0\n\t"
        "s_setpc_b64 s[98:99]\n\t"
        "6:\n\t"
        "s_branch 20f\n\t"
        "s_branch 21f\n\t"
        "s_branch 22f\n\t"
        "20: v_add_u32 %[r], 100, %[r]\n\t"
        "21: v_add_u32 %[r], 10, %[r]\n\t"
        "22: v_add_u32 %[r], 1, %[r]\n\t"
        : [r] "+v"(r) : [st] "s"(start) : "s97", "s98", "s99", "scc");
    out[threadIdx.x] = r;
}
int main() {
    unsigned* d; hipMalloc(&d, 256);
    for (unsigned s = 0; s < 3; s++) { k<<<1, 64>>>(d, s); unsigned h; hipMemcpy(&h, d, 4, hipMemcpyDeviceToHost); printf("start %u -> %u\n", s, h); }
    return 0;
}
