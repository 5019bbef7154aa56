// Which SIMD does wave w of a 12-wave (768-thread) workgroup land on?  HW_REG_HW_ID (register 4): bits 5:4 = SIMD id,
// 11:8 = CU id, 15:13 = SE id.   hipcc --offload-arch=gfx950 -O2 tools/micro/wave_simd.hip -o wave_simd && ./wave_simd
#include <hip/hip_runtime.h>
#include <cstdio>
__global__ void k(unsigned* out) {
    unsigned id = __builtin_amdgcn_s_getreg((4 /*HW_REG_HW_ID*/) | (0 << 6) | (31 << 11));
    if ((threadIdx.x & 63) == 0) out[blockIdx.x * 16 + (threadIdx.x >> 6)] = id;
}
int main() {
    unsigned* d;
    hipMalloc(&d, 4 * 16 * 4);
    for (int threads : {768, 576, 512}) {
        hipMemset(d, 0, 4 * 16 * 4);
        hipLaunchKernelGGL(k, dim3(4), dim3(threads), 0, 0, d);
        unsigned h[64];
        hipMemcpy(h, d, sizeof(h), hipMemcpyDeviceToHost);
        for (int b = 0; b < 4; ++b) {
            printf("threads %d block %d: simd of waves:", threads, b);
            for (int w = 0; w < threads / 64; ++w) printf(" %u", (h[b * 16 + w] >> 4) & 3);
            printf("   (cu %u se %u)\n", (h[b * 16] >> 8) & 15, (h[b * 16] >> 13) & 7);
        }
    }
    return 0;
}
