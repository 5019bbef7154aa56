// Microbenchmark: the plane sweep's write pattern (store_pattern.hip, 32x4 tiles, 76 KB of LDS per block = 2 blocks per CU,
// 400 VALU per plane and wave in between) under every cache-policy modifier of global_store_dwordx4 on gfx950.
// Build: hipcc -O3 --offload-arch=gfx950 tools/micro/store_policy.hip -o tools/micro/store_policy
#include <hip/hip_runtime.h>
#include <cstdio>

typedef float v4f __attribute__((ext_vector_type(4)));

template <int POL>
__device__ __forceinline__ void store16(float* p, v4f v) {
    if (POL == 0) asm volatile("global_store_dwordx4 %0, %1, off" ::"v"(p), "v"(v) : "memory");
    if (POL == 1) asm volatile("global_store_dwordx4 %0, %1, off nt" ::"v"(p), "v"(v) : "memory");
    if (POL == 2) asm volatile("global_store_dwordx4 %0, %1, off sc0" ::"v"(p), "v"(v) : "memory");
    if (POL == 3) asm volatile("global_store_dwordx4 %0, %1, off sc1" ::"v"(p), "v"(v) : "memory");
    if (POL == 4) asm volatile("global_store_dwordx4 %0, %1, off sc0 sc1" ::"v"(p), "v"(v) : "memory");
    if (POL == 5) asm volatile("global_store_dwordx4 %0, %1, off sc0 nt" ::"v"(p), "v"(v) : "memory");
    if (POL == 6) asm volatile("global_store_dwordx4 %0, %1, off sc1 nt" ::"v"(p), "v"(v) : "memory");
    if (POL == 7) asm volatile("global_store_dwordx4 %0, %1, off sc0 sc1 nt" ::"v"(p), "v"(v) : "memory");
}

template <int POL, int NV>
__global__ __launch_bounds__(256) void store_valu_kernel(float* __restrict__ var, int N, int C, int D, int H, int W, int tiles_x, int tiles, float seed) {
    constexpr int TW = 32, TH = 4;
    const int S = C / 32;
    const int HW = H * W;
    const int id = blockIdx.x;
    const int slab = id % S, bt = id / S;
    const int tile = bt % tiles, n = bt / tiles;
    const int tx0 = (tile % tiles_x) * TW, ty0 = (tile / tiles_x) * TH;
    const int lane = threadIdx.x & 63, wave = (threadIdx.x >> 6) & 3;
    const int g = lane & 7, ps = lane >> 3;
    const int p0 = 32 * wave + 4 * ps;
    const int px0 = tx0 + p0 % TW, py = ty0 + p0 / TW;
    const size_t st_off = (size_t)py * W + px0;
    float acc[16];
#pragma unroll
    for (int k = 0; k < 16; ++k) acc[k] = seed * (float)(lane + k);
    for (int d = 0; d < D; ++d) {
#pragma unroll
        for (int r = 0; r < NV / 16; ++r)
#pragma unroll
            for (int k = 0; k < 16; ++k) acc[k] = fmaf(acc[k], 1.0001f, seed);
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            const int c = slab * 32 + 8 * i + g;
            float* dst = var + (((size_t)n * C + c) * D + d) * HW + st_off;
            const v4f vv = {acc[4 * i], acc[4 * i + 1], acc[4 * i + 2], acc[4 * i + 3]};
            store16<POL>(dst, vv);
        }
    }
}

template <typename F>
float time_ms(F&& f, int reps = 4) {
    hipEvent_t a, b;
    (void)hipEventCreate(&a); (void)hipEventCreate(&b);
    f();
    float best = 1e30f;
    for (int r = 0; r < reps; ++r) {
        (void)hipEventRecord(a);
        f();
        (void)hipEventRecord(b);
        (void)hipEventSynchronize(b);
        float ms;
        (void)hipEventElapsedTime(&ms, a, b);
        best = ms < best ? ms : best;
    }
    return best;
}

int main() {
    const int N = 40, C = 256, D = 64, H = 120, W = 160;
    const size_t elems = (size_t)N * C * D * H * W;
    float* var;
    if (hipMalloc(&var, elems * 4) != hipSuccess) { printf("alloc failed\n"); return 1; }
    const double gb = elems * 4 / 1e9;
    const int tiles_x = W / 32, tiles = tiles_x * (H / 4);
    const int blocks = N * tiles * (C / 32);
    const char* names[8] = {"(none)", "nt", "sc0", "sc1", "sc0 sc1", "sc0 nt", "sc1 nt", "sc0 sc1 nt"};
#define RUN(POL, NV)                                                                                                       \
    {                                                                                                                      \
        (void)hipFuncSetAttribute(reinterpret_cast<const void*>(store_valu_kernel<POL, NV>), hipFuncAttributeMaxDynamicSharedMemorySize, 76 * 1024); \
        float ms = time_ms([&] { hipLaunchKernelGGL((store_valu_kernel<POL, NV>), dim3(blocks), dim3(256), 76 * 1024, 0, var, N, C, D, H, W, tiles_x, tiles, 0.5f); }); \
        printf("%-11s %3d VALU: %.3f ms  %.0f GB/s\n", names[POL], NV, ms, gb / ms * 1e3);                                 \
    }
    RUN(0, 0) RUN(1, 0) RUN(2, 0) RUN(3, 0) RUN(4, 0) RUN(5, 0) RUN(6, 0) RUN(7, 0)
    RUN(0, 400) RUN(1, 400) RUN(2, 400) RUN(3, 400) RUN(4, 400) RUN(5, 400) RUN(6, 400) RUN(7, 400)
    return 0;
}
