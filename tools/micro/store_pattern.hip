// Microbenchmark: how fast does MI355X take the plane sweep's WRITE pattern alone (no taps, no LDS)?
// Same block -> address map as plane_sweep_variance_kernel: block = (view, TWxTH tile of 128 px, 32-channel slab),
// loops over D planes; per plane each wave stores 4 x (8 channel rows x 16 B per lane).  Variants: tile width
// (contiguous bytes per channel row = 4*TW), non-temporal or plain stores, blocks per CU.
// Build: hipcc -O3 --offload-arch=gfx950 tools/micro/store_pattern.hip -o /tmp/store_pattern
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>

typedef float v4f __attribute__((ext_vector_type(4)));

// the same stores with NV dependent-free VALU instructions per plane and wave in between (how well do stores and
// arithmetic overlap?)
template <int NV, int WPB>
__global__ __launch_bounds__(WPB * 64) void store_valu_kernel(float* __restrict__ var, int N, int C, int D, int H, int W, int tiles_x, int tiles, float seed) {
    constexpr int TW = 32, TH = 4;
    const int S = C / 32;
    const int HW = H * W;
    const int id = blockIdx.x;
    const int slab = id % S, bt = id / S;
    const int tile = bt % tiles, n = bt / tiles;
    const int tx0 = (tile % tiles_x) * TW, ty0 = (tile / tiles_x) * TH;
    const int lane = threadIdx.x & 63, wave = (threadIdx.x >> 6) & 3, grp = threadIdx.x >> 8;
    const int g = lane & 7, ps = lane >> 3;
    const int p0 = 32 * wave + 4 * ps;
    const int px0 = tx0 + p0 % TW, py = ty0 + p0 / TW;
    const size_t st_off = (size_t)py * W + px0;
    float acc[16];
#pragma unroll
    for (int k = 0; k < 16; ++k) acc[k] = seed * (float)(lane + k);
    for (int d = grp; d < D; d += WPB / 4) {
#pragma unroll
        for (int r = 0; r < NV / 16; ++r)
#pragma unroll
            for (int k = 0; k < 16; ++k) acc[k] = fmaf(acc[k], 1.0001f, seed);
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            const int c = slab * 32 + 8 * i + g;
            float* dst = var + (((size_t)n * C + c) * D + d) * HW + st_off;
            const v4f vv = {acc[4 * i], acc[4 * i + 1], acc[4 * i + 2], acc[4 * i + 3]};
            __builtin_nontemporal_store(vv, reinterpret_cast<v4f*>(dst));
        }
    }
}

template <int TW, bool NT, int ORDER>
__global__ __launch_bounds__(256) void store_kernel(float* __restrict__ var, int N, int C, int D, int H, int W, int tiles_x, int tiles) {
    constexpr int TH = 128 / TW;
    const int S = C / 32;
    const int HW = H * W;
    const int id = blockIdx.x;
    const int slab = id % S, bt = id / S;
    const int tile = bt % tiles, n = bt / tiles;
    const int tx0 = (tile % tiles_x) * TW, ty0 = (tile / tiles_x) * TH;
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int g = lane & 7, ps = lane >> 3;
    const int p0 = 32 * wave + 4 * ps;
    const int px0 = tx0 + p0 % TW, py = ty0 + p0 / TW;
    const size_t st_off = (size_t)py * W + px0;
    const v4f vv = {1.0f * id, 2.0f, 3.0f, (float)lane};
    for (int d = 0; d < D; ++d) {
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            // ORDER 0: channel 8*i + g (the sweep's packing: 8 rows per instruction); ORDER 1: channel 4*g + i
            const int c = slab * 32 + (ORDER == 0 ? 8 * i + g : 4 * g + i);
            float* dst = var + (((size_t)n * C + c) * D + d) * HW + st_off;
            if (NT) __builtin_nontemporal_store(vv, reinterpret_cast<v4f*>(dst));
            else *reinterpret_cast<v4f*>(dst) = vv;
        }
    }
}

// lanes = 64 consecutive pixels x 1 channel per instruction (256 B ... 1 KiB contiguous): the friendliest pattern
template <bool NT>
__global__ __launch_bounds__(256) void store_rows_kernel(float* __restrict__ var, size_t n4) {
    const size_t stride = (size_t)gridDim.x * 256;
    const v4f vv = {1.0f, 2.0f, 3.0f, 4.0f};
    for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < n4; i += stride) {
        if (NT) __builtin_nontemporal_store(vv, reinterpret_cast<v4f*>(var) + i);
        else reinterpret_cast<v4f*>(var)[i] = vv;
    }
}

template <typename F>
float time_ms(F&& f, int reps = 5) {
    hipEvent_t a, b;
    hipEventCreate(&a); hipEventCreate(&b);
    f();
    float best = 1e30f;
    for (int r = 0; r < reps; ++r) {
        hipEventRecord(a);
        f();
        hipEventRecord(b);
        hipEventSynchronize(b);
        float ms;
        hipEventElapsedTime(&ms, a, b);
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
#define RUN(TWV, NTV, ORD)                                                                                     \
    {                                                                                                          \
        const int tiles_x = W / TWV, tiles = tiles_x * (H / (128 / TWV));                                      \
        const int blocks = N * tiles * (C / 32);                                                               \
        float ms = time_ms([&] { hipLaunchKernelGGL((store_kernel<TWV, NTV, ORD>), dim3(blocks), dim3(256), 0, 0, var, N, C, D, H, W, tiles_x, tiles); }); \
        printf("tile %3dx%d  nt=%d order=%d : %.3f ms  %.0f GB/s\n", TWV, 128 / TWV, (int)NTV, ORD, ms, gb / ms * 1e3); \
    }
    RUN(32, true, 0) RUN(32, false, 0) RUN(16, true, 0) RUN(32, true, 1)
    RUN(64, true, 0) RUN(128, true, 0) RUN(128, false, 0)
#define RUNVL(NV, WPB, LDSKB)                                                                                  \
    {                                                                                                          \
        const int tiles_x = W / 32, tiles = tiles_x * (H / 4);                                                 \
        const int blocks = N * tiles * (C / 32);                                                               \
        hipFuncSetAttribute(reinterpret_cast<const void*>(store_valu_kernel<NV, WPB>), hipFuncAttributeMaxDynamicSharedMemorySize, LDSKB * 1024); \
        float ms = time_ms([&] { hipLaunchKernelGGL((store_valu_kernel<NV, WPB>), dim3(blocks), dim3(WPB * 64), LDSKB * 1024, 0, var, N, C, D, H, W, tiles_x, tiles, 0.5f); }); \
        printf("32x4 stores + %4d VALU/plane/wave, %d waves/block, %d KB LDS/block (%d blocks/CU): %.3f ms  %.0f GB/s\n", NV, WPB, LDSKB, 160 / LDSKB, ms, gb / ms * 1e3); \
    }
#define RUNV(NV, WPB)                                                                                          \
    {                                                                                                          \
        const int tiles_x = W / 32, tiles = tiles_x * (H / 4);                                                 \
        const int blocks = N * tiles * (C / 32);                                                               \
        float ms = time_ms([&] { hipLaunchKernelGGL((store_valu_kernel<NV, WPB>), dim3(blocks), dim3(WPB * 64), 0, 0, var, N, C, D, H, W, tiles_x, tiles, 0.5f); }); \
        printf("32x4 stores + %4d VALU/plane/wave, %d waves/block: %.3f ms  %.0f GB/s\n", NV, WPB, ms, gb / ms * 1e3); \
    }
    RUNVL(0, 4, 76) RUNVL(400, 4, 76) RUNVL(0, 8, 76) RUNVL(400, 8, 76) RUNVL(400, 4, 50) RUNVL(400, 4, 38) RUNVL(400, 8, 38) RUNVL(0, 4, 38)
    RUNV(0, 4) RUNV(128, 4) RUNV(256, 4) RUNV(400, 4) RUNV(512, 4) RUNV(800, 4) RUNV(400, 8) RUNV(800, 8)
    {
        float ms = time_ms([&] { hipLaunchKernelGGL((store_rows_kernel<true>), dim3(256 * 8), dim3(256), 0, 0, var, elems / 4); });
        printf("linear fill nt=1: %.3f ms  %.0f GB/s\n", ms, gb / ms * 1e3);
        ms = time_ms([&] { hipLaunchKernelGGL((store_rows_kernel<false>), dim3(256 * 8), dim3(256), 0, 0, var, elems / 4); });
        printf("linear fill nt=0: %.3f ms  %.0f GB/s\n", ms, gb / ms * 1e3);
    }
    return 0;
}
