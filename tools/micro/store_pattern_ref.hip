// Microbenchmark for the reference-true sweep shape (40 views, C=256, D=12, 60x80 maps, W=80 = 2.5 x 32): how fast does
// MI355X take the variance WRITE stream alone under different pixel -> block maps?  All variants write every element of
// the (N,C,D,H,W) volume exactly once with the sweep's lane map (lane = 4 consecutive pixels x channel 8*i+g; a
// wave-instruction = 8 channel rows x 128 B of 32 pixels), 12 planes per block, block id -> (slab = id % 8, tile = id / 8).
//   tile16x8   : the shipped choice (64-B runs: 16 px per row)
//   tile32x4   : 32x4 tiles, third tile column half outside (lanes predicated off)
//   flat128    : a block owns 128 CONSECUTIVE flat pixels (1.6 rows): every wave-instruction writes whole 128-B lines
//   lines4     : a block owns 4 vertically stacked 128-B lines (rows r..r+3; the line grid shifts by 16 px on odd rows)
// Build: hipcc -O3 --offload-arch=gfx950 tools/micro/store_pattern_ref.hip -o tools/micro/store_pattern_ref
#include <hip/hip_runtime.h>
#include <cstdio>

typedef float v4f __attribute__((ext_vector_type(4)));

template <int MODE, bool NT>
__global__ __launch_bounds__(256) void store_kernel(float* __restrict__ var, int N, int C, int D, int H, int W, int tiles) {
    const int S = C / 32, HW = H * W;
    int HWp = HW;
    const int id = blockIdx.x;
    const int slab = id % S, bt = id / S;
    const int tile = bt % tiles, n = bt / tiles;
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int g = lane & 7, ps = lane >> 3;
    long flat = -1;   // flat pixel index of the lane's first of 4 consecutive pixels (-1: nothing)
    if (MODE == 0) {          // 16x8
        const int tx = tile % 5, ty = tile / 5;
        const int p0 = 32 * wave + 4 * ps;
        const int px = tx * 16 + p0 % 16, py = ty * 8 + p0 / 16;
        if (py < H) flat = (long)py * W + px;
    } else if (MODE == 1) {   // 32x4, 3 tile columns
        const int tx = tile % 3, ty = tile / 3;
        const int px = tx * 32 + 4 * ps, py = ty * 4 + wave;
        if (px < W && py < H) flat = (long)py * W + px;
    } else if (MODE == 2) {   // flat 128
        const long f = (long)tile * 128 + 32 * wave + 4 * ps;
        if (f < HW) flat = f;
    } else if (MODE == 4 || MODE == 5) {   // 32x4 tiles on rows PITCHED to 96 px (3 lines): MODE 4 real pixels only, MODE 5 pad too
        const int tx = tile % 3, ty = tile / 3;
        const int px = tx * 32 + 4 * ps, py = ty * 4 + wave;
        if ((MODE == 5 || px < W) && py < H) flat = (long)py * 96 + px;
    } else {                  // 4 stacked lines: the 5 lines of a row pair are A r0[0:32] B r0[32:64] C r0[64:80]+r1[0:16] D r1[16:48] E r1[48:80]
        // tile = (row-pair group of 2 pairs = 4 rows, column class k in 0..4): lines of class k in the two row pairs -> only 2 lines
        // per class per 4 rows; use 8 rows: 4 row pairs x class k
        const int k = tile % 5, rp0 = (tile / 5) * 4;
        const int rp = rp0 + wave;      // row pair
        const long f = (long)rp * 160 + 32 * k + 4 * ps;
        if (f < HW) flat = f;
    }
    const v4f vv = {1.0f * id, 2.0f, 3.0f, (float)lane};
    if (flat < 0) return;
    if (MODE == 4 || MODE == 5) HWp = H * 96;
    for (int d = 0; d < D; ++d) {
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            const int c = slab * 32 + 8 * i + g;
            float* dst = var + (((size_t)n * C + c) * D + d) * HWp + flat;
            if (NT) __builtin_nontemporal_store(vv, reinterpret_cast<v4f*>(dst));
            else *reinterpret_cast<v4f*>(dst) = vv;
        }
    }
}

template <typename F>
float time_ms(F&& f, int reps = 10) {
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
    const int N = 40, C = 256, D = 12, H = 60, W = 80;
    const size_t elems = (size_t)N * C * D * H * W;
    float* var;
    if (hipMalloc(&var, elems * 4 * 96 / 80) != hipSuccess) { printf("alloc failed\n"); return 1; }
    const double gb = elems * 4 / 1e9;
#define RUN(MODE, NTV, TILES, NAME)                                                                                         \
    {                                                                                                                       \
        const int blocks = N * (TILES) * (C / 32);                                                                          \
        float ms = time_ms([&] { hipLaunchKernelGGL((store_kernel<MODE, NTV>), dim3(blocks), dim3(256), 0, 0, var, N, C, D, H, W, TILES); }); \
        printf("%-10s nt=%d blocks=%6d : %.3f ms  %.0f GB/s\n", NAME, (int)NTV, blocks, ms, gb / ms * 1e3);                \
    }
    RUN(0, true, 5 * 8, "tile16x8") RUN(0, false, 5 * 8, "tile16x8")
    RUN(1, true, 3 * 15, "tile32x4") RUN(1, false, 3 * 15, "tile32x4")
    RUN(2, true, 38, "flat128") RUN(2, false, 38, "flat128")
    RUN(3, true, 5 * 8, "lines4") RUN(3, false, 5 * 8, "lines4")
    RUN(4, true, 3 * 15, "p96 real") RUN(4, false, 3 * 15, "p96 real")
    RUN(5, true, 3 * 15, "p96 +pad") RUN(5, false, 3 * 15, "p96 +pad")
    return 0;
}
