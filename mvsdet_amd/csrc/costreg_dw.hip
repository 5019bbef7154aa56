// Weight gradient of the stride-1 3x3x3 convolutions of the cost regularisation network (conv0 256->64 above all:
// MIOpen needs 360 ms for it at the reference-true shape, 42 % of a whole training step of the network):
//
//   dW[o][c][kd][kh][kw] = sum over (n, d, h, w) of dY[n][o][d][h][w] * X[n][c][d+kd-1][h+kh-1][w+kw-1]
//
// As a GEMM the reduction runs over the voxels: D[o][c] += A[o][k] * B[k][c] with k = voxel, one 32x32 accumulator per
// tap, on v_mfma_f32_32x32x2_f32 (two voxels per instruction; exact fp32 FMA sums).
//   block  = (32 input channels, 32 output channels, one of S voxel splits); wave w owns the taps w, w+4, ... (7,7,7,6)
//            = 7 accumulators (112 VGPRs)
//   tile   = 1 x 4 x 16 voxels: dY[32][64] and the X halo [32][3 x 6 x 18] staged in LDS (odd row strides: the operand
//            reads vary the channel across lanes), 32 voxel pairs x 7 taps = 224 MFMAs per wave and tile; a block walks
//            (view, h-tile, w-tile) columns along d with the halo as a ring of three planes, so each tile loads one new
//            plane of X instead of three
//   output = partial[s][o][c][27] per split, summed by the caller (deterministic, no atomics)
// Bound: fp32 MFMA (same 2.04 TFLOP as the forward of conv0).
#include "common.h"

namespace mvsdet {

typedef float f32x16 __attribute__((ext_vector_type(16)));

constexpr int kDwTH = 4, kDwTW = 16, kDwVox = kDwTH * kDwTW;        // 64 voxels per tile
constexpr int kDwHD = 3, kDwHH = kDwTH + 2, kDwHW = kDwTW + 2;      // halo 3 x 6 x 18
constexpr int kDwHalo = kDwHD * kDwHH * kDwHW;                      // 324
constexpr int kDwXStride = kDwHalo + 1;                             // 325: odd, conflict-free across channels
constexpr int kDwYStride = kDwVox + 1;                              // 65
constexpr int kDwTapsPerWave = 7;

__global__ __launch_bounds__(kThreads, 2) void conv3d_k3_dw_mfma_kernel(const float* __restrict__ x, const float* __restrict__ gy,
                                                                      float* __restrict__ partial, int N, int Cin, int Cout,
                                                                      int D, int H, int W, int tiles_w, int tiles_h,
                                                                      int ncols, int nsplit) {
    // X halo as a ring of three planes per channel: walking a (view, h-tile, w-tile) column along d, only ONE new
    // plane (6 x 18 values per channel) is loaded per tile instead of three
    constexpr int kPlane = kDwHH * kDwHW;  // 108
    __shared__ float s_x[32 * kDwXStride];
    __shared__ float s_y[32 * kDwYStride];
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int split = blockIdx.x, c0 = blockIdx.y * 32, o0 = blockIdx.z * 32;
    const size_t plane = (size_t)H * W, vol = (size_t)D * plane;
    const int col = lane & 31, kk = lane >> 5;

    // this wave's taps: (kd, in-plane offset)
    int tap_kd[kDwTapsPerWave], tap_in[kDwTapsPerWave];
#pragma unroll
    for (int i = 0; i < kDwTapsPerWave; ++i) {
        const int t = wave + 4 * i < 27 ? wave + 4 * i : wave;   // the 28th slot repeats a valid tap; its result is dropped
        tap_kd[i] = t / 9;
        tap_in[i] = ((t / 3) % 3) * kDwHW + (t % 3);
    }

    f32x16 acc[kDwTapsPerWave];
#pragma unroll
    for (int i = 0; i < kDwTapsPerWave; ++i)
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[i][r] = 0.0f;

    // loads input plane dd (may be outside [0, D): zeros) of the column into ring slot dd mod 3
    auto load_plane = [&](int n, int h0, int w0, int dd) {
        const int slot = ((dd % 3) + 3) % 3;
        for (int e = tid; e < 32 * kPlane; e += kThreads) {
            const int c = e / kPlane, r = e - c * kPlane;
            const int hy = r / kDwHW, wx = r - hy * kDwHW;
            const int hh = h0 + hy - 1, ww = w0 + wx - 1;
            float v = 0.0f;
            if (c0 + c < Cin && dd >= 0 && dd < D && hh >= 0 && hh < H && ww >= 0 && ww < W)
                v = x[((size_t)n * Cin + c0 + c) * vol + (size_t)dd * plane + (size_t)hh * W + ww];
            s_x[c * kDwXStride + slot * kPlane + r] = v;
        }
    };

    const int cols_per_view = tiles_h * tiles_w;
    for (int cidx = split; cidx < ncols; cidx += nsplit) {
        const int n = cidx / cols_per_view, t2 = cidx - n * cols_per_view;
        const int h0 = (t2 / tiles_w) * kDwTH, w0 = (t2 % tiles_w) * kDwTW;
        __syncthreads();  // previous column fully consumed
        load_plane(n, h0, w0, -1);
        load_plane(n, h0, w0, 0);
        for (int d = 0; d < D; ++d) {
            if (d > 0) __syncthreads();  // tile d-1 fully consumed: its oldest plane and s_y may be replaced
            load_plane(n, h0, w0, d + 1);
            for (int e = tid; e < 32 * kDwVox; e += kThreads) {
                const int o = e / kDwVox, q = e - o * kDwVox;
                const int hh = h0 + q / kDwTW, ww = w0 + q % kDwTW;
                float v = 0.0f;
                if (o0 + o < Cout && hh < H && ww < W)
                    v = gy[((size_t)n * Cout + o0 + o) * vol + (size_t)d * plane + (size_t)hh * W + ww];
                s_y[o * kDwYStride + q] = v;
            }
            __syncthreads();
            int tapoff[kDwTapsPerWave];
#pragma unroll
            for (int i = 0; i < kDwTapsPerWave; ++i) tapoff[i] = ((d + tap_kd[i] - 1 + 3) % 3) * kPlane + tap_in[i];
            const float* ay = s_y + col * kDwYStride + kk;   // A[i = o][k = voxel parity]
            const float* bx = s_x + col * kDwXStride;        // B[k][j = c]
            // all operand reads of a voxel pair are issued before its MFMAs (no branch in between): the reads of
            // the next pair overlap the matrix instructions of this one.  A wave with 6 taps multiplies a seventh,
            // unused accumulator (1/28 of the work) rather than branch.
#pragma unroll 2
            for (int vp = 0; vp < kDwVox / 2; ++vp) {
                const int q = 2 * vp + kk;                   // this lane's voxel of the pair
                const int base = (q / kDwTW) * kDwHW + (q % kDwTW);
                const float a = ay[2 * vp];
                float b[kDwTapsPerWave];
#pragma unroll
                for (int i = 0; i < kDwTapsPerWave; ++i) b[i] = bx[base + tapoff[i]];
#pragma unroll
                for (int i = 0; i < kDwTapsPerWave; ++i) acc[i] = __builtin_amdgcn_mfma_f32_32x32x2f32(a, b[i], acc[i], 0, 0, 0);
            }
        }
    }
    // partial[split][o][c][tap]; C/D map: column = lane & 31 (c), row = (reg & 3) + 8*(reg >> 2) + 4*(lane >> 5) (o)
#pragma unroll
    for (int i = 0; i < kDwTapsPerWave; ++i) {
        const int t = wave + 4 * i;
        if (t >= 27) continue;
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            const int o = o0 + (r & 3) + 8 * (r >> 2) + 4 * kk, c = c0 + col;
            if (o < Cout && c < Cin) partial[(((size_t)split * Cout + o) * Cin + c) * 27 + t] = acc[i][r];
        }
    }
}

}  // namespace mvsdet

using namespace mvsdet;

extern "C" size_t mvsdet_conv3d_k3_dw_partial_bytes(int Cin, int Cout, int nsplit) {
    if (Cin <= 0 || Cout <= 0 || nsplit <= 0) return 0;
    return (size_t)nsplit * Cout * Cin * 27 * sizeof(float);
}

extern "C" int mvsdet_conv3d_k3_dw_mfma_f32(const float* x, const float* grad_out, float* partial, size_t partial_bytes,
                                            int nsplit, int N, int Cin, int Cout, int D, int H, int W,
                                            mvsdet_stream_t stream) {
    MVS_REQUIRE(x && grad_out && partial, "conv3d_k3_dw: NULL pointer");
    MVS_REQUIRE(N > 0 && Cin > 0 && Cout > 0 && D > 0 && H > 0 && W > 0, "conv3d_k3_dw: bad shape");
    MVS_REQUIRE(nsplit > 0 && nsplit <= 65535, "conv3d_k3_dw: nsplit=%d outside [1,65535]", nsplit);
    if (partial_bytes < mvsdet_conv3d_k3_dw_partial_bytes(Cin, Cout, nsplit)) {
        set_error("conv3d_k3_dw: partial buffer %zu B < %zu B", partial_bytes, mvsdet_conv3d_k3_dw_partial_bytes(Cin, Cout, nsplit));
        return MVSDET_ERR_WORKSPACE;
    }
    const int tiles_w = (W + kDwTW - 1) / kDwTW, tiles_h = (H + kDwTH - 1) / kDwTH;
    const long long ncols = (long long)N * tiles_h * tiles_w;  // (view, h-tile, w-tile) columns, walked along d
    MVS_REQUIRE(ncols < INT32_MAX, "conv3d_k3_dw: too many tiles");
    dim3 grid((unsigned)nsplit, (unsigned)((Cin + 31) / 32), (unsigned)((Cout + 31) / 32));
    MVS_REQUIRE(grid.y <= 65535 && grid.z <= 65535, "conv3d_k3_dw: too many channel blocks");
    hipLaunchKernelGGL(conv3d_k3_dw_mfma_kernel, grid, dim3(kThreads), 0, (hipStream_t)stream, x, grad_out, partial, N, Cin,
                       Cout, D, H, W, tiles_w, tiles_h, (int)ncols, nsplit);
    MVS_LAUNCH_CHECK("conv3d_k3_dw");
    return MVSDET_OK;
}
