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
//
// STRIDE = 2 is the weight gradient of the stride-2 layers (conv1, conv3: X at the fine resolution, dY at the coarse one)
//   dW[o][c][kd][kh][kw] = sum over (n, d, h, w) of dY[n][o][d][h][w] * X[n][c][2d+kd-1][2h+kh-1][2w+kw-1]
// and, with the roles of the two tensors exchanged by the caller, of the transposed layers (conv9, conv11: their dY is the
// fine tensor, their X the coarse one).  Same kernel on a 1 x 4 x 8 tile of dY with a 3 x 9 x 17 halo of X; walking along
// d, two of the three ring planes are new per tile.
#include "common.h"

namespace mvsdet {

typedef float f32x16 __attribute__((ext_vector_type(16)));

constexpr int kDwTapsPerWave = 7;

template <int STRIDE>
struct DwTile {
    static constexpr int TH = 4, TW = STRIDE == 1 ? 16 : 8, Vox = TH * TW;             // dY voxels per tile: 64 / 32
    static constexpr int HH = STRIDE * (TH - 1) + 3, HW = STRIDE * (TW - 1) + 3;       // X halo rows / columns: 6 x 18 / 9 x 17
    static constexpr int Plane = HH * HW, Halo = 3 * Plane;                            // 108 / 153 per plane, three planes
    static constexpr int XStride = Halo | 1;                                           // odd: conflict-free across channels
    static constexpr int YStride = Vox + 1;
};

template <int STRIDE>
__global__ __launch_bounds__(kThreads, 2) void conv3d_k3_dw_mfma_kernel(const float* __restrict__ x, const float* __restrict__ gy,
                                                                      float* __restrict__ partial, int N, int Cin, int Cout,
                                                                      int D, int H, int W, int Do, int Ho, int Wo, int tiles_w,
                                                                      int tiles_h, int ncols, int nsplit) {
    // X halo as a ring of three planes per channel: walking a (view, h-tile, w-tile) column along d, only the NEW
    // planes (one at stride 1, two at stride 2) are loaded per tile instead of three
    using T = DwTile<STRIDE>;
    constexpr int kPlane = T::Plane, kDwTW = T::TW, kDwTH = T::TH, kDwVox = T::Vox, kDwHW = T::HW;
    constexpr int kDwXStride = T::XStride, kDwYStride = T::YStride;
    __shared__ float s_x[32 * kDwXStride];
    __shared__ float s_y[32 * kDwYStride];
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int split = blockIdx.x, c0 = blockIdx.y * 32, o0 = blockIdx.z * 32;
    const size_t plane = (size_t)H * W, vol = (size_t)D * plane;            // X (D, H, W)
    const size_t oplane = (size_t)Ho * Wo, ovol = (size_t)Do * oplane;      // dY (Do, Ho, Wo)
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

    // loads input plane dd (may be outside [0, D): zeros) of the column into ring slot dd mod 3; (h0, w0) = the tile's
    // first dY voxel
    auto load_plane = [&](int n, int h0, int w0, int dd) {
        const int slot = ((dd % 3) + 3) % 3;
        for (int e = tid; e < 32 * kPlane; e += kThreads) {
            const int c = e / kPlane, r = e - c * kPlane;
            const int hy = r / kDwHW, wx = r - hy * kDwHW;
            const int hh = STRIDE * h0 + hy - 1, ww = STRIDE * w0 + wx - 1;
            float v = 0.0f;
            if (c0 + c < Cin && dd >= 0 && dd < D && hh >= 0 && hh < H && ww >= 0 && ww < W)
                v = x[((size_t)n * Cin + c0 + c) * vol + (size_t)dd * plane + (size_t)hh * W + ww];
            s_x[c * kDwXStride + slot * kPlane + r] = v;
        }
    };

    // the NEW planes of X and the dY tile of output plane d, fetched into registers while the MFMAs of plane d-1 run
    // (the ring holds all three planes a tile reads, so they cannot land in LDS before that tile is done)
    constexpr int kNewPlanes = STRIDE;                                  // per tile: plane d+1, or planes 2d and 2d+1
    // X: four lanes per halo row (row = channel * HH + hy), each taking the columns l, l + 4, ...: one division per row pass
    // instead of two per element (the flat element order cost ~30 VALU instructions per value -- as much time as the MFMAs
    // of a stride-2 tile), at 16 rows x 16 bytes per wave-instruction.
    constexpr int kRows = 32 * T::HH, kRowPass = kThreads / 4;          // 192 / 288 rows, 64 per pass
    constexpr int kPasses = (kRows + kRowPass - 1) / kRowPass;          // 3 / 5
    constexpr int kColIter = (kDwHW + 3) / 4;                           // 5
    constexpr int kYIter = 32 * kDwVox / kThreads;                      // 8 / 4
    float px[kNewPlanes][kPasses][kColIter], py[kYIter];
    auto new_plane = [&](int d, int pl) { return STRIDE == 1 ? d + 1 : 2 * d + pl; };
    // 32-bit element offsets from a block-uniform base (the host checks that 32 channels of one view fit)
    auto fetch = [&](int n, int h0, int w0, int d) {
        // the index arithmetic below is the same for every tile: left to itself the compiler hoists all of it out of the
        // loops (60-150 VGPRs of loop invariants, spilled).  An opaque copy of the thread index keeps it here.
        int tid = threadIdx.x;
        asm volatile("" : "+v"(tid));
        const float* xb = x + ((size_t)n * Cin + c0) * vol;
        const float* yb = gy + ((size_t)n * Cout + o0) * ovol + (size_t)d * oplane;
        const int l4 = tid & 3;
#pragma unroll
        for (int p = 0; p < kPasses; ++p) {
            const int row = p * kRowPass + (tid >> 2);
            const int c = row / T::HH, hy = row - c * T::HH;
            const int hh = STRIDE * h0 + hy - 1;
            const bool row_ok = row < kRows && c0 + c < Cin && hh >= 0 && hh < H;
            const int rbase = c * (int)vol + hh * W + STRIDE * w0 - 1;
#pragma unroll
            for (int pl = 0; pl < kNewPlanes; ++pl) {
                const int dd = new_plane(d, pl);
                const float* xp = xb + (size_t)dd * plane;
#pragma unroll
                for (int k = 0; k < kColIter; ++k) {
                    const int wx = l4 + 4 * k, ww = STRIDE * w0 + wx - 1;
                    float v = 0.0f;
                    if (row_ok && dd < D && wx < kDwHW && ww >= 0 && ww < W) v = xp[(unsigned)(rbase + wx)];
                    px[pl][p][k] = v;
                }
            }
        }
#pragma unroll
        for (int it = 0; it < kYIter; ++it) {
            const int e = tid + it * kThreads;
            const int o = e / kDwVox, q = e - o * kDwVox;
            const int hh = h0 + q / kDwTW, ww = w0 + q % kDwTW;
            float v = 0.0f;
            if (o0 + o < Cout && hh < Ho && ww < Wo)
                v = yb[(unsigned)(o * (int)ovol + hh * Wo + ww)];
            py[it] = v;
        }
    };
    auto commit = [&](int d) {
        int tid = threadIdx.x;
        asm volatile("" : "+v"(tid));
        const int l4 = tid & 3;
#pragma unroll
        for (int p = 0; p < kPasses; ++p) {
            const int row = p * kRowPass + (tid >> 2);
            const int c = row / T::HH, hy = row - c * T::HH;
            const int lbase = c * kDwXStride + hy * kDwHW;
#pragma unroll
            for (int pl = 0; pl < kNewPlanes; ++pl) {
                const int slot = new_plane(d, pl) % 3;
#pragma unroll
                for (int k = 0; k < kColIter; ++k) {
                    const int wx = l4 + 4 * k;
                    if (row < kRows && wx < kDwHW) s_x[lbase + slot * kPlane + wx] = px[pl][p][k];
                }
            }
        }
#pragma unroll
        for (int it = 0; it < kYIter; ++it) {
            const int e = tid + it * kThreads;
            const int o = e / kDwVox, q = e - o * kDwVox;
            s_y[o * kDwYStride + q] = py[it];
        }
    };

    const int cols_per_view = tiles_h * tiles_w;
    for (int cidx = split; cidx < ncols; cidx += nsplit) {
        const int n = cidx / cols_per_view, t2 = cidx - n * cols_per_view;
        const int h0 = (t2 / tiles_w) * kDwTH, w0 = (t2 % tiles_w) * kDwTW;
        fetch(n, h0, w0, 0);
        __syncthreads();  // previous column fully consumed
        load_plane(n, h0, w0, -1);
        if (STRIDE == 1) load_plane(n, h0, w0, 0);
        for (int d = 0; d < Do; ++d) {
            if (d > 0) __syncthreads();  // tile d-1 fully consumed: its oldest planes and s_y may be replaced
            commit(d);
            __syncthreads();
            if (d + 1 < Do) fetch(n, h0, w0, d + 1);
            int tapoff[kDwTapsPerWave];
#pragma unroll
            for (int i = 0; i < kDwTapsPerWave; ++i) tapoff[i] = ((STRIDE * d + tap_kd[i] - 1 + 3) % 3) * kPlane + tap_in[i];
            const float* ay = s_y + col * kDwYStride + kk;   // A[i = o][k = voxel parity]
            const float* bx = s_x + col * kDwXStride;        // B[k][j = c]
            // all operand reads of a voxel pair are issued before its MFMAs (no branch in between): the reads of
            // the next pair overlap the matrix instructions of this one.  A wave with 6 taps multiplies a seventh,
            // unused accumulator (1/28 of the work) rather than branch.
#pragma unroll 2
            for (int vp = 0; vp < kDwVox / 2; ++vp) {
                const int q = 2 * vp + kk;                   // this lane's voxel of the pair
                const int base = STRIDE * ((q / kDwTW) * kDwHW + (q % kDwTW));
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

namespace {
int launch_dw(int stride, const float* x, const float* grad_out, float* partial, size_t partial_bytes, int nsplit, int N, int Cin,
              int Cout, int D, int H, int W, mvsdet_stream_t stream) {
    MVS_REQUIRE(x && grad_out && partial, "conv3d_k3_dw: NULL pointer");
    MVS_REQUIRE(N > 0 && Cin > 0 && Cout > 0 && D > 0 && H > 0 && W > 0, "conv3d_k3_dw: bad shape");
    MVS_REQUIRE(nsplit > 0 && nsplit <= 65535, "conv3d_k3_dw: nsplit=%d outside [1,65535]", nsplit);
    MVS_REQUIRE(stride == 1 || (D % 2 == 0 && H % 2 == 0 && W % 2 == 0), "conv3d_k3_s2_dw: D, H, W must be even");
    MVS_REQUIRE((long long)D * H * W * 32 < INT32_MAX, "conv3d_k3_dw: 32 channels of one view exceed 2^31 elements");
    if (partial_bytes < mvsdet_conv3d_k3_dw_partial_bytes(Cin, Cout, nsplit)) {
        set_error("conv3d_k3_dw: partial buffer %zu B < %zu B", partial_bytes, mvsdet_conv3d_k3_dw_partial_bytes(Cin, Cout, nsplit));
        return MVSDET_ERR_WORKSPACE;
    }
    const int Do = D / stride, Ho = H / stride, Wo = W / stride;   // dY
    const int tw = stride == 1 ? DwTile<1>::TW : DwTile<2>::TW, th = DwTile<1>::TH;
    const int tiles_w = (Wo + tw - 1) / tw, tiles_h = (Ho + th - 1) / th;
    const long long ncols = (long long)N * tiles_h * tiles_w;  // (view, h-tile, w-tile) columns, walked along d
    MVS_REQUIRE(ncols < INT32_MAX, "conv3d_k3_dw: too many tiles");
    dim3 grid((unsigned)nsplit, (unsigned)((Cin + 31) / 32), (unsigned)((Cout + 31) / 32));
    MVS_REQUIRE(grid.y <= 65535 && grid.z <= 65535, "conv3d_k3_dw: too many channel blocks");
    if (stride == 1)
        hipLaunchKernelGGL(conv3d_k3_dw_mfma_kernel<1>, grid, dim3(kThreads), 0, (hipStream_t)stream, x, grad_out, partial, N, Cin,
                           Cout, D, H, W, Do, Ho, Wo, tiles_w, tiles_h, (int)ncols, nsplit);
    else
        hipLaunchKernelGGL(conv3d_k3_dw_mfma_kernel<2>, grid, dim3(kThreads), 0, (hipStream_t)stream, x, grad_out, partial, N, Cin,
                           Cout, D, H, W, Do, Ho, Wo, tiles_w, tiles_h, (int)ncols, nsplit);
    MVS_LAUNCH_CHECK("conv3d_k3_dw");
    return MVSDET_OK;
}
}  // namespace

extern "C" int mvsdet_conv3d_k3_dw_mfma_f32(const float* x, const float* grad_out, float* partial, size_t partial_bytes,
                                            int nsplit, int N, int Cin, int Cout, int D, int H, int W,
                                            mvsdet_stream_t stream) {
    return launch_dw(1, x, grad_out, partial, partial_bytes, nsplit, N, Cin, Cout, D, H, W, stream);
}

extern "C" int mvsdet_conv3d_k3_s2_dw_mfma_f32(const float* x, const float* grad_out, float* partial, size_t partial_bytes,
                                               int nsplit, int N, int Cin, int Cout, int D, int H, int W,
                                               mvsdet_stream_t stream) {
    return launch_dw(2, x, grad_out, partial, partial_bytes, nsplit, N, Cin, Cout, D, H, W, stream);
}
