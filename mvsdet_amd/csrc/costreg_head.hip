// Last layer of the cost regularisation network (mvs_models/mvsnet.py:102,112): Conv3d(Cin -> 2, kernel 3, padding 1,
// bias) on the full-resolution (N,Cin,D,H,W) feature volume -> (N,2,D,H,W) {cost logits, offset logits}.
//
// Two output channels make a poor GEMM (MIOpen: 8.2 ms for 8 GFLOP at the reference-true shape) but a trivial
// streaming kernel: every input value is read once per tile (+ halo) and used for 54 fused multiply-adds.
//
//   block  = (view n, TD x TH x TW = 4 x 8 x 32 output voxels); thread = one (h, w) column of TD = 4 voxels
//   loop over input channels in chunks of CC = 4: the chunk's halo tile (6 x 10 x 34 floats per channel) is staged
//   in LDS with zero padding; per channel and (kh, kw) a thread reads its 6 values along d once and feeds the
//   3 kd x 4 voxels x 2 outputs = 24 FMAs; the 54 weights of a channel are wave-uniform scalar loads.
//   The next chunk's halo values are fetched into registers while the current chunk is computed.
//   Bound: VALU issue (4 FMA per LDS dword read; 8 GFMA = 0.2-0.25 ms of fp32 FMA issue) + the 0.6 GB input stream
//   (x2 with halos, from L2).  Measured 0.44 ms at (40,64,12,60,80) against 8.2 ms for MIOpen's convolution.
//
// Accumulation order: channels ascending, then kh, kw, kd -- fp32 FMA chain; differs from MIOpen's / ATen's order by
// rounding only (tests compare against ATen-CPU conv3d within 1e-5 of the output scale).
#include "common.h"

namespace mvsdet {

constexpr int kTD = 4, kTH = 8, kTW = 32, kCC = 4;
constexpr int kHD = kTD + 2, kHH = kTH + 2, kHW = kTW + 2;  // halo tile 6 x 10 x 34
constexpr int kHaloVox = kHD * kHH * kHW;                  // 2040 floats per channel

__global__ __launch_bounds__(kThreads) void conv3d_k3_cout2_kernel(const float* __restrict__ x, const float* __restrict__ wgt,
                                                                    const float* __restrict__ bias, float* __restrict__ out,
                                                                    int Cin, int D, int H, int W, int tiles_w, int tiles_h) {
    __shared__ float s_in[kCC][kHaloVox];
    const int tid = threadIdx.x;
    const int tw = tid % kTW, th = tid / kTW;  // 32 x 8 threads
    const int bw = blockIdx.x % tiles_w, bh = blockIdx.x / tiles_w;
    const int w0 = bw * kTW, h0 = bh * kTH, d0 = blockIdx.y * kTD, n = blockIdx.z;
    const size_t plane = (size_t)H * W, vol = (size_t)D * plane;
    const float* xn = x + (size_t)n * Cin * vol;

    float acc[2][kTD];
#pragma unroll
    for (int o = 0; o < 2; ++o)
#pragma unroll
        for (int t = 0; t < kTD; ++t) acc[o][t] = bias ? bias[o] : 0.0f;

    // staging plan of this thread, the same for every channel: halo elements r = tid + 256*k -> offset inside a
    // channel volume (or -1 outside the volume: zero padding)
    constexpr int kStage = (kHaloVox + kThreads - 1) / kThreads;  // 8
    int s_off[kStage];
#pragma unroll
    for (int k = 0; k < kStage; ++k) {
        const int r = tid + k * kThreads;
        const int dz = r / (kHH * kHW), r2 = r - dz * (kHH * kHW);
        const int hy = r2 / kHW, wx = r2 - hy * kHW;
        const int d = d0 + dz - 1, h = h0 + hy - 1, w = w0 + wx - 1;
        const bool ok = r < kHaloVox && d >= 0 && d < D && h >= 0 && h < H && w >= 0 && w < W;
        s_off[k] = ok ? (int)((size_t)d * plane + (size_t)h * W + w) : -1;
    }

    // software pipeline: the next chunk's halo values travel global -> registers while this chunk is computed
    float nxt[kCC][kStage];
    auto fetch = [&](int c0) {
#pragma unroll
        for (int cc = 0; cc < kCC; ++cc) {
            const float* xc = xn + (size_t)(c0 + cc) * vol;
            const bool cok = c0 + cc < Cin;
#pragma unroll
            for (int k = 0; k < kStage; ++k) nxt[cc][k] = (cok && s_off[k] >= 0) ? xc[s_off[k]] : 0.0f;
        }
    };
    fetch(0);
    for (int c0 = 0; c0 < Cin; c0 += kCC) {
        __syncthreads();  // previous chunk fully consumed
#pragma unroll
        for (int cc = 0; cc < kCC; ++cc)
#pragma unroll
            for (int k = 0; k < kStage; ++k) {
                const int r = tid + k * kThreads;
                if (r < kHaloVox) s_in[cc][r] = nxt[cc][k];
            }
        __syncthreads();
        if (c0 + kCC < Cin) fetch(c0 + kCC);
#pragma unroll
        for (int cc = 0; cc < kCC; ++cc) {
            const int c = c0 + cc;
            if (c >= Cin) break;
            const float* w0p = wgt + (size_t)c * 27;                   // weight[0][c][kd][kh][kw]
            const float* w1p = wgt + ((size_t)Cin + c) * 27;           // weight[1][c][...]
#pragma unroll
            for (int kh = 0; kh < 3; ++kh) {
#pragma unroll
                for (int kw = 0; kw < 3; ++kw) {
                    const float* col = &s_in[cc][(th + kh) * kHW + (tw + kw)];
                    float v[kHD];
#pragma unroll
                    for (int dz = 0; dz < kHD; ++dz) v[dz] = col[dz * (kHH * kHW)];
#pragma unroll
                    for (int kd = 0; kd < 3; ++kd) {
                        const float a = w0p[(kd * 3 + kh) * 3 + kw], b = w1p[(kd * 3 + kh) * 3 + kw];  // wave-uniform
#pragma unroll
                        for (int t = 0; t < kTD; ++t) {
                            acc[0][t] = fmaf(v[t + kd], a, acc[0][t]);
                            acc[1][t] = fmaf(v[t + kd], b, acc[1][t]);
                        }
                    }
                }
            }
        }
    }
    const int h = h0 + th, w = w0 + tw;
    if (h < H && w < W) {
#pragma unroll
        for (int o = 0; o < 2; ++o)
#pragma unroll
            for (int t = 0; t < kTD; ++t)
                if (d0 + t < D) out[((size_t)n * 2 + o) * vol + (size_t)(d0 + t) * plane + (size_t)h * W + w] = acc[o][t];
    }
}


// ---------------------------------------------------------------------------------------------------------------
// Round 4 form of the forward kernel.  The round-1 kernel above read one LDS dword per 4 FMAs (the LDS, not the vector
// pipe, set its time: 0.41 ms = 1.4 TB/s of input) and staged its halo with scalar loads in chunks of four channels.
//   thread = 4 (d) x 2 (w) output voxels x 2 outputs = 16 sums: per (channel, kh) it reads the 6 x 6 halo values it needs as
//            18 ds_read_b64 and feeds 3 kw x 3 kd x 8 voxels x 2 outputs = 144 FMAs (8 FMAs per LDS instruction);
//   block  = 4 x 12 x 40 output voxels (240 of 256 threads compute: 60 x 80 maps tile without padding), halo rows of 48 floats
//            = 12 aligned float4 (w0 - 4 .. w0 + 43), staged one channel at a time through registers (4 float4 per thread and
//            input, fetched while the previous channel is multiplied);
//   x2     = optional second input ADDED to x while staging: mvsnet.py:111-112 `x = conv0 + self.conv11(x); x = self.prob(x)` --
//            the transposed layer then leaves out the skip addition (its epilogue becomes a pure store stream instead of
//            load - wait - store round trips: csrc/costreg_bf16.hip) and the sum is formed here, in the same fp32 addition.
// W % 4 == 0 and 16-byte aligned tensors (whole float4 either inside or outside the volume); other shapes take the kernel above.
// ---------------------------------------------------------------------------------------------------------------
constexpr int kV2TD = 4, kV2TH = 12, kV2WP = 20, kV2TW = 2 * kV2WP;
constexpr int kV2HD = kV2TD + 2, kV2HH = kV2TH + 2, kV2Pitch = 48, kV2Q = kV2Pitch / 4;
constexpr int kV2Halo4 = kV2HD * kV2HH * kV2Q;                          // 1008 float4 per channel
constexpr int kV2Stage = (kV2Halo4 + kThreads - 1) / kThreads;          // 4 per thread

__global__ __launch_bounds__(kThreads, 4) void conv3d_k3_cout2_v2_kernel(const float* __restrict__ x, const float* __restrict__ x2,
                                                                          const float* __restrict__ wgt, const float* __restrict__ bias,
                                                                          float* __restrict__ out, int Cin, int D, int H, int W,
                                                                          int tiles_w, int tiles_h) {
    __shared__ float4 s_in[2][kV2Halo4];   // two channels in flight: one being multiplied, one being written
    const int tid = threadIdx.x;
    const int wp = tid % kV2WP, th = tid / kV2WP;            // th 0..12 (12: the 16 threads that only stage)
    const bool worker = th < kV2TH;
    const int bw = blockIdx.x % tiles_w, bh = blockIdx.x / tiles_w;
    const int w0 = bw * kV2TW, h0 = bh * kV2TH, d0 = blockIdx.y * kV2TD, n = blockIdx.z;
    const size_t plane = (size_t)H * W, vol = (size_t)D * plane;
    const float* xn = x + (size_t)n * Cin * vol;
    const float* x2n = x2 ? x2 + (size_t)n * Cin * vol : nullptr;

    // the two outputs of a voxel as one packed pair: v_pk_fma_f32 (value, value) * (w[0][..], w[1][..]) halves the multiply-adds
    // (one input 0.41 -> 0.36 ms; with two inputs the kernel is bound by their 1.18 GB and stays at 0.53)
    typedef float hf2 __attribute__((ext_vector_type(2)));
    hf2 acc[kV2TD][2];
#pragma unroll
    for (int t = 0; t < kV2TD; ++t) acc[t][0] = acc[t][1] = hf2{bias ? bias[0] : 0.0f, bias ? bias[1] : 0.0f};

    // staging plan, the same for every channel: float4 f = tid + 256*k of the halo tile -> offset inside a channel volume,
    // or -1 (outside the volume: zeros)
    int s_off[kV2Stage];
#pragma unroll
    for (int k = 0; k < kV2Stage; ++k) {
        const int f = tid + k * kThreads;
        const int dz = f / (kV2HH * kV2Q), r2 = f - dz * (kV2HH * kV2Q);
        const int hy = r2 / kV2Q, q = r2 - hy * kV2Q;
        const int d = d0 + dz - 1, h = h0 + hy - 1, w = w0 - 4 + 4 * q;
        const bool ok = f < kV2Halo4 && d >= 0 && d < D && h >= 0 && h < H && w >= 0 && w + 3 < W;
        s_off[k] = ok ? (int)((size_t)d * plane + (size_t)h * W + w) : -1;
    }
    float4 nxt[kV2Stage];
    auto fetch = [&](int c) {
        const float* xc = xn + (size_t)c * vol;
        const float* yc = x2n ? x2n + (size_t)c * vol : nullptr;
#pragma unroll
        for (int k = 0; k < kV2Stage; ++k) {
            float4 a = make_float4(0.f, 0.f, 0.f, 0.f);
            if (s_off[k] >= 0) {
                a = *reinterpret_cast<const float4*>(xc + s_off[k]);
                if (yc) {
                    const float4 b = *reinterpret_cast<const float4*>(yc + s_off[k]);
                    a.x = a.x + b.x; a.y = a.y + b.y; a.z = a.z + b.z; a.w = a.w + b.w;
                }
            }
            nxt[k] = a;
        }
    };
    auto stage = [&](int buf) {
#pragma unroll
        for (int k = 0; k < kV2Stage; ++k) {
            const int f = tid + k * kThreads;
            if (f < kV2Halo4) s_in[buf][f] = nxt[k];
        }
    };
    fetch(0);
    stage(0);
    if (Cin > 1) fetch(1);
    for (int c = 0; c < Cin; ++c) {
        const int buf = c & 1;
        __syncthreads();   // channel c is in s_in[buf]; everybody is done with s_in[buf ^ 1] (channel c - 1)
        if (c + 1 < Cin) stage(buf ^ 1);
        if (c + 2 < Cin) fetch(c + 2);
        if (worker) {
            const float* w0p = wgt + (size_t)c * 27;                   // weight[0][c][kd][kh][kw] (wave-uniform: scalar loads)
            const float* w1p = wgt + ((size_t)Cin + c) * 27;           // weight[1][c][...]
            const float* tile = reinterpret_cast<const float*>(s_in[buf]);
#pragma unroll 1
            for (int kh = 0; kh < 3; ++kh) {   // not unrolled: three rows' worth of halo values in flight spill
                // halo floats 2*wp + 3 .. 2*wp + 6 of the row (index 4 = w0): read as three aligned pairs from 2*wp + 2
                float v[kV2HD][4];
#pragma unroll
                for (int dz = 0; dz < kV2HD; ++dz) {
                    const float2* rowp = reinterpret_cast<const float2*>(tile + ((size_t)dz * kV2HH + th + kh) * kV2Pitch + 2 * wp + 2);
                    const float2 p0 = rowp[0], p1 = rowp[1], p2 = rowp[2];
                    v[dz][0] = p0.y; v[dz][1] = p1.x; v[dz][2] = p1.y; v[dz][3] = p2.x;
                }
#pragma unroll
                for (int kw = 0; kw < 3; ++kw)
#pragma unroll
                    for (int kd = 0; kd < 3; ++kd) {
                        const hf2 ab = {w0p[(kd * 3 + kh) * 3 + kw], w1p[(kd * 3 + kh) * 3 + kw]};
#pragma unroll
                        for (int t = 0; t < kV2TD; ++t)
#pragma unroll
                            for (int e = 0; e < 2; ++e) {
                                const float x = v[t + kd][e + kw];
                                acc[t][e] = __builtin_elementwise_fma(hf2{x, x}, ab, acc[t][e]);
                            }
                    }
            }
        }
    }
    const int h = h0 + th, w = w0 + 2 * wp;
    if (worker && h < H && w < W) {   // W % 4 == 0 and w even: both voxels of the pair are inside
#pragma unroll
        for (int o = 0; o < 2; ++o)
#pragma unroll
            for (int t = 0; t < kV2TD; ++t)
                if (d0 + t < D)
                    *reinterpret_cast<float2*>(out + ((size_t)n * 2 + o) * vol + (size_t)(d0 + t) * plane + (size_t)h * W + w) =
                        o ? make_float2(acc[t][0].y, acc[t][1].y) : make_float2(acc[t][0].x, acc[t][1].x);
    }
}
}  // namespace mvsdet

using namespace mvsdet;

// out (N,2,D,H,W) = conv3d(x [+ x2], weight (2,Cin,3,3,3), bias); x2 (same shape as x, or NULL) is added while staging.
extern "C" int mvsdet_conv3d_k3_cout2_sum_f32(const float* x, const float* x2, const float* weight, const float* bias, float* out,
                                              int N, int Cin, int D, int H, int W, mvsdet_stream_t stream) {
    MVS_REQUIRE(x && weight && out, "conv3d_k3_cout2: NULL pointer");
    MVS_REQUIRE(N > 0 && Cin > 0 && D > 0 && H > 0 && W > 0, "conv3d_k3_cout2: bad shape N=%d Cin=%d D=%d H=%d W=%d", N, Cin, D,
                H, W);
    MVS_REQUIRE((size_t)D * H * W < (size_t)INT32_MAX, "conv3d_k3_cout2: one channel volume exceeds 2^31 elements");
    const bool v2 = W % 4 == 0 && (((uintptr_t)x | (uintptr_t)x2) & 15u) == 0 && ((uintptr_t)out & 7u) == 0;
    MVS_REQUIRE(v2 || !x2, "conv3d_k3_cout2: a second input needs W %% 4 == 0 and 16-byte aligned tensors");
    if (v2) {
        const int tiles_w = (W + kV2TW - 1) / kV2TW, tiles_h = (H + kV2TH - 1) / kV2TH, tiles_d = (D + kV2TD - 1) / kV2TD;
        MVS_REQUIRE(N <= 65535 && tiles_d <= 65535, "conv3d_k3_cout2: N or D too large");
        dim3 grid((unsigned)(tiles_w * tiles_h), (unsigned)tiles_d, (unsigned)N);
        hipLaunchKernelGGL(conv3d_k3_cout2_v2_kernel, grid, dim3(kThreads), 0, (hipStream_t)stream, x, x2, weight, bias, out, Cin, D,
                           H, W, tiles_w, tiles_h);
    } else {
        const int tiles_w = (W + kTW - 1) / kTW, tiles_h = (H + kTH - 1) / kTH, tiles_d = (D + kTD - 1) / kTD;
        MVS_REQUIRE(N <= 65535 && tiles_d <= 65535, "conv3d_k3_cout2: N or D too large");
        dim3 grid((unsigned)(tiles_w * tiles_h), (unsigned)tiles_d, (unsigned)N);
        hipLaunchKernelGGL(conv3d_k3_cout2_kernel, grid, dim3(kThreads), 0, (hipStream_t)stream, x, weight, bias, out, Cin, D, H,
                           W, tiles_w, tiles_h);
    }
    MVS_LAUNCH_CHECK("conv3d_k3_cout2");
    return MVSDET_OK;
}

extern "C" int mvsdet_conv3d_k3_cout2_f32(const float* x, const float* weight, const float* bias, float* out, int N,
                                          int Cin, int D, int H, int W, mvsdet_stream_t stream) {
    return mvsdet_conv3d_k3_cout2_sum_f32(x, nullptr, weight, bias, out, N, Cin, D, H, W, stream);
}

// ---------------------------------------------------------------------------------------------------------------
// Backward of the head (training).  MIOpen needs 363 ms for forward + backward of this 64 -> 2 convolution at the
// reference-true shape -- two output channels make its GEMM formulations degenerate -- while both gradients are
// streaming problems of 8 GFMA each:
//   dX[n][c][v]       = sum over (o, tap) of gy[n][o][v - tap + 1] * W[o][c][tap]        (gy: 2 channels, staged once)
//   dW[o][c][tap]     = sum over (n, v)   of gy[n][o][v] * X[n][c][v + tap - 1]          (54 sums per channel)
// Same tile and thread layout as the forward kernel: 4 x 8 x 32 voxels per block, one (h, w) column of 4 voxels per
// thread, values along d read once from LDS and reused for the three kd.
// ---------------------------------------------------------------------------------------------------------------
namespace mvsdet {

__global__ __launch_bounds__(kThreads) void conv3d_k3_cout2_dx_kernel(const float* __restrict__ gy, const float* __restrict__ wgt,
                                                                       float* __restrict__ gx, int Cin, int D, int H, int W,
                                                                       int tiles_w, int tiles_h) {
    __shared__ float s_g[2][kHaloVox];
    const int tid = threadIdx.x;
    const int tw = tid % kTW, th = tid / kTW;
    const int bw = blockIdx.x % tiles_w, bh = blockIdx.x / tiles_w;
    const int w0 = bw * kTW, h0 = bh * kTH, d0 = blockIdx.y * kTD, n = blockIdx.z;
    const size_t plane = (size_t)H * W, vol = (size_t)D * plane;
    for (int e = tid; e < 2 * kHaloVox; e += kThreads) {
        const int o = e / kHaloVox, r = e - o * kHaloVox;
        const int dz = r / (kHH * kHW), r2 = r - dz * (kHH * kHW);
        const int hy = r2 / kHW, wx = r2 - hy * kHW;
        const int d = d0 + dz - 1, h = h0 + hy - 1, w = w0 + wx - 1;
        float v = 0.0f;
        if (d >= 0 && d < D && h >= 0 && h < H && w >= 0 && w < W) v = gy[((size_t)n * 2 + o) * vol + (size_t)d * plane + (size_t)h * W + w];
        s_g[o][r] = v;
    }
    __syncthreads();
    const int h = h0 + th, w = w0 + tw;
    const bool inside = h < H && w < W;
    for (int c = 0; c < Cin; ++c) {
        float acc[kTD] = {0.0f, 0.0f, 0.0f, 0.0f};
#pragma unroll
        for (int o = 0; o < 2; ++o) {
            const float* wp = wgt + ((size_t)o * Cin + c) * 27;   // wave-uniform
#pragma unroll
            for (int kh = 0; kh < 3; ++kh)
#pragma unroll
                for (int kw = 0; kw < 3; ++kw) {
                    // gy at (v - tap + 1): halo index of voxel t for tap k is t + 2 - k per dimension
                    const float* colp = &s_g[o][(th + 2 - kh) * kHW + (tw + 2 - kw)];
                    float v[kHD];
#pragma unroll
                    for (int dz = 0; dz < kHD; ++dz) v[dz] = colp[dz * (kHH * kHW)];
#pragma unroll
                    for (int kd = 0; kd < 3; ++kd) {
                        const float a = wp[(kd * 3 + kh) * 3 + kw];
#pragma unroll
                        for (int t = 0; t < kTD; ++t) acc[t] = fmaf(v[t + 2 - kd], a, acc[t]);
                    }
                }
        }
        if (inside) {
#pragma unroll
            for (int t = 0; t < kTD; ++t)
                if (d0 + t < D) gx[((size_t)n * Cin + c) * vol + (size_t)(d0 + t) * plane + (size_t)h * W + w] = acc[t];
        }
    }
}

// block = (input channel c, voxel split); 54 running sums per thread over all its tiles, one block reduction at the end
__global__ __launch_bounds__(kThreads) void conv3d_k3_cout2_dw_kernel(const float* __restrict__ x, const float* __restrict__ gy,
                                                                       float* __restrict__ partial, int N, int Cin, int D,
                                                                       int H, int W, int tiles_w, int tiles_h, int tiles_d,
                                                                       int nsplit) {
    __shared__ float s_x[kHaloVox];
    __shared__ float s_red[4][54];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int tw = tid % kTW, th = tid / kTW;
    const int c = blockIdx.x, split = blockIdx.y;
    const size_t plane = (size_t)H * W, vol = (size_t)D * plane;
    typedef float hf2 __attribute__((ext_vector_type(2)));   // the two output channels of a tap as one packed pair (v_pk_fma_f32)
    hf2 acc[27];
#pragma unroll
    for (int k = 0; k < 27; ++k) acc[k] = hf2{0.0f, 0.0f};
    const int tiles_per_view = tiles_d * tiles_h * tiles_w;
    const int ntiles = N * tiles_per_view;
    for (int tile = split; tile < ntiles; tile += nsplit) {
        const int n = tile / tiles_per_view, tv = tile - n * tiles_per_view;
        const int d0 = (tv / (tiles_h * tiles_w)) * kTD, t2 = tv % (tiles_h * tiles_w);
        const int h0 = (t2 / tiles_w) * kTH, w0 = (t2 % tiles_w) * kTW;
        __syncthreads();
        for (int r = tid; r < kHaloVox; r += kThreads) {
            const int dz = r / (kHH * kHW), r2 = r - dz * (kHH * kHW);
            const int hy = r2 / kHW, wx = r2 - hy * kHW;
            const int d = d0 + dz - 1, h = h0 + hy - 1, w = w0 + wx - 1;
            float v = 0.0f;
            if (d >= 0 && d < D && h >= 0 && h < H && w >= 0 && w < W) v = x[((size_t)n * Cin + c) * vol + (size_t)d * plane + (size_t)h * W + w];
            s_x[r] = v;
        }
        __syncthreads();
        const int h = h0 + th, w = w0 + tw;
        hf2 g[kTD];
#pragma unroll
        for (int t = 0; t < kTD; ++t) {
            const bool in = h < H && w < W && d0 + t < D;
            const size_t at = (size_t)(d0 + t) * plane + (size_t)h * W + w;
            g[t] = hf2{in ? gy[((size_t)n * 2 + 0) * vol + at] : 0.0f, in ? gy[((size_t)n * 2 + 1) * vol + at] : 0.0f};
        }
#pragma unroll
        for (int kh = 0; kh < 3; ++kh)
#pragma unroll
            for (int kw = 0; kw < 3; ++kw) {
                const float* colp = &s_x[(th + kh) * kHW + (tw + kw)];
                float v[kHD];
#pragma unroll
                for (int dz = 0; dz < kHD; ++dz) v[dz] = colp[dz * (kHH * kHW)];
#pragma unroll
                for (int kd = 0; kd < 3; ++kd)
#pragma unroll
                    for (int t = 0; t < kTD; ++t)
                        acc[(kd * 3 + kh) * 3 + kw] = __builtin_elementwise_fma(g[t], hf2{v[t + kd], v[t + kd]}, acc[(kd * 3 + kh) * 3 + kw]);
            }
    }
    // block reduction of the 54 sums: butterfly inside each wave, then the four waves through LDS
#pragma unroll
    for (int o = 0; o < 2; ++o)
#pragma unroll
        for (int k = 0; k < 27; ++k) {
            float v = o ? acc[k].y : acc[k].x;
#pragma unroll
            for (int off = 32; off > 0; off >>= 1) v += __shfl_xor(v, off, 64);
            if (lane == 0) s_red[wave][o * 27 + k] = v;
        }
    __syncthreads();
    if (tid < 54) {
        const float v = (s_red[0][tid] + s_red[1][tid]) + (s_red[2][tid] + s_red[3][tid]);
        const int o = tid / 27, k = tid % 27;
        partial[(((size_t)split * 2 + o) * Cin + c) * 27 + k] = v;
    }
}


// ---------------------------------------------------------------------------------------------------------------------------
// The same weight gradient on the bf16 matrix cores with three-term split operands (the arithmetic of costreg_dw_bf16.hip):
//
//     dW[o][c][kd,kh,kw] = sum over views and voxels u of  x[n,c,u] * gy[n,o,u - (kd-1,kh-1,kw-1)]      (gy = 0 outside the volume)
//
// one GEMM with the voxels as the reduction: M = the input channels (MT tiles of 16), N = the 54 (o, tap) pairs padded to 64, on
// v_mfma_f32_16x16x32_bf16.  The SMALL tensor is the shifted one: x (N,Cin,D,H,W -- 590 MB at the reference-true shape, the
// only stream that matters) goes from global memory straight into the A fragments, 16 bytes per lane, never through the LDS;
// gy (2 channels) is staged per x row as its 3 x 3 neighbouring rows of both channels with zero halos, and a lane reads its
// (o, tap) column's 8 voxels from there at the tap's shift.
//   wave  : walks x rows (n, d, h), all input channels of a row, 32 voxels per k-step.  The 32 k of an instruction are the voxels
//           w0 + 4g + {0..3} and w0 + 16 + 4g + {0..3} of k-group g = lane >> 4: the four groups of a channel read 64
//           contiguous bytes per load instruction
//   block : 4 independent waves (their own gy rows in LDS, no barrier in the loop); the four accumulator sets are added
//           through the LDS at the end: partial[block][o][c][27], the caller adds the blocks up
// The fp32 kernel above: 1.0 ms at (40,64,12,60,80); this one is bound by the x stream.
typedef short hd_bf16x8 __attribute__((ext_vector_type(8)));
typedef float hd_f32x4 __attribute__((ext_vector_type(4)));
typedef unsigned hd_u32x4 __attribute__((ext_vector_type(4)));

constexpr int kHdwRows = 19;   // 2 output channels x 3 x 3 neighbouring rows of grad_out, and a row that stays zero
constexpr int kHdwLead = 4;    // floats in front of a row's voxel 0 (the last of them is the left halo): the data is 16-byte aligned

__host__ __device__ constexpr int hdw_pitch(int W) {
    // voxels up to the k-step's end + the right halo, a multiple of 4 floats that is no multiple of 32 (rows on different banks)
    const int p = kHdwLead + (W + 31) / 32 * 32 + 4;
    return p % 32 == 0 ? p + 4 : p;
}
__host__ __device__ constexpr size_t hdw_lds_bytes(int Cin, int W) { return ((size_t)4 * kHdwRows * hdw_pitch(W) + (size_t)Cin * 64) * sizeof(float); }

__device__ __forceinline__ unsigned hd_pack(__bf16 a, __bf16 b) {
    return (unsigned)__builtin_bit_cast(unsigned short, a) | ((unsigned)__builtin_bit_cast(unsigned short, b) << 16);
}
// 8 floats -> their bf16 roundings (hi) and the roundings of the exact remainders (mid)
__device__ __forceinline__ void hd_cut8(const float (&f)[8], hd_bf16x8& hi, hd_bf16x8& mid) {
    hd_u32x4 h, m;
#pragma unroll
    for (int p = 0; p < 4; ++p) {
        const __bf16 a0 = (__bf16)f[2 * p], a1 = (__bf16)f[2 * p + 1];
        h[p] = hd_pack(a0, a1);
        m[p] = hd_pack((__bf16)(f[2 * p] - (float)a0), (__bf16)(f[2 * p + 1] - (float)a1));
    }
    hi = __builtin_bit_cast(hd_bf16x8, h);
    mid = __builtin_bit_cast(hd_bf16x8, m);
}

template <int MT>
__global__ __launch_bounds__(kThreads) void conv3d_k3_cout2_dw_bf16x3_kernel(const float* __restrict__ x, const float* __restrict__ gy,
                                                                             float* __restrict__ partial, int N, int D, int H, int W) {
    constexpr int Cin = 16 * MT;
    extern __shared__ float s_hd[];
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int g = lane >> 4, r16 = lane & 15;
    const int pitch = hdw_pitch(W);
    float* const rows = s_hd + wave * kHdwRows * pitch;
    float* const red = s_hd + 4 * kHdwRows * pitch;
    for (int i = lane; i < kHdwRows * pitch; i += 64) rows[i] = 0.0f;   // halos, the columns behind W and the zero row stay so
    // where the lane's (o, tap) column of each N tile starts reading: row (o, 1 - (kd-1), 1 - (kh-1)), column shifted by 1 - kw
    int boff[4];
#pragma unroll
    for (int nt = 0; nt < 4; ++nt) {
        const int j = 16 * nt + r16;
        const int o = j / 27, tap = j - 27 * o, kd = tap / 9, kh = (tap / 3) % 3, kw = tap % 3;
        boff[nt] = (j < 54 ? (o * 9 + (2 - kd) * 3 + (2 - kh)) * pitch + 1 - kw : 18 * pitch) + kHdwLead + 4 * g;
    }
    const int HW = H * W, W4 = W >> 2;
    const size_t vol = (size_t)D * HW;
    const int nrows = N * D * H, ksteps = (W + 31) >> 5;
    hd_f32x4 acc[MT][4];
#pragma unroll
    for (int mt = 0; mt < MT; ++mt)
#pragma unroll
        for (int nt = 0; nt < 4; ++nt) acc[mt][nt] = hd_f32x4{0.f, 0.f, 0.f, 0.f};

    for (int row = blockIdx.x * 4 + wave; row < nrows; row += gridDim.x * 4) {
        const int n = row / (D * H), dh = row - n * (D * H), d = dh / H, h = dh - d * H;
        // the 18 rows of grad_out around (d, h): zero where the volume ends
        for (int q = lane; q < 18 * W4; q += 64) {
            const int rr = q / W4, c4 = q - rr * W4;
            const int o = rr / 9, zd = d + (rr % 9) / 3 - 1, zh = h + rr % 3 - 1;
            float4 v = make_float4(0.f, 0.f, 0.f, 0.f);
            if (zd >= 0 && zd < D && zh >= 0 && zh < H)
                v = *reinterpret_cast<const float4*>(gy + ((size_t)n * 2 + o) * vol + (size_t)zd * HW + (size_t)zh * W + 4 * c4);
            *reinterpret_cast<float4*>(rows + rr * pitch + kHdwLead + 4 * c4) = v;
        }
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
        __builtin_amdgcn_wave_barrier();
        const float* xr = x + ((size_t)n * Cin + r16) * vol + (size_t)d * HW + (size_t)h * W + 4 * g;
        for (int ks = 0; ks < ksteps; ++ks) {
            const int w0 = 32 * ks + 4 * g;
            hd_bf16x8 ahi[MT], amid[MT];
#pragma unroll
            for (int mt = 0; mt < MT; ++mt) {
                const float* p = xr + (size_t)(16 * mt) * vol + 32 * ks;
                float4 v0 = make_float4(0.f, 0.f, 0.f, 0.f), v1 = v0;
                if (w0 < W) v0 = *reinterpret_cast<const float4*>(p);
                if (w0 + 16 < W) v1 = *reinterpret_cast<const float4*>(p + 16);
                const float f[8] = {v0.x, v0.y, v0.z, v0.w, v1.x, v1.y, v1.z, v1.w};
                hd_cut8(f, ahi[mt], amid[mt]);
            }
#pragma unroll
            for (int nt = 0; nt < 4; ++nt) {
                const float* b = rows + boff[nt] + 32 * ks;
                const float f[8] = {b[0], b[1], b[2], b[3], b[16], b[17], b[18], b[19]};
                hd_bf16x8 bhi, bmid;
                hd_cut8(f, bhi, bmid);
#pragma unroll
                for (int mt = 0; mt < MT; ++mt) {
                    acc[mt][nt] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(amid[mt], bhi, acc[mt][nt], 0, 0, 0);
                    acc[mt][nt] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(ahi[mt], bmid, acc[mt][nt], 0, 0, 0);
                    acc[mt][nt] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(ahi[mt], bhi, acc[mt][nt], 0, 0, 0);
                }
            }
        }
        __builtin_amdgcn_wave_barrier();   // the row's reads are issued before the next row's stores (one wave: LDS in order)
    }
    // the four waves' sums, one after the other into the reduction area; row = channel 16 mt + 4 g + i, column = 16 nt + r16
#pragma unroll 1
    for (int wv = 0; wv < 4; ++wv) {
        if (wave == wv) {
#pragma unroll
            for (int mt = 0; mt < MT; ++mt)
#pragma unroll
                for (int nt = 0; nt < 4; ++nt)
#pragma unroll
                    for (int i = 0; i < 4; ++i) {
                        float* cell = red + (16 * mt + 4 * g + i) * 64 + 16 * nt + r16;
                        *cell = (wv ? *cell : 0.0f) + acc[mt][nt][i];
                    }
        }
        __syncthreads();
    }
    for (int e = tid; e < Cin * 54; e += kThreads) {
        const int c = e / 54, j = e - 54 * c, o = j / 27, tap = j - 27 * o;
        partial[(((size_t)blockIdx.x * 2 + o) * Cin + c) * 27 + tap] = red[c * 64 + j];
    }
}

}  // namespace mvsdet

extern "C" int mvsdet_conv3d_k3_cout2_dx_f32(const float* grad_out, const float* weight, float* grad_x, int N, int Cin, int D,
                                             int H, int W, mvsdet_stream_t stream) {
    MVS_REQUIRE(grad_out && weight && grad_x, "conv3d_k3_cout2_dx: NULL pointer");
    MVS_REQUIRE(N > 0 && Cin > 0 && D > 0 && H > 0 && W > 0, "conv3d_k3_cout2_dx: bad shape");
    const int tiles_w = (W + kTW - 1) / kTW, tiles_h = (H + kTH - 1) / kTH, tiles_d = (D + kTD - 1) / kTD;
    MVS_REQUIRE(N <= 65535 && tiles_d <= 65535, "conv3d_k3_cout2_dx: N or D too large");
    dim3 grid((unsigned)(tiles_w * tiles_h), (unsigned)tiles_d, (unsigned)N);
    hipLaunchKernelGGL(conv3d_k3_cout2_dx_kernel, grid, dim3(kThreads), 0, (hipStream_t)stream, grad_out, weight, grad_x, Cin,
                       D, H, W, tiles_w, tiles_h);
    MVS_LAUNCH_CHECK("conv3d_k3_cout2_dx");
    return MVSDET_OK;
}

extern "C" int mvsdet_conv3d_k3_cout2_dw_f32(const float* x, const float* grad_out, float* partial, size_t partial_bytes,
                                             int nsplit, int N, int Cin, int D, int H, int W, mvsdet_stream_t stream) {
    MVS_REQUIRE(x && grad_out && partial, "conv3d_k3_cout2_dw: NULL pointer");
    MVS_REQUIRE(N > 0 && Cin > 0 && D > 0 && H > 0 && W > 0, "conv3d_k3_cout2_dw: bad shape");
    MVS_REQUIRE(nsplit > 0 && nsplit <= 65535 && Cin <= 65535, "conv3d_k3_cout2_dw: nsplit or Cin too large");
    if (partial_bytes < (size_t)nsplit * 2 * Cin * 27 * sizeof(float)) {
        set_error("conv3d_k3_cout2_dw: partial buffer %zu B too small", partial_bytes);
        return MVSDET_ERR_WORKSPACE;
    }
    const int tiles_w = (W + kTW - 1) / kTW, tiles_h = (H + kTH - 1) / kTH, tiles_d = (D + kTD - 1) / kTD;
    MVS_REQUIRE((long long)N * tiles_d * tiles_h * tiles_w < INT32_MAX, "conv3d_k3_cout2_dw: too many tiles");
    dim3 grid((unsigned)Cin, (unsigned)nsplit);
    hipLaunchKernelGGL(conv3d_k3_cout2_dw_kernel, grid, dim3(kThreads), 0, (hipStream_t)stream, x, grad_out, partial, N, Cin, D,
                       H, W, tiles_w, tiles_h, tiles_d, nsplit);
    MVS_LAUNCH_CHECK("conv3d_k3_cout2_dw");
    return MVSDET_OK;
}

// Whether mvsdet_conv3d_k3_cout2_dw_bf16x3 takes this (Cin, W): its shape conditions, the LDS of a row stage included -- what a
// caller asks before it chooses between that entry point and the fp32 one (the pointer alignment is the caller's to check).
extern "C" int mvsdet_conv3d_k3_cout2_dw_bf16x3_ok(int Cin, int W) {
    return (Cin == 16 || Cin == 32 || Cin == 64) && W > 0 && W % 4 == 0 && W <= 4096 && hdw_lds_bytes(Cin, W) <= 160 * 1024;
}

// bf16x3 form of the weight gradient (see conv3d_k3_cout2_dw_bf16x3_kernel): Cin in {16, 32, 64}, W % 4 == 0, 16-byte aligned
// tensors; nsplit = blocks = rows of `partial`.  Within ~1e-5 of the fp32 kernel's sums (three-term split operands).
extern "C" int mvsdet_conv3d_k3_cout2_dw_bf16x3(const float* x, const float* grad_out, float* partial, size_t partial_bytes,
                                                int nsplit, int N, int Cin, int D, int H, int W, mvsdet_stream_t stream) {
    MVS_REQUIRE(x && grad_out && partial, "conv3d_k3_cout2_dw_bf16x3: NULL pointer");
    MVS_REQUIRE(N > 0 && D > 0 && H > 0 && W > 0, "conv3d_k3_cout2_dw_bf16x3: bad shape");
    MVS_REQUIRE(Cin == 16 || Cin == 32 || Cin == 64, "conv3d_k3_cout2_dw_bf16x3: Cin=%d is not 16, 32 or 64", Cin);
    MVS_REQUIRE(W % 4 == 0 && W <= 4096, "conv3d_k3_cout2_dw_bf16x3: W=%d is no multiple of 4 (or > 4096)", W);
    MVS_REQUIRE(((uintptr_t)x | (uintptr_t)grad_out) % 16 == 0, "conv3d_k3_cout2_dw_bf16x3: tensors must be 16-byte aligned");
    MVS_REQUIRE(nsplit > 0 && nsplit <= 65535, "conv3d_k3_cout2_dw_bf16x3: nsplit=%d outside [1,65535]", nsplit);
    MVS_REQUIRE((long long)N * D * H < INT32_MAX && (long long)D * H * W < INT32_MAX, "conv3d_k3_cout2_dw_bf16x3: volume too large");
    if (partial_bytes < (size_t)nsplit * 2 * Cin * 27 * sizeof(float)) {
        set_error("conv3d_k3_cout2_dw_bf16x3: partial buffer %zu B too small", partial_bytes);
        return MVSDET_ERR_WORKSPACE;
    }
    const size_t lds = hdw_lds_bytes(Cin, W);
    MVS_REQUIRE(lds <= 160 * 1024, "conv3d_k3_cout2_dw_bf16x3: W=%d needs %zu B of LDS", W, lds);
#define MVS_HDW_LAUNCH(MTV)                                                                                              \
    {                                                                                                                    \
        auto* k = conv3d_k3_cout2_dw_bf16x3_kernel<MTV>;                                                                 \
        if (lds > 48 * 1024 && hipFuncSetAttribute(reinterpret_cast<const void*>(k), hipFuncAttributeMaxDynamicSharedMemorySize, \
                                                   160 * 1024) != hipSuccess) {                                          \
            set_error("conv3d_k3_cout2_dw_bf16x3: hipFuncSetAttribute(MaxDynamicSharedMemorySize) failed");              \
            return MVSDET_ERR_HIP;                                                                                       \
        }                                                                                                                \
        hipLaunchKernelGGL(k, dim3((unsigned)nsplit), dim3(kThreads), lds, (hipStream_t)stream, x, grad_out, partial, N, D, H, W); \
    }
    if (Cin == 64) MVS_HDW_LAUNCH(4)
    else if (Cin == 32) MVS_HDW_LAUNCH(2)
    else MVS_HDW_LAUNCH(1)
#undef MVS_HDW_LAUNCH
    MVS_LAUNCH_CHECK("conv3d_k3_cout2_dw_bf16x3");
    return MVSDET_OK;
}
