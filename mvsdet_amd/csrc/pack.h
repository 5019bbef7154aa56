// Channel-last "packed" feature layout (include/mvsdet_hip.h) <-> (N,C,H,W).
//
//   packed[n][s][pix][q]   s = channel slab (32 channels), pix = y*W + x, q in [0,32)
//   q = 4*g + i  <->  channel c = 32*s + 8*i + g        (g in [0,8): lane of a texel, i in [0,4): float4 slot)
//
// One texel of one slab is 128 B = one cache line = 8 lanes x float4.  Slab-major order gives every slab
// of a view a contiguous H*W*128-byte image: a slab's share of the source maps of one reference view
// (K * H*W*128 B, 4.9 MB at 120x160, K=2) is what has to stay resident in ONE XCD's 4 MiB L2 while that
// XCD sweeps its slab (planesweep.hip).  The i-major channel permutation makes the LDS transposes of the
// sweep / lifting kernels bank-conflict free (lane g writes rows 8*i+g).
//
// Kernels are static: each translation unit that launches them carries its own copy (no -fgpu-rdc).
#pragma once
#include <hip/hip_fp16.h>
#include "common.h"

namespace mvsdet {

constexpr int kSlab = 32;  // channels per slab

__host__ __device__ __forceinline__ int num_slabs(int C) { return (C + kSlab - 1) / kSlab; }

__device__ __forceinline__ float load_as_float(const float* p) { return *p; }
__device__ __forceinline__ float load_as_float(const __half* p) { return __half2float(*p); }  // exact

// One block = 64 pixels x one slab.  Reads are coalesced along W (256 B per channel row), writes are
// 128-byte texels, 8 KiB contiguous per block.  InT = float, or __half for fp16 feature maps (the packed
// maps stay fp32: they are 4 % of the sweep's traffic and the arithmetic is fp32 either way).
template <typename InT>
static __global__ __launch_bounds__(kThreads) void pack_features_kernel(const InT* __restrict__ feat, int64_t s0,
                                                                         int64_t s1, int64_t s2, int64_t s3,
                                                                         float* __restrict__ packed, int C, int S,
                                                                         int H, int W) {
    __shared__ float tile[kSlab][65];
    const int HW = H * W;
    const int pix0 = blockIdx.x * 64, s = blockIdx.y, n = blockIdx.z;
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    {
        const int pix = pix0 + lane;
        const int y = pix / W, x = pix - y * W;
        const InT* src = feat + (int64_t)n * s0 + (int64_t)y * s2 + (int64_t)x * s3;
        for (int r = wave; r < kSlab; r += 4) {
            const int c = s * kSlab + r;
            float v = 0.0f;
            if (pix < HW && c < C) v = load_as_float(src + (int64_t)c * s1);
            tile[r][lane] = v;
        }
    }
    __syncthreads();
    const int g = threadIdx.x & 7;
    for (int p = threadIdx.x >> 3; p < 64; p += 32) {
        const int pix = pix0 + p;
        if (pix < HW) {
            const float4 v = make_float4(tile[g][p], tile[8 + g][p], tile[16 + g][p], tile[24 + g][p]);
            *reinterpret_cast<float4*>(packed + (((size_t)n * S + s) * HW + pix) * kSlab + 4 * g) = v;
        }
    }
}

// The same for dense fp32 maps (pixel stride 1, row stride W, H*W and the view / channel strides multiples of 4, 16-byte aligned):
// one block = 128 pixels x one slab, a thread loads FOUR float4 (4 pixels of 4 channel rows: 512 B per channel row and
// wave-instruction instead of 256), LDS rows 136 floats apart: the float4 writes are aligned and the transposing reads
// (lane = (pixel, g): rows g, 8+g, 16+g, 24+g of one pixel) fall on bank 8*g + pixel -- all 64 banks, no conflict.
static __global__ __launch_bounds__(kThreads) void pack_features_dense_kernel(const float* __restrict__ feat, int64_t s0, int64_t s1,
                                                                               float* __restrict__ packed, int C, int S, int HW) {
    constexpr int kPix = 128, kPitch = 136;
    __shared__ float tile[kSlab * kPitch];
    const int pix0 = blockIdx.x * kPix, s = blockIdx.y, n = blockIdx.z;
    const int q = threadIdx.x & 31, r0 = threadIdx.x >> 5;   // pixel quad, first of the thread's four channel rows
    {
        const int pix = pix0 + 4 * q;
        const float* src = feat + (int64_t)n * s0 + pix;
        float4 v[4];
#pragma unroll
        for (int k = 0; k < 4; ++k) {
            const int c = s * kSlab + r0 + 8 * k;
            v[k] = (pix < HW && c < C) ? *reinterpret_cast<const float4*>(src + (int64_t)c * s1) : make_float4(0.f, 0.f, 0.f, 0.f);
        }
#pragma unroll
        for (int k = 0; k < 4; ++k) *reinterpret_cast<float4*>(&tile[(r0 + 8 * k) * kPitch + 4 * q]) = v[k];
    }
    __syncthreads();
    const int g = threadIdx.x & 7;
#pragma unroll
    for (int p = threadIdx.x >> 3; p < kPix; p += 32) {
        const int pix = pix0 + p;
        if (pix < HW) {
            const float4 v = make_float4(tile[g * kPitch + p], tile[(8 + g) * kPitch + p], tile[(16 + g) * kPitch + p],
                                         tile[(24 + g) * kPitch + p]);
            *reinterpret_cast<float4*>(packed + (((size_t)n * S + s) * HW + pix) * kSlab + 4 * g) = v;
        }
    }
}

// unpack: feat[n][c][pix] = packed[n][s][pix][4g+i]  (dense NCHW output; used by the backward pass)
static __global__ __launch_bounds__(kThreads) void unpack_features_kernel(const float* __restrict__ packed,
                                                                           float* __restrict__ feat, int C, int S, int H,
                                                                           int W) {
    __shared__ float tile[kSlab][65];
    const int HW = H * W;
    const int pix0 = blockIdx.x * 64, s = blockIdx.y, n = blockIdx.z;
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int g = threadIdx.x & 7;
    for (int p = threadIdx.x >> 3; p < 64; p += 32) {
        const int pix = pix0 + p;
        float4 v = make_float4(0.f, 0.f, 0.f, 0.f);
        if (pix < HW) v = *reinterpret_cast<const float4*>(packed + (((size_t)n * S + s) * HW + pix) * kSlab + 4 * g);
        tile[g][p] = v.x;
        tile[8 + g][p] = v.y;
        tile[16 + g][p] = v.z;
        tile[24 + g][p] = v.w;
    }
    __syncthreads();
    const int pix = pix0 + lane;
    for (int r = wave; r < kSlab; r += 4) {
        const int c = s * kSlab + r;
        if (pix < HW && c < C) feat[((size_t)n * C + c) * HW + pix] = tile[r][lane];
    }
}

// The same for H*W a multiple of 4 and a 16-byte aligned output: 128 pixels per block, float4 stores (the mirror image of
// pack_features_dense_kernel: LDS rows 136 floats apart, lane (pixel, g) writes rows g, 8+g, 16+g, 24+g on bank 8*g + pixel).
static __global__ __launch_bounds__(kThreads) void unpack_features_dense_kernel(const float* __restrict__ packed, float* __restrict__ feat,
                                                                                 int C, int S, int HW) {
    constexpr int kPix = 128, kPitch = 136;
    __shared__ float tile[kSlab * kPitch];
    const int pix0 = blockIdx.x * kPix, s = blockIdx.y, n = blockIdx.z;
    const int g = threadIdx.x & 7;
#pragma unroll
    for (int p = threadIdx.x >> 3; p < kPix; p += 32) {
        const int pix = pix0 + p;
        float4 v = make_float4(0.f, 0.f, 0.f, 0.f);
        if (pix < HW) v = *reinterpret_cast<const float4*>(packed + (((size_t)n * S + s) * HW + pix) * kSlab + 4 * g);
        tile[g * kPitch + p] = v.x;
        tile[(8 + g) * kPitch + p] = v.y;
        tile[(16 + g) * kPitch + p] = v.z;
        tile[(24 + g) * kPitch + p] = v.w;
    }
    __syncthreads();
    const int q = threadIdx.x & 31, r0 = threadIdx.x >> 5;
    const int pix = pix0 + 4 * q;
#pragma unroll
    for (int k = 0; k < 4; ++k) {
        const int c = s * kSlab + r0 + 8 * k;
        if (pix < HW && c < C)
            *reinterpret_cast<float4*>(feat + ((size_t)n * C + c) * HW + pix) = *reinterpret_cast<const float4*>(&tile[(r0 + 8 * k) * kPitch + 4 * q]);
    }
}

}  // namespace mvsdet
