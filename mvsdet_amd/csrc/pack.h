// Channel-last "packed" feature layout (include/mvsdet_hip.h) <-> (N,C,H,W).
// Kernels are static: each translation unit that launches them carries its own copy (no -fgpu-rdc).
#pragma once
#include "common.h"

namespace mvsdet {

// ---------------------------------------------------------------------------------------------
// pack: (N,C,H,W) strided -> packed[n][pix][4*g+i] = feat[n][i*G+g][pix]
// One block = 64 pixels x 16 channel groups (64 channels).  Reads are coalesced along W, writes are
// 256-byte runs along the packed channel axis.
// ---------------------------------------------------------------------------------------------
static __global__ __launch_bounds__(kThreads) void pack_features_kernel(const float* __restrict__ feat, int64_t s0, int64_t s1,
                                                                  int64_t s2, int64_t s3, float* __restrict__ packed,
                                                                  int C, int G, int H, int W) {
    __shared__ float tile[64][65];
    const int HW = H * W;
    const int pix0 = blockIdx.x * 64, g0 = blockIdx.y * 16, n = blockIdx.z;
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    {
        const int pix = pix0 + lane;
        const int y = pix / W, x = pix - y * W;
        const float* src = feat + (int64_t)n * s0 + (int64_t)y * s2 + (int64_t)x * s3;
        for (int r = wave; r < 64; r += 4) {
            const int g = g0 + (r & 15), c = (r >> 4) * G + g;
            float v = 0.0f;
            if (pix < HW && g < G && c < C) v = src[(int64_t)c * s1];
            tile[r][lane] = v;
        }
    }
    __syncthreads();
    const int gg = threadIdx.x & 15;
    for (int p = threadIdx.x >> 4; p < 64; p += 16) {
        const int pix = pix0 + p;
        if (pix < HW && g0 + gg < G) {
            float4 v = make_float4(tile[gg][p], tile[16 + gg][p], tile[32 + gg][p], tile[48 + gg][p]);
            *reinterpret_cast<float4*>(packed + ((size_t)n * HW + pix) * (size_t)(4 * G) + 4 * (g0 + gg)) = v;
        }
    }
}

// unpack-add: gfeat[n][c][pix] = gpacked[n][pix][4g+i]  (used by the backward pass)
static __global__ __launch_bounds__(kThreads) void unpack_features_kernel(const float* __restrict__ packed,
                                                                    float* __restrict__ feat, int C, int G, int H,
                                                                    int W) {
    __shared__ float tile[64][65];
    const int HW = H * W;
    const int pix0 = blockIdx.x * 64, g0 = blockIdx.y * 16, n = blockIdx.z;
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int gg = threadIdx.x & 15;
    for (int p = threadIdx.x >> 4; p < 64; p += 16) {
        const int pix = pix0 + p;
        float4 v = make_float4(0.f, 0.f, 0.f, 0.f);
        if (pix < HW && g0 + gg < G)
            v = *reinterpret_cast<const float4*>(packed + ((size_t)n * HW + pix) * (size_t)(4 * G) + 4 * (g0 + gg));
        tile[gg][p] = v.x;
        tile[16 + gg][p] = v.y;
        tile[32 + gg][p] = v.z;
        tile[48 + gg][p] = v.w;
    }
    __syncthreads();
    const int pix = pix0 + lane;
    for (int r = wave; r < 64; r += 4) {
        const int g = g0 + (r & 15), c = (r >> 4) * G + g;
        if (pix < HW && g < G && c < C) feat[((size_t)n * C + c) * HW + pix] = tile[r][lane];
    }
}

}  // namespace mvsdet
