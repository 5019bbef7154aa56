// Backward of the fused plane-sweep variance (a3+a4) with respect to the 2-D features -- the only input
// with a gradient: the sampling grid is built under torch.no_grad() (mvs_models/module.py:115).
//
//   var = Q/(K+1) - (S/(K+1))^2,  S = f + sum_j w_j,  Q = f^2 + sum_j w_j^2
//   dvar/dv = 2 v r - 2 S r^2   (r = 1/(K+1)) for each contributing value v in {f, w_1..w_K}
//   w_j = sum_t weight_t * tap_t  ->  dL/dtap_t += weight_t * dL/dw_j   (bilinear scatter)
//
// Same decomposition as the forward kernel (block = view x pixel tile, packed channel-last maps); the
// incoming gradient tile is staged through LDS so it is read as full rows of the (N,C,D,H,W) tensor, the
// warped values are recomputed, and gradients are accumulated with fp32 atomics into a zero-initialised
// packed gradient map that is unpacked to (N,C,H,W) afterwards.  Bound: the chip-wide float-atomic rate
// (MI355X_MICROARCH "Global float atomics"), 4*K atomic dwords per output element.
#include "common.h"
#include "pack.h"

namespace mvsdet {

template <int K, int TP>
__global__ __launch_bounds__(kThreads) void plane_sweep_variance_bwd_kernel(
    const float* __restrict__ packed, const int64_t* __restrict__ nbr, const float* __restrict__ proj,
    const float* __restrict__ depth, const float* __restrict__ gvar, float* __restrict__ gpacked, int N, int C, int G,
    int D, int H, int W, int tiles, int lp_log2) {
    constexpr int KK = K > 0 ? K : 1;
    constexpr int PW = TP / 4;
    __shared__ float s_tile[256 * (TP + 1)];
    __shared__ int4 s_off[KK][TP];
    __shared__ float4 s_w[KK][TP];

    const int HW = H * W;
    const int L = xcd_contiguous_id(blockIdx.x, gridDim.x);
    const int n = L / tiles, tile = L - n * tiles;
    const int pix0 = tile * TP;
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int LP = 1 << lp_log2, PPI = 64 >> lp_log2;
    const int gl = lane & (LP - 1), ps = lane >> lp_log2;
    // packed layout (pack.h): [view][slab][pixel][32]; channel group gg = 8*slab + g, G = 8*S groups
    const size_t slab_stride = (size_t)HW * kSlab;
    const size_t view_stride = slab_stride * (size_t)(G / 8);
    const float* ref_base = packed + (size_t)n * view_stride;
    size_t nb_view[KK];
#pragma unroll
    for (int j = 0; j < K; ++j) {
        int64_t v = nbr[(size_t)n * K + j];
        v = v < 0 ? 0 : (v >= N ? N - 1 : v);
        nb_view[j] = (size_t)v * view_stride;
    }
    const float r = 1.0f / (float)(K + 1);
    const float two_r = 2.0f * r, two_r2 = 2.0f * r * r;
    const int chunks = (G + 63) / 64;

    for (int ci = 0; ci < chunks; ++ci) {
        const int rg = min(64, G - ci * 64);
        const bool gvalid = gl < rg;
        const int gq_ = ci * 64 + (gvalid ? gl : 0);
        const size_t goff = (size_t)(gq_ >> 3) * slab_stride + 4 * (gq_ & 7);  // slab image + lane slot
        for (int d = 0; d < D; ++d) {
            // tap table + gradient tile [channel row][pixel]
            if (K > 0) {
                const float dval = depth[(size_t)n * D + d];
                for (int idx = threadIdx.x; idx < K * TP; idx += kThreads) {
                    const int j = idx / TP, p = idx - j * TP;
                    const int pix = pix0 + p;
                    int4 o = make_int4(0, 0, 0, 0);
                    float4 w = make_float4(0.f, 0.f, 0.f, 0.f);
                    if (pix < HW) {
                        const int y = pix / W, x = pix - y * W;
                        compute_taps(proj + ((size_t)n * K + j) * 16, (float)x, (float)y, dval, H, W, kSlab, o, w);
                    }
                    s_off[j][p] = o;
                    s_w[j][p] = w;
                }
            }
            {
                constexpr int RPI = 64 / TP;
                const int pp = lane % TP, rsub = lane / TP;
                const int rows = 4 * rg;
                const bool pvalid = pix0 + pp < HW;
                for (int rr = wave * RPI + rsub; rr < rows; rr += 4 * RPI) {
                    const int i = rr / rg, gg = rr - i * rg;
                    const int gq = ci * 64 + gg;
                    const int c = (gq >> 3) * kSlab + 8 * i + (gq & 7);
                    float v = 0.0f;
                    if (c < C && pvalid) v = gvar[(((size_t)n * C + c) * D + d) * HW + pix0 + pp];
                    s_tile[rr * (TP + 1) + pp] = v;
                }
            }
            __syncthreads();
            for (int s = 0; s < PW / PPI; ++s) {
                const int p = wave * PW + s * PPI + ps;
                const bool live = gvalid && (pix0 + p < HW);
                const int pix = min(pix0 + p, HW - 1);
                const float4 f = *reinterpret_cast<const float4*>(ref_base + (size_t)pix * kSlab + goff);
                const float* t = s_tile + gl * (TP + 1) + p;
                float go[4] = {0.f, 0.f, 0.f, 0.f};
                if (gvalid) {
                    go[0] = t[0];
                    go[1] = t[rg * (TP + 1)];
                    go[2] = t[2 * rg * (TP + 1)];
                    go[3] = t[3 * rg * (TP + 1)];
                }
                float fv[4] = {f.x, f.y, f.z, f.w};
                float S[4] = {f.x, f.y, f.z, f.w};
                float wv[KK][4];
#pragma unroll
                for (int j = 0; j < K; ++j) {
                    const int4 o = s_off[j][p];
                    const float4 w = s_w[j][p];
                    const float* b = packed + nb_view[j] + goff;
                    const float4 t0 = *reinterpret_cast<const float4*>(b + o.x);
                    const float4 t1 = *reinterpret_cast<const float4*>(b + o.y);
                    const float4 t2 = *reinterpret_cast<const float4*>(b + o.z);
                    const float4 t3 = *reinterpret_cast<const float4*>(b + o.w);
                    const float a0[4] = {t0.x, t0.y, t0.z, t0.w}, a1[4] = {t1.x, t1.y, t1.z, t1.w};
                    const float a2[4] = {t2.x, t2.y, t2.z, t2.w}, a3[4] = {t3.x, t3.y, t3.z, t3.w};
#pragma unroll
                    for (int i = 0; i < 4; ++i) {
                        float v = a0[i] * w.x;
                        v = fmaf(a1[i], w.y, v);
                        v = fmaf(a2[i], w.z, v);
                        v = fmaf(a3[i], w.w, v);
                        wv[j][i] = v;
                        S[i] += v;
                    }
                }
                if (live) {
                    float* gref = gpacked + (size_t)n * view_stride + (size_t)pix * kSlab + goff;
#pragma unroll
                    for (int i = 0; i < 4; ++i) atomicAdd(gref + i, go[i] * (two_r * fv[i] - two_r2 * S[i]));
#pragma unroll
                    for (int j = 0; j < K; ++j) {
                        const int4 o = s_off[j][p];
                        const float4 w = s_w[j][p];
                        float* gb = gpacked + nb_view[j] + goff;
#pragma unroll
                        for (int i = 0; i < 4; ++i) {
                            const float gw = go[i] * (two_r * wv[j][i] - two_r2 * S[i]);
                            // an outside tap has weight 0 and offset 0: skip it instead of adding 0 to pixel 0
                            if (w.x != 0.0f) atomicAdd(gb + o.x + i, gw * w.x);
                            if (w.y != 0.0f) atomicAdd(gb + o.y + i, gw * w.y);
                            if (w.z != 0.0f) atomicAdd(gb + o.z + i, gw * w.z);
                            if (w.w != 0.0f) atomicAdd(gb + o.w + i, gw * w.w);
                        }
                    }
                }
            }
            __syncthreads();
        }
    }
}

}  // namespace mvsdet

using namespace mvsdet;

extern "C" int mvsdet_plane_sweep_variance_bwd_f32(const float* feat, const int64_t* nbr, const float* proj,
                                                   const float* depth, const float* g, float* gfeat, void* workspace,
                                                   size_t workspace_bytes, int N, int K, int C, int D, int H, int W,
                                                   mvsdet_stream_t stream_) {
    hipStream_t stream = (hipStream_t)stream_;
    MVS_REQUIRE(feat && depth && g && gfeat && workspace, "plane_sweep_variance_bwd: NULL pointer");
    MVS_REQUIRE(K == 0 || (nbr && proj), "plane_sweep_variance_bwd: NULL neighbour arrays with K=%d", K);
    MVS_REQUIRE(N > 0 && C > 0 && D > 0 && H > 1 && W > 1, "plane_sweep_variance_bwd: bad shape");
    MVS_REQUIRE(K >= 0 && K <= MVSDET_MAX_NEIGHBORS, "plane_sweep_variance_bwd: K=%d outside [0,%d]", K, MVSDET_MAX_NEIGHBORS);
    MVS_REQUIRE((size_t)H * W * kSlab < (size_t)INT32_MAX, "plane_sweep_variance_bwd: one slab image exceeds 2^31 elements");
    const size_t pb = mvsdet_packed_bytes(N, C, H, W);
    if (workspace_bytes < 2 * pb) {
        set_error("plane_sweep_variance_bwd: workspace %zu B < %zu B", workspace_bytes, 2 * pb);
        return MVSDET_ERR_WORKSPACE;
    }
    float* packed = (float*)workspace;
    float* gpacked = (float*)((char*)workspace + pb);
    const int64_t fs[4] = {(int64_t)C * H * W, (int64_t)H * W, W, 1};
    if (int rc = mvsdet_pack_features_f32(feat, fs, packed, N, C, H, W, stream_)) return rc;
    if (hipMemsetAsync(gpacked, 0, pb, stream) != hipSuccess) {
        set_error("plane_sweep_variance_bwd: hipMemsetAsync failed");
        return MVSDET_ERR_HIP;
    }
    constexpr int TP = 32;
    const int S = num_slabs(C);
    const int G = 8 * S;  // channel groups of 4, slab padding included
    const int HW = H * W;
    const int tiles = (HW + TP - 1) / TP;
    int lp_log2 = 0;
    while ((1 << lp_log2) < (G < 64 ? G : 64)) ++lp_log2;
    while ((64 >> lp_log2) > TP / 4) ++lp_log2;
    const long long nblocks = (long long)N * tiles;
    MVS_REQUIRE(nblocks <= INT32_MAX, "plane_sweep_variance_bwd: grid too large");
    dim3 grid((unsigned)nblocks);
#define MVS_BWD_CASE(KV)                                                                                            \
    case KV:                                                                                                        \
        hipLaunchKernelGGL((plane_sweep_variance_bwd_kernel<KV, TP>), grid, dim3(kThreads), 0, stream, packed, nbr, \
                           proj, depth, g, gpacked, N, C, G, D, H, W, tiles, lp_log2);                               \
        break;
    switch (K) {
        MVS_BWD_CASE(0)
        MVS_BWD_CASE(1)
        MVS_BWD_CASE(2)
        MVS_BWD_CASE(3)
        MVS_BWD_CASE(4)
    }
#undef MVS_BWD_CASE
    MVS_LAUNCH_CHECK("plane_sweep_variance_bwd");
    dim3 ugrid((HW + 63) / 64, S, N);
    hipLaunchKernelGGL(unpack_features_kernel, ugrid, dim3(kThreads), 0, stream, gpacked, gfeat, C, S, H, W);
    MVS_LAUNCH_CHECK("unpack_features");
    return MVSDET_OK;
}
