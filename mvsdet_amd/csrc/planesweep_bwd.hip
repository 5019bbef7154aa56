// Backward of the fused plane-sweep variance (a3+a4) with respect to the 2-D features -- the only input
// with a gradient: the sampling grid is built under torch.no_grad() (mvs_models/module.py:115).
//
//   var = Q/(K+1) - (S/(K+1))^2,  S = f + sum_j w_j,  Q = f^2 + sum_j w_j^2
//   dvar/dv = 2 v r - 2 S r^2   (r = 1/(K+1)) for each contributing value v in {f, w_1..w_K}
//   w_j = sum_t weight_t * tap_t  ->  dL/dtap_t += weight_t * dL/dw_j   (bilinear scatter)
//
// Same decomposition as the forward slab kernel (sweep_kernel.h): block = (reference view, 128-pixel tile,
// 32-channel slab), lanes = (pixel slot, channel group), the channel-independent sampling table and the tap
// bounding boxes come from plane_sweep_coords_kernel.  Per plane:
//   1. the incoming gradient tile is read as whole 128-byte rows of (N,C,D,H,W) and transposed through LDS;
//   2. the warped values are recomputed (taps gathered from the slab images, which sit in the XCD's L2);
//   3. per neighbour the tap gradients are accumulated with LDS atomics into a gradient image of the footprint
//      box and then flushed to the packed gradient map with fp32 global atomics shaped as 256 contiguous bytes
//      per wave-instruction -- about 3x fewer and far better shaped atomics than one per tap
//      (MI355X_MICROARCH "Global float atomics": full rate only for contiguous 256-byte instructions);
//      a footprint that does not fit the LDS box falls back to per-tap global atomics;
//   4. the reference view's own term is kept in registers across the planes and added once per block.
// Bound: the chip-wide float-atomic rate.  The packed gradient map is unpacked to (N,C,H,W) afterwards.
#include "common.h"

#include <algorithm>

#include "pack.h"
#include "sweep_kernel.h"

namespace mvsdet {

template <int K, int TW>
__global__ __launch_bounds__(kThreads, 3) void plane_sweep_variance_bwd_kernel(
    const float* __restrict__ packed, const int64_t* __restrict__ nbr, const float2* __restrict__ table,
    const int4* __restrict__ boxes, const float* __restrict__ gvar, float* __restrict__ gpacked, int N, int C, int S,
    int D, int H, int W, int tiles_x, int tiles) {
    constexpr int KK = K > 0 ? K : 1;
    constexpr int TH = kTilePix / TW;
    constexpr int ITER = (KK * kTilePix + kThreads - 1) / kThreads;
    __shared__ float4 s_g4[kBoxCap * 8];      // gradient image of one neighbour's footprint box; first the dL/dvar tile
    __shared__ int2 s_xy[KK][kTilePix];       // tap origin per (neighbour, pixel)
    __shared__ float4 s_w[KK][kTilePix];      // tap weights
    float* s_g = reinterpret_cast<float*>(s_g4);

    const int HW = H * W;
    const int id = blockIdx.x;
    const int slab = id % S;
    const int bt = id / S;
    const int tile = bt % tiles, n = bt / tiles;
    const int tx0 = (tile % tiles_x) * TW, ty0 = (tile / tiles_x) * TH;
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int g = lane & 7, ps = lane >> 3;
    const size_t slab_stride = (size_t)HW * kSlab;
    const float* ref_img = packed + ((size_t)n * S + slab) * slab_stride;
    const float4* nb_img[KK];
    float* nb_grad[KK];
#pragma unroll
    for (int j = 0; j < K; ++j) {
        int64_t v = nbr[(size_t)n * K + j];
        v = v < 0 ? 0 : (v >= N ? N - 1 : v);
        nb_img[j] = reinterpret_cast<const float4*>(packed + ((size_t)v * S + slab) * slab_stride);
        nb_grad[j] = gpacked + ((size_t)v * S + slab) * slab_stride;
    }
    const float r = 1.0f / (float)(K + 1);
    const float two_r = 2.0f * r, two_r2 = 2.0f * r * r;

    // the lane's 4 pixels (one per step), their reference features and the running reference-term gradient
    float4 f[4];
    int ppix[4];
    bool pok[4];
    float gref[4][4];
#pragma unroll
    for (int s = 0; s < 4; ++s) {
        const int p = (wave * 4 + s) * 8 + ps;
        const int x = tx0 + (p % TW), y = ty0 + (p / TW);
        pok[s] = (x < W) && (y < H);
        ppix[s] = min(y, H - 1) * W + min(x, W - 1);
        f[s] = *reinterpret_cast<const float4*>(ref_img + (size_t)ppix[s] * kSlab + 4 * g);
        gref[s][0] = gref[s][1] = gref[s][2] = gref[s][3] = 0.0f;
    }
    // row loader of the dL/dvar tile: float4 slot sq of the 128-pixel tile, channel rows 8*wave + 2*k + sh
    const int sq = lane & 31, sh = lane >> 5;
    const int st_x = tx0 + (sq % (TW / 4)) * 4, st_y = ty0 + sq / (TW / 4);
    const int st_off = st_y * W + st_x;
    const int st_n = (st_y < H) ? max(0, min(4, W - st_x)) : 0;
    const bool st_vec = (st_n == 4) && ((W & 3) == 0) && ((HW & 3) == 0);

    for (int d = 0; d < D; ++d) {
        const size_t tbase = ((size_t)bt * D + d) * K;
        // ---- 1. dL/dvar tile -> LDS [channel row][pixel]; table entries -> tap origin + weights
        {
            float* t = s_g + (wave * 8 + sh) * kTileStride + 4 * sq;
#pragma unroll
            for (int k = 0; k < 4; ++k) {
                const int c = slab * kSlab + wave * 8 + 2 * k + sh;
                float4 v = make_float4(0.f, 0.f, 0.f, 0.f);
                if (c < C) {
                    const float* src = gvar + (((size_t)n * C + c) * D + d) * HW + st_off;
                    if (st_vec) {
                        v = *reinterpret_cast<const float4*>(src);
                    } else {
                        if (st_n > 0) v.x = src[0];
                        if (st_n > 1) v.y = src[1];
                        if (st_n > 2) v.z = src[2];
                        if (st_n > 3) v.w = src[3];
                    }
                }
                *reinterpret_cast<float4*>(t + 2 * k * kTileStride) = v;
            }
        }
#pragma unroll
        for (int it = 0; it < ITER; ++it) {
            const int j = __builtin_amdgcn_readfirstlane((it * kThreads + tid) / kTilePix);
            if (j < K) {
                const int p = tid % kTilePix;
                const float2 e = table[(tbase + j) * kTilePix + p];
                const SampleTaps tp = decode_sample(e.x, e.y, H, W);
                s_w[j][p] = tap_weights(tp);
                s_xy[j][p] = make_int2(tp.x0, tp.y0);
            }
        }
        __syncthreads();
        float go[4][4];
#pragma unroll
        for (int s = 0; s < 4; ++s) {
            const float* t = s_g + g * kTileStride + (wave * 4 + s) * 8 + ps;
#pragma unroll
            for (int i = 0; i < 4; ++i) go[s][i] = pok[s] ? t[8 * i * kTileStride] : 0.0f;
        }
        // ---- 2. recompute the warped values (taps from the slab images) and S
        bool skip[KK];  // neighbour entirely outside the source image (and all positions finite): w_j == 0, no gradient
#pragma unroll
        for (int j = 0; j < K; ++j) {
            const int4 bx = boxes[tbase + j];
            skip[j] = (__builtin_amdgcn_readfirstlane(bx.y) < __builtin_amdgcn_readfirstlane(bx.x)) &&
                      (__builtin_amdgcn_readfirstlane(bx.w) == kBoxSkip);
        }
        float S_[4][4], wv[KK][4][4];
#pragma unroll
        for (int s = 0; s < 4; ++s) { S_[s][0] = f[s].x; S_[s][1] = f[s].y; S_[s][2] = f[s].z; S_[s][3] = f[s].w; }
#pragma unroll
        for (int j = 0; j < K; ++j) {
            if (skip[j]) {
#pragma unroll
                for (int s = 0; s < 4; ++s) wv[j][s][0] = wv[j][s][1] = wv[j][s][2] = wv[j][s][3] = 0.0f;
                continue;
            }
#pragma unroll
            for (int s = 0; s < 4; ++s) {
                const int p = (wave * 4 + s) * 8 + ps;
                const int2 xy = s_xy[j][p];
                const float4 w = s_w[j][p];
                const int xa = clampi(xy.x, 0, W - 1), xb = clampi(xy.x + 1, 0, W - 1);
                const int ya = clampi(xy.y, 0, H - 1) * W, yb = clampi(xy.y + 1, 0, H - 1) * W;
                const float4* b = nb_img[j] + g;
                const float4 t0 = b[(ya + xa) * 8], t1 = b[(ya + xb) * 8], t2 = b[(yb + xa) * 8], t3 = b[(yb + xb) * 8];
                const float a0[4] = {t0.x, t0.y, t0.z, t0.w}, a1[4] = {t1.x, t1.y, t1.z, t1.w};
                const float a2[4] = {t2.x, t2.y, t2.z, t2.w}, a3[4] = {t3.x, t3.y, t3.z, t3.w};
#pragma unroll
                for (int i = 0; i < 4; ++i) {
                    float v = a0[i] * w.x;
                    v = fmaf(a1[i], w.y, v);
                    v = fmaf(a2[i], w.z, v);
                    v = fmaf(a3[i], w.w, v);
                    wv[j][s][i] = v;
                    S_[s][i] += v;
                }
            }
        }
        // ---- 4. reference term, kept in registers across the planes
#pragma unroll
        for (int s = 0; s < 4; ++s) {
            const float fv[4] = {f[s].x, f[s].y, f[s].z, f[s].w};
#pragma unroll
            for (int i = 0; i < 4; ++i) gref[s][i] = fmaf(go[s][i], two_r * fv[i] - two_r2 * S_[s][i], gref[s][i]);
        }
        // ---- 3. tap gradients of every neighbour
#pragma unroll
        for (int j = 0; j < K; ++j) {
            const int4 bx = boxes[tbase + j];
            const int bx0 = __builtin_amdgcn_readfirstlane(bx.x), bx1 = __builtin_amdgcn_readfirstlane(bx.y);
            const int by0 = __builtin_amdgcn_readfirstlane(bx.z), by1 = __builtin_amdgcn_readfirstlane(bx.w);
            const int nc = bx1 - bx0 + 1, nr = by1 - by0 + 1;
            const bool nonempty = (bx1 >= bx0) && (by1 >= by0);
            const bool boxed = nonempty && (nc * nr <= kBoxCap);
            __syncthreads();  // tile (first neighbour) / previous flush fully read
            if (boxed) {
                const int nfl = nc * nr * kSlab;
                for (int e = tid; e < nfl / 4; e += kThreads) s_g4[e] = make_float4(0.f, 0.f, 0.f, 0.f);
                __syncthreads();
#pragma unroll
                for (int s = 0; s < 4; ++s) {
                    const int p = (wave * 4 + s) * 8 + ps;
                    const int2 xy = s_xy[j][p];
                    const float4 w = s_w[j][p];
                    const float wt[4] = {w.x, w.y, w.z, w.w};
                    // invalid taps carry weight 0: adding 0 to a clamped texel of the box is harmless
                    const int xa = clampi(xy.x, bx0, bx1) - bx0, xb = clampi(xy.x + 1, bx0, bx1) - bx0;
                    const int ya = (clampi(xy.y, by0, by1) - by0) * nc, yb = (clampi(xy.y + 1, by0, by1) - by0) * nc;
                    const int to[4] = {(ya + xa) * kSlab, (ya + xb) * kSlab, (yb + xa) * kSlab, (yb + xb) * kSlab};
#pragma unroll
                    for (int i = 0; i < 4; ++i) {
                        const float gw = go[s][i] * (two_r * wv[j][s][i] - two_r2 * S_[s][i]);
#pragma unroll
                        for (int t = 0; t < 4; ++t)
                            if (wt[t] != 0.0f) atomicAdd(s_g + to[t] + 4 * g + i, gw * wt[t]);
                    }
                }
                __syncthreads();
                // flush: rows of the box are contiguous nc*32 floats in the packed gradient image
                const int row_fl = nc * kSlab;
                for (int row = wave; row < nr; row += 4) {
                    float* dst = nb_grad[j] + ((size_t)(by0 + row) * W + bx0) * kSlab;
                    const float* src = s_g + row * row_fl;
                    for (int q = lane; q < row_fl; q += 64) {
                        const float v = src[q];
                        if (v != 0.0f) atomicAdd(dst + q, v);
                    }
                }
            } else if (nonempty) {  // footprint larger than the LDS box: one global atomic per tap
#pragma unroll
                for (int s = 0; s < 4; ++s) {
                    const int p = (wave * 4 + s) * 8 + ps;
                    const int2 xy = s_xy[j][p];
                    const float4 w = s_w[j][p];
                    const float wt[4] = {w.x, w.y, w.z, w.w};
                    const int xa = clampi(xy.x, 0, W - 1), xb = clampi(xy.x + 1, 0, W - 1);
                    const int ya = clampi(xy.y, 0, H - 1) * W, yb = clampi(xy.y + 1, 0, H - 1) * W;
                    const int to[4] = {(ya + xa) * kSlab, (ya + xb) * kSlab, (yb + xa) * kSlab, (yb + xb) * kSlab};
#pragma unroll
                    for (int i = 0; i < 4; ++i) {
                        const float gw = go[s][i] * (two_r * wv[j][s][i] - two_r2 * S_[s][i]);
#pragma unroll
                        for (int t = 0; t < 4; ++t)
                            if (wt[t] != 0.0f) atomicAdd(nb_grad[j] + to[t] + 4 * g + i, gw * wt[t]);
                    }
                }
            }
        }
        __syncthreads();  // last flush / tables fully read before the next plane overwrites them
    }
    // ---- reference term: once per block
    float* gr = gpacked + ((size_t)n * S + slab) * slab_stride;
#pragma unroll
    for (int s = 0; s < 4; ++s)
        if (pok[s]) {
#pragma unroll
            for (int i = 0; i < 4; ++i) atomicAdd(gr + (size_t)ppix[s] * kSlab + 4 * g + i, gref[s][i]);
        }
}

}  // namespace mvsdet

using namespace mvsdet;

// defined in planesweep.hip
extern "C" size_t mvsdet_plane_sweep_scratch_bytes(int N, int K, int D, int H, int W);
namespace mvsdet {
size_t sweep_table_bytes(int N, int K, int D, int H, int W);
int sweep_tile_width(int W);  // the tile shape the sweep geometry is built for
int sweep_build_geometry_and_table(const float* proj, const float* depth, void* scratch, size_t scratch_bytes, void* table,
                                   int N, int K, int D, int H, int W, mvsdet_stream_t stream);
}

extern "C" size_t mvsdet_plane_sweep_bwd_workspace_bytes(int N, int K, int C, int D, int H, int W) {
    const size_t pb = (mvsdet_packed_bytes(N, C, H, W) + 255) / 256 * 256;
    const size_t sb = (mvsdet_plane_sweep_scratch_bytes(N, K, D, H, W) + 255) / 256 * 256;
    return 2 * pb + sb + sweep_table_bytes(N, K, D, H, W);
}

extern "C" int mvsdet_plane_sweep_variance_bwd_f32(const float* feat, const int64_t* nbr, const float* proj,
                                                   const float* depth, const float* g, float* gfeat, void* workspace,
                                                   size_t workspace_bytes, int N, int K, int C, int D, int H, int W,
                                                   mvsdet_stream_t stream_) {
    hipStream_t stream = (hipStream_t)stream_;
    MVS_REQUIRE(feat && depth && g && gfeat && workspace, "plane_sweep_variance_bwd: NULL pointer");
    MVS_REQUIRE(K == 0 || (nbr && proj), "plane_sweep_variance_bwd: NULL neighbour arrays with K=%d", K);
    MVS_REQUIRE(N > 0 && C > 0 && D > 0 && H > 1 && W > 1, "plane_sweep_variance_bwd: bad shape");
    MVS_REQUIRE(K >= 0 && K <= MVSDET_MAX_NEIGHBORS, "plane_sweep_variance_bwd: K=%d outside [0,%d]", K, MVSDET_MAX_NEIGHBORS);
    MVS_REQUIRE(D <= 65535 && H < 65535 && W < 65535, "plane_sweep_variance_bwd: D, H or W > 65534");
    MVS_REQUIRE((size_t)H * W * kSlab < (size_t)INT32_MAX, "plane_sweep_variance_bwd: one slab image exceeds 2^31 elements");
    const size_t pb = (mvsdet_packed_bytes(N, C, H, W) + 255) / 256 * 256;
    const size_t sb = (mvsdet_plane_sweep_scratch_bytes(N, K, D, H, W) + 255) / 256 * 256;
    const size_t tb = sweep_table_bytes(N, K, D, H, W);
    if (workspace_bytes < 2 * pb + sb + tb) {
        set_error("plane_sweep_variance_bwd: workspace %zu B < %zu B", workspace_bytes, 2 * pb + sb + tb);
        return MVSDET_ERR_WORKSPACE;
    }
    float* packed = (float*)workspace;
    float* gpacked = (float*)((char*)workspace + pb);
    void* scratch = (char*)workspace + 2 * pb;
    void* table_mem = (char*)workspace + 2 * pb + sb;
    const int64_t fs[4] = {(int64_t)C * H * W, (int64_t)H * W, W, 1};
    if (int rc = mvsdet_pack_features_f32(feat, fs, packed, N, C, H, W, stream_)) return rc;
    if (hipMemsetAsync(gpacked, 0, pb, stream) != hipSuccess) {
        set_error("plane_sweep_variance_bwd: hipMemsetAsync failed");
        return MVSDET_ERR_HIP;
    }
    if (K > 0)
        if (int rc = sweep_build_geometry_and_table(proj, depth, scratch, sb, table_mem, N, K, D, H, W, stream_)) return rc;
    const int tw = sweep_tile_width(W);
    const int th = kTilePix / tw;
    const int S = num_slabs(C);
    const int HW = H * W;
    const int tiles_x = (W + tw - 1) / tw, tiles = tiles_x * ((H + th - 1) / th);
    const long long nblocks = (long long)N * tiles * S;
    MVS_REQUIRE(nblocks <= INT32_MAX, "plane_sweep_variance_bwd: grid too large");
    const float2* table = reinterpret_cast<const float2*>(table_mem);
    const int4* boxes = sweep_geometry(scratch, N, K, D, tiles).boxes;
    dim3 grid((unsigned)nblocks);
#define MVS_BWD_CASE(KV)                                                                                               \
    case KV:                                                                                                           \
        if (tw == 16)                                                                                                  \
            hipLaunchKernelGGL((plane_sweep_variance_bwd_kernel<KV, 16>), grid, dim3(kThreads), 0, stream, packed, nbr, \
                               table, boxes, g, gpacked, N, C, S, D, H, W, tiles_x, tiles);                            \
        else                                                                                                           \
            hipLaunchKernelGGL((plane_sweep_variance_bwd_kernel<KV, 32>), grid, dim3(kThreads), 0, stream, packed, nbr, \
                               table, boxes, g, gpacked, N, C, S, D, H, W, tiles_x, tiles);                            \
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
