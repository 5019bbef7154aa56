// Stage 3 of the MVSDet hot path on gfx950: depth-probability-weighted lifting of 2-D features into
// the voxel grid (a9, backproject_Weigh, mvsdet.py:1372-1492) and its fusion with the per-voxel mean
// over views (a10, mvsdet.py:511-515).  torch_scatter is not involved: the operation is
// voxel-centric -- every voxel looks up ONE pixel per view -- so no atomics are needed going forward.
//
// Roofline: HBM.  Per-view form writes N*C*V*4 B (mostly zeros); the fused mean writes C*V*4 B and reads
// one packed feature column (C*4 B, contiguous) per valid (view, voxel) pair.
#include "common.h"
#include "pack.h"

namespace mvsdet {

__device__ __forceinline__ int clamp_to_i32(float v) {
    if (!(v == v)) return INT32_MIN;
    if (v > 1073741824.0f) return 1073741824;
    if (v < -1073741824.0f) return -1073741824;
    return (int)v;
}

// ---------------------------------------------------------------------------------------------
// a9, per-view volume (API parity with the reference's return value).  One thread per (view, voxel),
// lanes along the voxel axis so that every channel row of the output is written coalesced.
// ---------------------------------------------------------------------------------------------
__global__ __launch_bounds__(kThreads) void backproject_weigh_kernel(
    const float* __restrict__ feat, int64_t fs0, int64_t fs1, int64_t fs2, int64_t fs3,
    const float* __restrict__ points, const float* __restrict__ projection, const float* __restrict__ depth,
    const float* __restrict__ dens, int64_t ds0, int64_t ds1, int64_t ds2, int64_t ds3, float* __restrict__ volume,
    uint8_t* __restrict__ valid, int32_t* __restrict__ xi_out, int32_t* __restrict__ yi_out, int C, int h, int w, int V,
    int J, float vz) {
    const int v = blockIdx.x * kThreads + threadIdx.x;
    const int i = blockIdx.y;
    if (v >= V) return;
    float xr, yr, z, wgt = 0.0f, psum;
    int arg;
    bool ok = project_voxel(projection + (size_t)i * 12, points[v], points[(size_t)V + v], points[2 * (size_t)V + v], h, w,
                            xr, yr, z);
    if (xi_out) xi_out[(size_t)i * V + v] = clamp_to_i32(xr);
    if (yi_out) yi_out[(size_t)i * V + v] = clamp_to_i32(yr);
    int xi = 0, yi = 0;
    if (ok) {
        xi = (int)xr;
        yi = (int)yr;
        ok = depth_window(depth + (int64_t)i * ds0, dens + (int64_t)i * ds0, ds1, ds2, ds3, J, yi, xi, z, vz, wgt, psum, arg);
    }
    valid[(size_t)i * V + v] = ok ? 1 : 0;
    const float* src = feat + (int64_t)i * fs0 + (int64_t)yi * fs2 + (int64_t)xi * fs3;
    float* o = volume + (size_t)i * C * V + v;
    for (int c = 0; c < C; ++c) {
        float val = 0.0f;
        if (ok) val = src[(int64_t)c * fs1] * wgt;
        o[(size_t)c * V] = val;
    }
}

// ---------------------------------------------------------------------------------------------
// a9+a10 fused: volume_mean (C,V) and the valid-view count (V), never materialising (N,C,V).
//
//   block   = TP consecutive voxels
//   phase 1 every (view, voxel) pair of a 64-view chunk is projected and depth-tested by one thread;
//           survivors leave {packed pixel offset, weight} in LDS and set their bit in the voxel's view mask
//   phase 2 lane = (voxel slot, channel group): walks the set bits in ascending view order (same order as
//           the oracle: bit-identical sums) and accumulates feature column * weight with 16-byte loads
//   phase 3 (after all view chunks) mean = acc / (count + 1e-8) -> LDS tile [channel][voxel] -> rows out
// ---------------------------------------------------------------------------------------------
template <int TP>
__global__ __launch_bounds__(kThreads) void backproject_mean_kernel(
    const float* __restrict__ packed, const float* __restrict__ points, const float* __restrict__ projection,
    const float* __restrict__ depth, const float* __restrict__ dens, int64_t ds0, int64_t ds1, int64_t ds2, int64_t ds3,
    float* __restrict__ mean, int32_t* __restrict__ count, int N, int C, int G, int H, int W, int h, int w, int V, int J,
    float vz, int lp_log2, int normalize) {
    constexpr int PW = TP / 4;
    constexpr int MAXSTEPS = PW;  // PPI >= 1
    __shared__ float s_tile[256 * (TP + 1)];
    __shared__ unsigned long long s_mask[TP];
    __shared__ int s_off[64][TP];
    __shared__ float s_wt[64][TP];

    const int v0 = blockIdx.x * TP;
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int LP = 1 << lp_log2, PPI = 64 >> lp_log2;
    const int gl = lane & (LP - 1), ps = lane >> lp_log2;
    const int steps = PW / PPI;
    // packed layout (pack.h): [view][slab][pixel][32]; channel group gg = 8*slab + g
    const size_t slab_stride = (size_t)H * W * kSlab;
    const size_t view_stride = slab_stride * (size_t)(G / 8);
    const int chunks = (G + 63) / 64;

    for (int ci = 0; ci < chunks; ++ci) {
        const int rg = min(64, G - ci * 64);
        const bool gvalid = gl < rg;
        const int gg_ = ci * 64 + (gvalid ? gl : 0);
        const size_t goff = (size_t)(gg_ >> 3) * slab_stride + 4 * (gg_ & 7);
        float acc[MAXSTEPS][4];
        int cnt[MAXSTEPS];
#pragma unroll
        for (int s = 0; s < MAXSTEPS; ++s) {
            acc[s][0] = acc[s][1] = acc[s][2] = acc[s][3] = 0.0f;
            cnt[s] = 0;
        }
        for (int vc = 0; vc < N; vc += 64) {
            const int nv = min(64, N - vc);
            __syncthreads();  // previous chunk's tables / tile fully consumed
            if (threadIdx.x < TP) s_mask[threadIdx.x] = 0ull;
            __syncthreads();
            // ---- phase 1
            for (int idx = threadIdx.x; idx < nv * TP; idx += kThreads) {
                const int il = idx / TP, p = idx - il * TP;
                const int v = v0 + p, i = vc + il;
                if (v < V) {
                    float xr, yr, z, wgt, psum;
                    int arg;
                    bool ok = project_voxel(projection + (size_t)i * 12, points[v], points[(size_t)V + v],
                                            points[2 * (size_t)V + v], h, w, xr, yr, z);
                    if (ok) {
                        const int xi = (int)xr, yi = (int)yr;
                        ok = depth_window(depth + (int64_t)i * ds0, dens + (int64_t)i * ds0, ds1, ds2, ds3, J, yi, xi, z,
                                          vz, wgt, psum, arg);
                        if (ok) {
                            s_off[il][p] = (yi * W + xi) * kSlab;
                            s_wt[il][p] = wgt;
                            atomicOr(&s_mask[p], 1ull << il);
                        }
                    }
                }
            }
            __syncthreads();
            // ---- phase 2
#pragma unroll
            for (int s = 0; s < MAXSTEPS; ++s) {
                if (s < steps) {
                    const int p = wave * PW + s * PPI + ps;
                    unsigned long long m = s_mask[p];
                    cnt[s] += __popcll(m);
                    while (m) {
                        const int il = __ffsll((long long)m) - 1;
                        m &= m - 1;
                        const float wt = s_wt[il][p];
                        const float4 f = *reinterpret_cast<const float4*>(packed + (size_t)(vc + il) * view_stride +
                                                                          s_off[il][p] + goff);
                        // mvsdet.py:1459-1460 then :511  (feature * weight, then summed over views)
                        acc[s][0] = acc[s][0] + f.x * wt;
                        acc[s][1] = acc[s][1] + f.y * wt;
                        acc[s][2] = acc[s][2] + f.z * wt;
                        acc[s][3] = acc[s][3] + f.w * wt;
                    }
                }
            }
        }
        // ---- phase 3
#pragma unroll
        for (int s = 0; s < MAXSTEPS; ++s) {
            if (s < steps) {
                const int p = wave * PW + s * PPI + ps;
                const float den = (float)cnt[s] + 1e-8f;  // mvsdet.py:514
                if (gvalid) {
                    float* t = s_tile + gl * (TP + 1) + p;
#pragma unroll
                    for (int i = 0; i < 4; ++i) t[i * rg * (TP + 1)] = !normalize ? acc[s][i] : (cnt[s] > 0 ? acc[s][i] / den : 0.0f);
                }
                if (ci == 0 && gl == 0 && v0 + p < V) count[v0 + p] = cnt[s];
            }
        }
        __syncthreads();
        {
            constexpr int RPI = 64 / TP;
            const int pp = lane % TP, rsub = lane / TP;
            const int rows = 4 * rg;
            const bool pvalid = v0 + pp < V;
            for (int r = wave * RPI + rsub; r < rows; r += 4 * RPI) {
                const int i = r / rg, gg = r - i * rg;
                const int gq = ci * 64 + gg;
                const int c = (gq >> 3) * kSlab + 8 * i + (gq & 7);
                if (c < C && pvalid) mean[(size_t)c * V + v0 + pp] = s_tile[r * (TP + 1) + pp];
            }
        }
    }
}

// ---------------------------------------------------------------------------------------------
// backward of a9 (and of a9+a10 when `count` is given: the incoming gradient of view i's volume is then
// g[c,v] / (count[v] + 1e-8)).  Several voxels fall on the same pixel, so gradients are accumulated with
// float atomics into zero-initialised dense buffers.
//   dL/dfeat[i,c,y,x] += go * weight
//   dL/ddens[i,j,y,x] += (sum_c go*feat) * (delta(j,arg) - weight) / psum      (weight = dens_arg / psum)
// ---------------------------------------------------------------------------------------------
__global__ __launch_bounds__(kThreads) void backproject_weigh_bwd_kernel(
    const float* __restrict__ feat, int64_t fs0, int64_t fs1, int64_t fs2, int64_t fs3,
    const float* __restrict__ points, const float* __restrict__ projection, const float* __restrict__ depth,
    const float* __restrict__ dens, int64_t ds0, int64_t ds1, int64_t ds2, int64_t ds3,
    const int32_t* __restrict__ count, const float* __restrict__ g, float* __restrict__ gfeat,
    float* __restrict__ gdens, int C, int h, int w, int V, int J, float vz) {
    const int v = blockIdx.x * kThreads + threadIdx.x;
    const int i = blockIdx.y;
    if (v >= V) return;
    float xr, yr, z, wgt = 0.0f, psum = 1.0f;
    int arg = -1;
    bool ok = project_voxel(projection + (size_t)i * 12, points[v], points[(size_t)V + v], points[2 * (size_t)V + v], h, w,
                            xr, yr, z);
    if (!ok) return;
    const int xi = (int)xr, yi = (int)yr;
    ok = depth_window(depth + (int64_t)i * ds0, dens + (int64_t)i * ds0, ds1, ds2, ds3, J, yi, xi, z, vz, wgt, psum, arg);
    if (!ok) return;
    float scale = 1.0f;
    const float* gp;
    size_t gstride;
    if (count) {  // fused-mean form: g is (C,V)
        scale = 1.0f / ((float)count[v] + 1e-8f);
        gp = g + v;
        gstride = (size_t)V;
    } else {  // per-view form: g is (N,C,V)
        gp = g + (size_t)i * C * V + v;
        gstride = (size_t)V;
    }
    const float* src = feat + (int64_t)i * fs0 + (int64_t)yi * fs2 + (int64_t)xi * fs3;
    float* gf = gfeat + ((size_t)i * C * h + yi) * w + xi;
    float gw = 0.0f;
    for (int c = 0; c < C; ++c) {
        const float go = gp[(size_t)c * gstride] * scale;
        atomicAdd(gf + (size_t)c * h * w, go * wgt);
        gw = fmaf(go, src[(int64_t)c * fs1], gw);
    }
    if (arg >= 0) {
        float* gd = gdens + ((size_t)i * J * h + yi) * w + xi;
        for (int j = 0; j < J; ++j) {
            const float dj = ((j == arg) ? 1.0f : 0.0f) - wgt;
            atomicAdd(gd + (size_t)j * h * w, gw * dj / psum);
        }
    }
}

}  // namespace mvsdet

using namespace mvsdet;

static int check_stage3(const char* name, int N, int C, int h, int w, int V, int J) {
    MVS_REQUIRE(N > 0 && C > 0 && h > 0 && w > 0 && V > 0, "%s: bad shape N=%d C=%d h=%d w=%d V=%d", name, N, C, h, w, V);
    MVS_REQUIRE(J >= 1 && J <= MVSDET_MAX_TOPK, "%s: J=%d outside [1,%d]", name, J, MVSDET_MAX_TOPK);
    MVS_REQUIRE(N <= 65535, "%s: N > 65535", name);
    return MVSDET_OK;
}

extern "C" int mvsdet_backproject_weigh_f32(const float* feat, const int64_t* fs, const float* points,
                                            const float* projection, const float* depth, const float* dens,
                                            const int64_t* ds, float* volume, uint8_t* valid, int32_t* xi, int32_t* yi,
                                            int N, int C, int h, int w, int V, int J, float vz, mvsdet_stream_t stream) {
    MVS_REQUIRE(feat && fs && points && projection && depth && dens && ds && volume && valid, "backproject_weigh: NULL pointer");
    if (int rc = check_stage3("backproject_weigh", N, C, h, w, V, J)) return rc;
    dim3 grid((V + kThreads - 1) / kThreads, N);
    hipLaunchKernelGGL(backproject_weigh_kernel, grid, dim3(kThreads), 0, (hipStream_t)stream, feat, fs[0], fs[1], fs[2],
                       fs[3], points, projection, depth, dens, ds[0], ds[1], ds[2], ds[3], volume, valid, xi, yi, C, h, w, V,
                       J, vz);
    MVS_LAUNCH_CHECK("backproject_weigh");
    return MVSDET_OK;
}

static int launch_stage3_fused(const char* name, const float* packed, const float* points, const float* projection,
                               const float* depth, const float* dens, const int64_t* ds, float* out, int32_t* count, int N,
                               int C, int H, int W, int h, int w, int V, int J, float vz, mvsdet_stream_t stream,
                               int normalize) {
    MVS_REQUIRE(packed && points && projection && depth && dens && ds && out && count, "%s: NULL pointer", name);
    if (int rc = check_stage3(name, N, C, h, w, V, J)) return rc;
    MVS_REQUIRE(h <= H && w <= W, "%s: crop %dx%d exceeds map %dx%d", name, h, w, H, W);
    MVS_REQUIRE((size_t)H * W * kSlab < (size_t)INT32_MAX, "%s: one slab image exceeds 2^31 elements", name);
    constexpr int TP = 32;
    const int G = 8 * num_slabs(C);  // channel groups of 4, slab padding included
    int lp_log2 = 0;
    while ((1 << lp_log2) < (G < 64 ? G : 64)) ++lp_log2;
    while ((64 >> lp_log2) > TP / 4) ++lp_log2;
    dim3 grid((V + TP - 1) / TP);
    hipLaunchKernelGGL((backproject_mean_kernel<TP>), grid, dim3(kThreads), 0, (hipStream_t)stream, packed, points,
                       projection, depth, dens, ds[0], ds[1], ds[2], ds[3], out, count, N, C, G, H, W, h, w, V, J, vz,
                       lp_log2, normalize);
    MVS_LAUNCH_CHECK(name);
    return MVSDET_OK;
}

extern "C" int mvsdet_backproject_weigh_mean_packed_f32(const float* packed, const float* points, const float* projection,
                                                        const float* depth, const float* dens, const int64_t* ds,
                                                        float* mean, int32_t* count, int N, int C, int H, int W, int h,
                                                        int w, int V, int J, float vz, mvsdet_stream_t stream) {
    return launch_stage3_fused("backproject_weigh_mean", packed, points, projection, depth, dens, ds, mean, count, N, C, H,
                               W, h, w, V, J, vz, stream, 1);
}

extern "C" int mvsdet_backproject_weigh_sum_packed_f32(const float* packed, const float* points, const float* projection,
                                                       const float* depth, const float* dens, const int64_t* ds,
                                                       float* sum, int32_t* count, int N, int C, int H, int W, int h,
                                                       int w, int V, int J, float vz, mvsdet_stream_t stream) {
    return launch_stage3_fused("backproject_weigh_sum", packed, points, projection, depth, dens, ds, sum, count, N, C, H,
                               W, h, w, V, J, vz, stream, 0);
}

static int launch_stage3_bwd(const char* name, const float* feat, const int64_t* fs, const float* points,
                             const float* projection, const float* depth, const float* dens, const int64_t* ds,
                             const int32_t* count, const float* g, float* gfeat, float* gdens, int N, int C, int h, int w,
                             int V, int J, float vz, hipStream_t stream) {
    MVS_REQUIRE(feat && fs && points && projection && depth && dens && ds && g && gfeat && gdens, "%s: NULL pointer", name);
    if (int rc = check_stage3(name, N, C, h, w, V, J)) return rc;
    if (hipMemsetAsync(gfeat, 0, (size_t)N * C * h * w * sizeof(float), stream) != hipSuccess ||
        hipMemsetAsync(gdens, 0, (size_t)N * J * h * w * sizeof(float), stream) != hipSuccess) {
        set_error("%s: hipMemsetAsync failed", name);
        return MVSDET_ERR_HIP;
    }
    dim3 grid((V + kThreads - 1) / kThreads, N);
    hipLaunchKernelGGL(backproject_weigh_bwd_kernel, grid, dim3(kThreads), 0, stream, feat, fs[0], fs[1], fs[2], fs[3], points,
                       projection, depth, dens, ds[0], ds[1], ds[2], ds[3], count, g, gfeat, gdens, C, h, w, V, J, vz);
    MVS_LAUNCH_CHECK(name);
    return MVSDET_OK;
}

extern "C" int mvsdet_backproject_weigh_bwd_f32(const float* feat, const int64_t* fs, const float* points,
                                                const float* projection, const float* depth, const float* dens,
                                                const int64_t* ds, const float* g, float* gfeat, float* gdens, int N, int C,
                                                int h, int w, int V, int J, float vz, mvsdet_stream_t stream) {
    return launch_stage3_bwd("backproject_weigh_bwd", feat, fs, points, projection, depth, dens, ds, nullptr, g, gfeat,
                             gdens, N, C, h, w, V, J, vz, (hipStream_t)stream);
}

extern "C" int mvsdet_backproject_weigh_mean_bwd_f32(const float* feat, const int64_t* fs, const float* points,
                                                     const float* projection, const float* depth, const float* dens,
                                                     const int64_t* ds, const int32_t* count, const float* g, float* gfeat,
                                                     float* gdens, int N, int C, int h, int w, int V, int J, float vz,
                                                     mvsdet_stream_t stream) {
    MVS_REQUIRE(count, "backproject_weigh_mean_bwd: NULL count");
    return launch_stage3_bwd("backproject_weigh_mean_bwd", feat, fs, points, projection, depth, dens, ds, count, g, gfeat,
                             gdens, N, C, h, w, V, J, vz, (hipStream_t)stream);
}
