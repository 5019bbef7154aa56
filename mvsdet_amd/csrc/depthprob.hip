// Stage 2 of the MVSDet hot path on gfx950: depth probability (softmax over planes), offset sigmoid,
// top-k plane selection and depth expectation, fused.  Reference: mvsdet.py:470-475, :266-283
// (sample_depth_prob), :298-317 (compute_avg_depth) of Pixie8888/MVSDet.
//
// Roofline: HBM.  Algorithmic bytes per scene: read 2*N*D*H*W*4, write N*(2D+2*topk+1)*H*W*4.
// One thread per pixel, lanes along W so every plane access is a coalesced 256-byte run.  DREG > 0 (D <= DREG, DREG =
// 16 / 64): the D cost logits, then their exponentials, stay in registers -- every input is read once, one expf per
// plane instead of two, and all loads of a pass are in flight together.  DREG = 0 (D up to 512): the logits are re-read
// from L2 in the second / third pass.  Same operations in the same order either way.
#include "common.h"

namespace mvsdet {

// KT = length of the candidate list kept per pixel (3 for the reference's topk = 3, MVSDET_MAX_TOPK otherwise): the
// insertion is 7 VALU instructions per entry and plane, and the kernel is VALU-bound (r02_stage_kernels_pmc.txt).
template <bool kFromLogits, int DREG, int KT>
__global__ __launch_bounds__(kThreads) void depth_prob_topk_kernel(
    const float* __restrict__ cost_reg, const float* __restrict__ off_logit, float* __restrict__ prob,
    float* __restrict__ off, float* __restrict__ est_depth, float* __restrict__ est_dens,
    int32_t* __restrict__ est_idx, float* __restrict__ avg_depth, int D, int HW, int topk, float near, float interval,
    size_t in_view_stride) {
    const int pix = blockIdx.x * kThreads + threadIdx.x;
    const int n = blockIdx.y;
    if (pix >= HW) return;
    const size_t base = (size_t)n * D * HW + pix;
    // the two inputs may be the channel slices of one (N, 2, D, H, W) tensor (the network's output, mvsdet.py:469): a view
    // is in_view_stride floats after the previous one, its (D, H, W) block dense
    const float* c = cost_reg + (size_t)n * in_view_stride + pix;
    const float* o = off_logit + (size_t)n * in_view_stride + pix;

    constexpr int kReg = DREG > 0 ? DREG : 1;
    float e[kReg];   // logits, then exp(logit - max)
    float m = 0.0f, s = 1.0f;
    if (kFromLogits) {
        if (DREG > 0) {
#pragma unroll
            for (int d = 0; d < DREG; ++d) e[d] = d < D ? c[(size_t)d * HW] : 0.0f;
            m = e[0];
#pragma unroll
            for (int d = 1; d < DREG; ++d)
                if (d < D) m = e[d] > m ? e[d] : m;
            s = 0.0f;
#pragma unroll
            for (int d = 0; d < DREG; ++d)
                if (d < D) {
                    e[d] = expf(e[d] - m);
                    s += e[d];
                }
        } else {
            m = c[0];
#pragma unroll 4
            for (int d = 1; d < D; ++d) {
                const float v = c[(size_t)d * HW];
                m = v > m ? v : m;
            }
            s = 0.0f;
#pragma unroll 4
            for (int d = 0; d < D; ++d) s += expf(c[(size_t)d * HW] - m);
        }
    }

    // sorted (descending) candidate list in registers; strict '>' keeps the lower plane on ties
    float bv[KT], bo[KT];
    int bi[KT];
#pragma unroll
    for (int k = 0; k < KT; ++k) {
        bv[k] = -1.0f;
        bo[k] = 0.0f;
        bi[k] = 0;
    }
    float avg = 0.0f;
    auto plane = [&](int d, float ed) {
        float pd, od;
        if (kFromLogits) {
            pd = (DREG > 0 ? ed : expf(c[(size_t)d * HW] - m)) / s;
            od = 1.0f / (1.0f + expf(-o[(size_t)d * HW]));
            prob[base + (size_t)d * HW] = pd;
            off[base + (size_t)d * HW] = od;
        } else {  // inputs already are prob / off (a6+a7 only)
            pd = c[(size_t)d * HW];
            od = o[(size_t)d * HW];
        }
        // mvsdet.py:278-282  depth = idx*interval + near + off*interval (each op rounded)
        const float dep = ((float)d * interval + near) + od * interval;
        avg = avg + dep * pd;
        float cv = pd, co = od;
        int cidx = d;
#pragma unroll
        for (int k = 0; k < KT; ++k) {
            const bool gt = cv > bv[k];
            const float tv = bv[k], to = bo[k];
            const int ti = bi[k];
            bv[k] = gt ? cv : tv;
            bo[k] = gt ? co : to;
            bi[k] = gt ? cidx : ti;
            cv = gt ? tv : cv;
            co = gt ? to : co;
            cidx = gt ? ti : cidx;
        }
    };
    if (kFromLogits && DREG > 0) {
#pragma unroll
        for (int d = 0; d < DREG; ++d)
            if (d < D) plane(d, e[d]);
    } else {
#pragma unroll 4
        for (int d = 0; d < D; ++d) plane(d, 0.0f);
    }
    avg_depth[(size_t)n * HW + pix] = avg;
#pragma unroll
    for (int k = 0; k < KT; ++k) {
        if (k < topk) {
            const size_t oi = ((size_t)n * topk + k) * HW + pix;
            est_depth[oi] = ((float)bi[k] * interval + near) + bo[k] * interval;
            est_dens[oi] = bv[k];
            if (est_idx) est_idx[oi] = bi[k];
        }
    }
}

// backward: dL/dcost = p * (gp - sum_d gp_d p_d), dL/doff_logit = go * off * (1 - off), where gp / go
// collect the gradients that reach prob / off through prob itself, the top-k densities and depths and
// the depth expectation.
__global__ __launch_bounds__(kThreads) void depth_prob_topk_bwd_kernel(
    const float* __restrict__ prob, const float* __restrict__ off, const int32_t* __restrict__ est_idx,
    const float* __restrict__ g_prob, const float* __restrict__ g_depth, const float* __restrict__ g_dens,
    const float* __restrict__ g_avg, float* __restrict__ g_cost, float* __restrict__ g_offlogit, int D, int HW, int topk,
    float near, float interval) {
    const int pix = blockIdx.x * kThreads + threadIdx.x;
    const int n = blockIdx.y;
    if (pix >= HW) return;
    const size_t base = (size_t)n * D * HW + pix;
    const float ga = g_avg ? g_avg[(size_t)n * HW + pix] : 0.0f;
    int ki[MVSDET_MAX_TOPK];
    float kd[MVSDET_MAX_TOPK], kz[MVSDET_MAX_TOPK];
#pragma unroll
    for (int k = 0; k < MVSDET_MAX_TOPK; ++k) {
        ki[k] = -1;
        kd[k] = 0.0f;
        kz[k] = 0.0f;
        if (k < topk) {
            const size_t oi = ((size_t)n * topk + k) * HW + pix;
            ki[k] = est_idx[oi];
            kd[k] = g_dens ? g_dens[oi] : 0.0f;
            kz[k] = g_depth ? g_depth[oi] : 0.0f;
        }
    }
    // pass 1: dot = sum_d gp_d * p_d
    float dot = 0.0f;
    for (int d = 0; d < D; ++d) {
        const float p = prob[base + (size_t)d * HW];
        const float od = off[base + (size_t)d * HW];
        float gp = g_prob ? g_prob[base + (size_t)d * HW] : 0.0f;
        gp += ga * (((float)d * interval + near) + od * interval);
#pragma unroll
        for (int k = 0; k < MVSDET_MAX_TOPK; ++k) gp += (ki[k] == d) ? kd[k] : 0.0f;
        dot += gp * p;
    }
    for (int d = 0; d < D; ++d) {
        const float p = prob[base + (size_t)d * HW];
        const float od = off[base + (size_t)d * HW];
        float gp = g_prob ? g_prob[base + (size_t)d * HW] : 0.0f;
        gp += ga * (((float)d * interval + near) + od * interval);
        float go = ga * p * interval;
#pragma unroll
        for (int k = 0; k < MVSDET_MAX_TOPK; ++k) {
            gp += (ki[k] == d) ? kd[k] : 0.0f;
            go += (ki[k] == d) ? kz[k] * interval : 0.0f;
        }
        g_cost[base + (size_t)d * HW] = p * (gp - dot);
        g_offlogit[base + (size_t)d * HW] = go * od * (1.0f - od);
    }
}

}  // namespace mvsdet

using namespace mvsdet;

static int check_stage2(const char* name, int N, int D, int H, int W, int topk) {
    MVS_REQUIRE(N > 0 && D > 0 && H > 0 && W > 0, "%s: bad shape N=%d D=%d H=%d W=%d", name, N, D, H, W);
    MVS_REQUIRE(N <= 65535, "%s: N > 65535", name);
    MVS_REQUIRE(D <= MVSDET_MAX_DEPTH, "%s: D=%d > %d", name, D, MVSDET_MAX_DEPTH);
    MVS_REQUIRE(topk >= 1 && topk <= MVSDET_MAX_TOPK && topk <= D, "%s: topk=%d outside [1,min(%d,D)]", name, topk,
                MVSDET_MAX_TOPK);
    MVS_REQUIRE((size_t)H * W < (size_t)INT32_MAX, "%s: H*W too large", name);
    return MVSDET_OK;
}

extern "C" int mvsdet_depth_prob_topk_strided_f32(const float* cost_reg, const float* off_logit, long long view_stride,
                                                  float* prob, float* off, float* est_depth, float* est_dens,
                                                  int32_t* est_idx, float* avg_depth, int N, int D, int H, int W, int topk,
                                                  float near, float interval, mvsdet_stream_t stream) {
    MVS_REQUIRE(cost_reg && off_logit && prob && off && est_depth && est_dens && avg_depth, "depth_prob_topk: NULL pointer");
    if (int rc = check_stage2("depth_prob_topk", N, D, H, W, topk)) return rc;
    MVS_REQUIRE(view_stride >= (long long)D * H * W, "depth_prob_topk: view stride %lld < D*H*W", view_stride);
    const int HW = H * W;
    dim3 grid((HW + kThreads - 1) / kThreads, N);
#define MVS_DP_LAUNCH(DR, KTV)                                                                                             \
    hipLaunchKernelGGL((depth_prob_topk_kernel<true, DR, KTV>), grid, dim3(kThreads), 0, (hipStream_t)stream, cost_reg, off_logit, \
                       prob, off, est_depth, est_dens, est_idx, avg_depth, D, HW, topk, near, interval, (size_t)view_stride)
#define MVS_DP_BY_D(KTV)                  \
    if (D <= 16) MVS_DP_LAUNCH(16, KTV);  \
    else if (D <= 64) MVS_DP_LAUNCH(64, KTV); \
    else MVS_DP_LAUNCH(0, KTV);
    if (topk <= 3) { MVS_DP_BY_D(3) } else { MVS_DP_BY_D(MVSDET_MAX_TOPK) }
#undef MVS_DP_BY_D
#undef MVS_DP_LAUNCH
    MVS_LAUNCH_CHECK("depth_prob_topk");
    return MVSDET_OK;
}

extern "C" int mvsdet_depth_prob_topk_f32(const float* cost_reg, const float* off_logit, float* prob, float* off,
                                          float* est_depth, float* est_dens, int32_t* est_idx, float* avg_depth, int N,
                                          int D, int H, int W, int topk, float near, float interval,
                                          mvsdet_stream_t stream) {
    return mvsdet_depth_prob_topk_strided_f32(cost_reg, off_logit, (long long)D * H * W, prob, off, est_depth, est_dens, est_idx,
                                              avg_depth, N, D, H, W, topk, near, interval, stream);
}

extern "C" int mvsdet_sample_depth_prob_f32(const float* prob, const float* off, float* est_depth, float* est_dens,
                                            int32_t* est_idx, float* avg_depth, int N, int D, int H, int W, int topk,
                                            float near, float interval, mvsdet_stream_t stream) {
    MVS_REQUIRE(prob && off && est_depth && est_dens && avg_depth, "sample_depth_prob: NULL pointer");
    if (int rc = check_stage2("sample_depth_prob", N, D, H, W, topk)) return rc;
    const int HW = H * W;
    dim3 grid((HW + kThreads - 1) / kThreads, N);
    if (topk <= 3)
        hipLaunchKernelGGL((depth_prob_topk_kernel<false, 0, 3>), grid, dim3(kThreads), 0, (hipStream_t)stream, prob, off,
                           (float*)nullptr, (float*)nullptr, est_depth, est_dens, est_idx, avg_depth, D, HW, topk, near, interval,
                           (size_t)D * HW);
    else
    hipLaunchKernelGGL((depth_prob_topk_kernel<false, 0, MVSDET_MAX_TOPK>), grid, dim3(kThreads), 0, (hipStream_t)stream, prob, off,
                       (float*)nullptr, (float*)nullptr, est_depth, est_dens, est_idx, avg_depth, D, HW, topk, near, interval,
                       (size_t)D * HW);
    MVS_LAUNCH_CHECK("sample_depth_prob");
    return MVSDET_OK;
}

// ---------------------------------------------------------------------------------------------
// NVS-branch input: mvsdet.py:1158-1216 compute_depth_scale[_MultiIntrin] + :494.
//   ray = lift(x, y, 1) with the feature-level intrinsics (mvsdet.py:1300-1313), depth_scale = z of the normalised ray
//   = 1 / |ray| (get_camera_params with the identity pose, :1272-1297), est_ray_depth = est_depth / (depth_scale + 1e-8).
// One thread per pixel of the un-padded (h, w) feature map; the candidates come from the padded (H, W) maps.
// ---------------------------------------------------------------------------------------------
namespace mvsdet {
__global__ __launch_bounds__(kThreads) void ray_depth_kernel(const float* __restrict__ intr, const float* __restrict__ est_depth,
                                                             float* __restrict__ scale, float* __restrict__ ray_depth, int J,
                                                             int H, int W, int h, int w) {
    const int pix = blockIdx.x * kThreads + threadIdx.x;
    const int n = blockIdx.y;
    if (pix >= h * w) return;
    const int y = pix / w, x = pix - y * w;
    const float fx = intr[n * 5 + 0], fy = intr[n * 5 + 1], cx = intr[n * 5 + 2], cy = intr[n * 5 + 3], sk = intr[n * 5 + 4];
    // mvsdet.py:1308-1309, same operator order (z = 1)
    const float xl = ((float)x - cx + cy * sk / fy - sk * (float)y / fy) / fx * 1.0f;
    const float yl = ((float)y - cy) / fy * 1.0f;
    const float nrm = fmaxf(sqrtf(xl * xl + yl * yl + 1.0f), 1e-12f);   // F.normalize(dim=2): v / max(|v|, eps)
    const float sc = 1.0f / nrm;
    scale[(size_t)n * h * w + pix] = sc;
    if (ray_depth)
        for (int j = 0; j < J; ++j)
            ray_depth[((size_t)n * J + j) * h * w + pix] = est_depth[((size_t)n * J + j) * H * W + (size_t)y * W + x] / (sc + 1e-8f);
}
}  // namespace mvsdet

extern "C" int mvsdet_ray_depth_f32(const float* intr, const float* est_depth, float* depth_scale, float* est_ray_depth, int N,
                                    int J, int H, int W, int h, int w, mvsdet_stream_t stream) {
    MVS_REQUIRE(intr && depth_scale, "ray_depth: NULL pointer");
    MVS_REQUIRE((est_depth == nullptr) == (est_ray_depth == nullptr), "ray_depth: est_depth and est_ray_depth come together");
    MVS_REQUIRE(N > 0 && N <= 65535 && J >= 0 && h > 0 && w > 0 && h <= H && w <= W, "ray_depth: bad shape N=%d J=%d H=%d W=%d h=%d w=%d",
                N, J, H, W, h, w);
    dim3 grid((h * w + kThreads - 1) / kThreads, N);
    hipLaunchKernelGGL(ray_depth_kernel, grid, dim3(kThreads), 0, (hipStream_t)stream, intr, est_depth, depth_scale, est_ray_depth,
                       J, H, W, h, w);
    MVS_LAUNCH_CHECK("ray_depth");
    return MVSDET_OK;
}

extern "C" int mvsdet_depth_prob_topk_bwd_f32(const float* prob, const float* off, const int32_t* est_idx,
                                              const float* g_prob, const float* g_depth, const float* g_dens,
                                              const float* g_avg, float* g_cost, float* g_offlogit, int N, int D, int H,
                                              int W, int topk, float near, float interval, mvsdet_stream_t stream) {
    MVS_REQUIRE(prob && off && est_idx && g_cost && g_offlogit, "depth_prob_topk_bwd: NULL pointer");
    if (int rc = check_stage2("depth_prob_topk_bwd", N, D, H, W, topk)) return rc;
    const int HW = H * W;
    dim3 grid((HW + kThreads - 1) / kThreads, N);
    hipLaunchKernelGGL(depth_prob_topk_bwd_kernel, grid, dim3(kThreads), 0, (hipStream_t)stream, prob, off, est_idx, g_prob,
                       g_depth, g_dens, g_avg, g_cost, g_offlogit, D, HW, topk, near, interval);
    MVS_LAUNCH_CHECK("depth_prob_topk_bwd");
    return MVSDET_OK;
}
