// Training-mode BatchNorm3d (+ ReLU) of the cost regularisation network (mvs_models/module.py:26-37 ConvBnReLU3D,
// mvsnet.py:92-100 the Sequential(ConvTranspose3d, BatchNorm3d, ReLU) blocks): batch statistics over (N, D, H, W) per
// channel, running statistics updated in place, affine and ReLU in the same pass; and its backward.
//
//   forward : mean_c, var_c (biased) over M = N*D*H*W;  y = relu(gamma_c * (x - mean_c) * invstd_c + beta_c)
//             running_mean = (1-m) running_mean + m mean;  running_var = (1-m) running_var + m var * M/(M-1)
//   backward: g' = g * [y > 0];  dbeta = sum g';  dgamma = sum g' * xhat;
//             dx = gamma * invstd * (g' - dbeta/M - xhat * dgamma/M)
//
// Memory-bound streaming kernels: x is read twice going forward (statistics, then apply) and x and g twice going backward
// (reductions, then apply); the ReLU mask is recomputed from x with the forward's own arithmetic instead of being stored.
// The sums are taken per thread in fp32 over a short strip, then in double across the block and the splits (a channel of
// the full-resolution layers has 2.3 M elements).  A channel's elements are N contiguous runs of `vol` floats.
#include "common.h"

#include <algorithm>

namespace mvsdet {

constexpr int kBnSplit = 32;   // blocks per channel in the reduction kernels

__device__ __forceinline__ double2 block_sum2(double a, double b) {
    __shared__ double s_a[kThreads / 64], s_b[kThreads / 64];
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) {
        a += __shfl_down(a, o, 64);
        b += __shfl_down(b, o, 64);
    }
    const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
    __syncthreads();   // a previous use of the buffers
    if (lane == 0) { s_a[wave] = a; s_b[wave] = b; }
    __syncthreads();
    double ra = 0.0, rb = 0.0;
#pragma unroll
    for (int w = 0; w < kThreads / 64; ++w) { ra += s_a[w]; rb += s_b[w]; }
    return make_double2(ra, rb);
}

// the strip of channel c that block (c, split) reduces: elements [begin, end) of the channel's N*vol, in units of 4 floats
// when vol % 4 == 0 (VEC) or single floats
template <bool VEC, typename F>
__device__ __forceinline__ void for_channel_strip(int N, int C, size_t vol, int c, int split, int nsplit, F&& body) {
    const size_t unit = VEC ? 4 : 1;
    const size_t per_view = vol / unit, total = (size_t)N * per_view;
    const size_t chunk = (total + nsplit - 1) / nsplit;
    const size_t begin = (size_t)split * chunk, end = begin + chunk < total ? begin + chunk : total;
    for (size_t e = begin + threadIdx.x; e < end; e += kThreads) {
        const size_t n = e / per_view, i = e - n * per_view;
        body((n * C + c) * vol + i * unit);
    }
}

template <bool VEC>
__global__ __launch_bounds__(kThreads) void bn_stats_kernel(const float* __restrict__ x, double2* __restrict__ partial, int N, int C,
                                                           size_t vol) {
    const int c = blockIdx.x, split = blockIdx.y;
    // sums of (x - pivot) and (x - pivot)^2 with the channel's first element as the pivot: E[x^2] - mean^2 on the raw
    // values cancels catastrophically in fp32 strips when |mean| >> std (a channel behind a large bias: error
    // ~1e-7 * mean^2 / var); shifted, the strips hold values of the size of the spread
    const float pv = x[(size_t)c * vol];
    float s = 0.0f, q = 0.0f;
    for_channel_strip<VEC>(N, C, vol, c, split, gridDim.y, [&](size_t off) {
        if (VEC) {
            float4 v = *reinterpret_cast<const float4*>(x + off);
            v.x -= pv; v.y -= pv; v.z -= pv; v.w -= pv;
            s += (v.x + v.y) + (v.z + v.w);
            q += (v.x * v.x + v.y * v.y) + (v.z * v.z + v.w * v.w);
        } else {
            const float v = x[off] - pv;
            s += v;
            q += v * v;
        }
    });
    const double2 r = block_sum2((double)s, (double)q);
    if (threadIdx.x == 0) partial[(size_t)c * gridDim.y + split] = r;
}

// one thread per channel: statistics from the partial sums, running statistics, the affine of the apply pass
__global__ void bn_finalize_kernel(const float* __restrict__ x, size_t vol, const double2* __restrict__ partial, int nsplit, const float* __restrict__ gamma,
                                   const float* __restrict__ beta, float* __restrict__ running_mean, float* __restrict__ running_var,
                                   float* __restrict__ save_mean, float* __restrict__ save_invstd, float* __restrict__ scale,
                                   float* __restrict__ shift, int C, double M, float momentum, float eps) {
    const int c = blockIdx.x * blockDim.x + threadIdx.x;
    if (c >= C) return;
    double s = 0.0, q = 0.0;
    for (int i = 0; i < nsplit; ++i) { s += partial[(size_t)c * nsplit + i].x; q += partial[(size_t)c * nsplit + i].y; }
    const double dm = s / M;                      // mean of (x - pivot), pivot = the channel's first element (bn_stats_kernel)
    const double mean = (double)x[(size_t)c * vol] + dm;
    double var = q / M - dm * dm;
    var = var < 0.0 ? 0.0 : var;
    const float invstd = (float)(1.0 / sqrt(var + (double)eps));
    save_mean[c] = (float)mean;
    save_invstd[c] = invstd;
    const float g = gamma ? gamma[c] : 1.0f, b = beta ? beta[c] : 0.0f;
    scale[c] = g * invstd;
    shift[c] = b - (float)mean * (g * invstd);
    if (running_mean) running_mean[c] = (1.0f - momentum) * running_mean[c] + momentum * (float)mean;
    if (running_var) running_var[c] = (1.0f - momentum) * running_var[c] + momentum * (float)(M > 1.0 ? var * M / (M - 1.0) : var);
}

// The same, from the partial sums a convolution's epilogue left (costreg_bf16.hip: sums of the raw values and of their squares, one
// double2 per (channel, block), of (value - pivot_c), added up in double from the block's fp32 lane sums).  One block per
// channel: the `parts` entries are added in a fixed order (strided over the threads, then block_sum2): the same bits on every run.
__global__ __launch_bounds__(kThreads) void bn_finalize_parts_kernel(const double2* __restrict__ partial, size_t parts, const float* __restrict__ pivot,
                                                                     const float* __restrict__ gamma,
                                                                     const float* __restrict__ beta, float* __restrict__ running_mean,
                                                                     float* __restrict__ running_var, float* __restrict__ save_mean,
                                                                     float* __restrict__ save_invstd, float* __restrict__ scale,
                                                                     float* __restrict__ shift, double M, float momentum, float eps) {
    const int c = blockIdx.x;
    double s = 0.0, q = 0.0;
    for (size_t i = threadIdx.x; i < parts; i += kThreads) {
        const double2 e = partial[(size_t)c * parts + i];
        s += e.x;
        q += e.y;
    }
    const double2 r = block_sum2(s, q);
    if (threadIdx.x != 0) return;
    const double dm = r.x / M;                    // mean of (x - pivot)
    const double mean = (pivot ? (double)pivot[c] : 0.0) + dm;
    double var = r.y / M - dm * dm;
    var = var < 0.0 ? 0.0 : var;
    const float invstd = (float)(1.0 / sqrt(var + (double)eps));
    save_mean[c] = (float)mean;
    save_invstd[c] = invstd;
    const float g = gamma ? gamma[c] : 1.0f, b = beta ? beta[c] : 0.0f;
    scale[c] = g * invstd;
    shift[c] = b - (float)mean * (g * invstd);
    if (running_mean) running_mean[c] = (1.0f - momentum) * running_mean[c] + momentum * (float)mean;
    if (running_var) running_var[c] = (1.0f - momentum) * running_var[c] + momentum * (float)(M > 1.0 ? var * M / (M - 1.0) : var);
}

// y = [relu](x * scale_c + shift_c) [+ residual]; grid (chunks of a (view, channel) volume, C, N)
template <bool VEC>
__global__ __launch_bounds__(kThreads) void bn_apply_kernel(const float* __restrict__ x, const float* __restrict__ scale,
                                                           const float* __restrict__ shift, const float* __restrict__ residual,
                                                           float* __restrict__ y, int C, size_t vol, int relu) {
    const int c = blockIdx.y, n = blockIdx.z;
    const float sc = scale[c], sh = shift[c];
    const size_t base = ((size_t)n * C + c) * vol;
    const size_t unit = VEC ? 4 : 1, cnt = vol / unit;
    for (size_t i = (size_t)blockIdx.x * kThreads + threadIdx.x; i < cnt; i += (size_t)gridDim.x * kThreads) {
        if (VEC) {
            float4 v = *reinterpret_cast<const float4*>(x + base + 4 * i);
            v.x = fmaf(v.x, sc, sh); v.y = fmaf(v.y, sc, sh); v.z = fmaf(v.z, sc, sh); v.w = fmaf(v.w, sc, sh);
            if (relu) { v.x = fmaxf(v.x, 0.f); v.y = fmaxf(v.y, 0.f); v.z = fmaxf(v.z, 0.f); v.w = fmaxf(v.w, 0.f); }
            if (residual) {   // the skip tensor of mvsnet.py:109-111, added after the activation as `skip + relu(bn(.))` does
                const float4 r = *reinterpret_cast<const float4*>(residual + base + 4 * i);
                v.x += r.x; v.y += r.y; v.z += r.z; v.w += r.w;
            }
            *reinterpret_cast<float4*>(y + base + 4 * i) = v;
        } else {
            float v = fmaf(x[base + i], sc, sh);
            v = relu ? fmaxf(v, 0.f) : v;
            y[base + i] = residual ? v + residual[base + i] : v;
        }
    }
}

// backward reductions: partial[c][split] = (sum g', sum g' * xhat)
template <bool VEC>
__global__ __launch_bounds__(kThreads) void bn_bwd_reduce_kernel(const float* __restrict__ x, const float* __restrict__ gy,
                                                                const float* __restrict__ mean, const float* __restrict__ invstd,
                                                                const float* __restrict__ scale, const float* __restrict__ shift,
                                                                double2* __restrict__ partial, int N, int C, size_t vol, int relu) {
    const int c = blockIdx.x, split = blockIdx.y;
    const float mu = mean[c], is = invstd[c], sc = scale[c], sh = shift[c];
    float s = 0.0f, q = 0.0f;
    auto one = [&](float xv, float g) {
        if (relu && !(fmaf(xv, sc, sh) > 0.0f)) g = 0.0f;   // the forward's own arithmetic: the same sign
        s += g;
        q += g * ((xv - mu) * is);
    };
    for_channel_strip<VEC>(N, C, vol, c, split, gridDim.y, [&](size_t off) {
        if (VEC) {
            const float4 xv = *reinterpret_cast<const float4*>(x + off), g = *reinterpret_cast<const float4*>(gy + off);
            one(xv.x, g.x); one(xv.y, g.y); one(xv.z, g.z); one(xv.w, g.w);
        } else {
            one(x[off], gy[off]);
        }
    });
    const double2 r = block_sum2((double)s, (double)q);
    if (threadIdx.x == 0) partial[(size_t)c * gridDim.y + split] = r;
}

__global__ void bn_affine_kernel(const float* __restrict__ gamma, const float* __restrict__ beta, const float* __restrict__ mean,
                                 const float* __restrict__ invstd, float* __restrict__ scale, float* __restrict__ shift, int C) {
    const int c = blockIdx.x * blockDim.x + threadIdx.x;
    if (c >= C) return;
    const float g = gamma ? gamma[c] : 1.0f, b = beta ? beta[c] : 0.0f;
    scale[c] = g * invstd[c];
    shift[c] = b - mean[c] * (g * invstd[c]);   // the expression of bn_finalize_kernel: the same bits
}

__global__ void bn_bwd_finalize_kernel(const double2* __restrict__ partial, int nsplit, float* __restrict__ ggamma,
                                       float* __restrict__ gbeta, float* __restrict__ sums, int C) {
    const int c = blockIdx.x * blockDim.x + threadIdx.x;
    if (c >= C) return;
    double s = 0.0, q = 0.0;
    for (int i = 0; i < nsplit; ++i) { s += partial[(size_t)c * nsplit + i].x; q += partial[(size_t)c * nsplit + i].y; }
    if (gbeta) gbeta[c] = (float)s;
    if (ggamma) ggamma[c] = (float)q;
    sums[2 * c] = (float)s;
    sums[2 * c + 1] = (float)q;
}

// dx = gamma * invstd * (g' - dbeta/M - xhat * dgamma/M)
template <bool VEC>
__global__ __launch_bounds__(kThreads) void bn_bwd_apply_kernel(const float* __restrict__ x, const float* __restrict__ gy,
                                                               const float* __restrict__ mean, const float* __restrict__ invstd,
                                                               const float* __restrict__ scale, const float* __restrict__ shift,
                                                               const float* __restrict__ sums, float* __restrict__ gx, int C,
                                                               size_t vol, float inv_m, int relu) {
    const int c = blockIdx.y, n = blockIdx.z;
    const float mu = mean[c], is = invstd[c], sc = scale[c], sh = shift[c];
    const float db = sums[2 * c] * inv_m, dg = sums[2 * c + 1] * inv_m;
    const size_t base = ((size_t)n * C + c) * vol;
    const size_t unit = VEC ? 4 : 1, cnt = vol / unit;
    auto one = [&](float xv, float g) {
        if (relu && !(fmaf(xv, sc, sh) > 0.0f)) g = 0.0f;
        return sc * (g - db - ((xv - mu) * is) * dg);    // sc = gamma * invstd
    };
    for (size_t i = (size_t)blockIdx.x * kThreads + threadIdx.x; i < cnt; i += (size_t)gridDim.x * kThreads) {
        if (VEC) {
            const float4 xv = *reinterpret_cast<const float4*>(x + base + 4 * i), g = *reinterpret_cast<const float4*>(gy + base + 4 * i);
            *reinterpret_cast<float4*>(gx + base + 4 * i) = make_float4(one(xv.x, g.x), one(xv.y, g.y), one(xv.z, g.z), one(xv.w, g.w));
        } else {
            gx[base + i] = one(x[base + i], gy[base + i]);
        }
    }
}

}  // namespace mvsdet

using namespace mvsdet;

// workspace: kBnSplit double2 per channel + scale, shift (C floats each) + 2 C floats of backward sums
extern "C" size_t mvsdet_bn3d_workspace_bytes(int C) {
    return C <= 0 ? 0 : (size_t)C * kBnSplit * sizeof(double2) + (size_t)4 * C * sizeof(float);
}

namespace {
struct BnWs {
    double2* partial;
    float *scale, *shift, *sums;
};
BnWs bn_ws(void* workspace, int C) {
    BnWs w;
    w.partial = static_cast<double2*>(workspace);
    w.scale = reinterpret_cast<float*>(w.partial + (size_t)C * kBnSplit);
    w.shift = w.scale + C;
    w.sums = w.shift + C;
    return w;
}
int bn_check(const char* name, int N, int C, long long vol, const void* workspace, size_t workspace_bytes) {
    MVS_REQUIRE(N > 0 && C > 0 && vol > 0, "%s: bad shape N=%d C=%d volume=%lld", name, N, C, vol);
    MVS_REQUIRE(C <= 65535 && N <= 65535, "%s: N or C > 65535", name);
    MVS_REQUIRE(workspace && ((uintptr_t)workspace & 15u) == 0, "%s: workspace NULL or not 16-byte aligned", name);
    if (workspace_bytes < mvsdet_bn3d_workspace_bytes(C)) {
        set_error("%s: workspace %zu B < %zu B", name, workspace_bytes, mvsdet_bn3d_workspace_bytes(C));
        return MVSDET_ERR_WORKSPACE;
    }
    return MVSDET_OK;
}
}  // namespace

extern "C" int mvsdet_bn3d_relu_train_fwd_res_f32(const float* x, const float* gamma, const float* beta, const float* residual,
                                                  float* running_mean, float* running_var, float* out, float* save_mean,
                                                  float* save_invstd, void* workspace, size_t workspace_bytes, int N, int C,
                                                  long long vol, float momentum, float eps, int relu, mvsdet_stream_t stream_) {
    hipStream_t stream = (hipStream_t)stream_;
    MVS_REQUIRE(x && out && save_mean && save_invstd, "bn3d_relu_train_fwd: NULL pointer");
    if (int rc = bn_check("bn3d_relu_train_fwd", N, C, vol, workspace, workspace_bytes)) return rc;
    const BnWs w = bn_ws(workspace, C);
    const bool vec = (vol % 4 == 0) && (((uintptr_t)x | (uintptr_t)out | (uintptr_t)residual) & 15u) == 0;
    dim3 rgrid((unsigned)C, kBnSplit);
    if (vec) hipLaunchKernelGGL(bn_stats_kernel<true>, rgrid, dim3(kThreads), 0, stream, x, w.partial, N, C, (size_t)vol);
    else hipLaunchKernelGGL(bn_stats_kernel<false>, rgrid, dim3(kThreads), 0, stream, x, w.partial, N, C, (size_t)vol);
    hipLaunchKernelGGL(bn_finalize_kernel, dim3((C + 63) / 64), dim3(64), 0, stream, x, (size_t)vol, w.partial, kBnSplit, gamma, beta, running_mean,
                       running_var, save_mean, save_invstd, w.scale, w.shift, C, (double)N * (double)vol, momentum, eps);
    const size_t cnt = (size_t)vol / (vec ? 4 : 1);
    dim3 agrid((unsigned)std::min<size_t>((cnt + kThreads - 1) / kThreads, 64), (unsigned)C, (unsigned)N);
    if (vec) hipLaunchKernelGGL(bn_apply_kernel<true>, agrid, dim3(kThreads), 0, stream, x, w.scale, w.shift, residual, out, C, (size_t)vol, relu);
    else hipLaunchKernelGGL(bn_apply_kernel<false>, agrid, dim3(kThreads), 0, stream, x, w.scale, w.shift, residual, out, C, (size_t)vol, relu);
    MVS_LAUNCH_CHECK("bn3d_relu_train_fwd");
    return MVSDET_OK;
}

// Training-mode BatchNorm3d [+ ReLU] [+ residual] whose statistics arrive as partial sums from the producing convolution
// (mvsdet_conv3d_k3_bf16x3_stats: partial[c * parts + i] = (sum, sum of squares) of (x - pivot_c) over channel c in block i; pivot as
// given to that call, NULL = zeros): x is read ONCE.
extern "C" int mvsdet_bn3d_relu_train_fwd_parts_f32(const float* x, const void* partial, size_t parts, const float* pivot, const float* gamma,
                                                    const float* beta, const float* residual, float* running_mean, float* running_var,
                                                    float* out, float* save_mean, float* save_invstd, void* workspace,
                                                    size_t workspace_bytes, int N, int C, long long vol, float momentum, float eps,
                                                    int relu, mvsdet_stream_t stream_) {
    hipStream_t stream = (hipStream_t)stream_;
    MVS_REQUIRE(x && out && save_mean && save_invstd && partial && parts > 0, "bn3d_relu_train_fwd_parts: NULL pointer or no partial sums");
    MVS_REQUIRE(((uintptr_t)partial & 15u) == 0, "bn3d_relu_train_fwd_parts: partial sums must be 16-byte aligned");
    if (int rc = bn_check("bn3d_relu_train_fwd_parts", N, C, vol, workspace, workspace_bytes)) return rc;
    const BnWs w = bn_ws(workspace, C);
    const bool vec = (vol % 4 == 0) && (((uintptr_t)x | (uintptr_t)out | (uintptr_t)residual) & 15u) == 0;
    hipLaunchKernelGGL(bn_finalize_parts_kernel, dim3((unsigned)C), dim3(kThreads), 0, stream, static_cast<const double2*>(partial), parts, pivot, gamma,
                       beta, running_mean, running_var, save_mean, save_invstd, w.scale, w.shift, (double)N * (double)vol, momentum, eps);
    const size_t cnt = (size_t)vol / (vec ? 4 : 1);
    dim3 agrid((unsigned)std::min<size_t>((cnt + kThreads - 1) / kThreads, 64), (unsigned)C, (unsigned)N);
    if (vec) hipLaunchKernelGGL(bn_apply_kernel<true>, agrid, dim3(kThreads), 0, stream, x, w.scale, w.shift, residual, out, C, (size_t)vol, relu);
    else hipLaunchKernelGGL(bn_apply_kernel<false>, agrid, dim3(kThreads), 0, stream, x, w.scale, w.shift, residual, out, C, (size_t)vol, relu);
    MVS_LAUNCH_CHECK("bn3d_relu_train_fwd_parts");
    return MVSDET_OK;
}

extern "C" int mvsdet_bn3d_relu_train_fwd_f32(const float* x, const float* gamma, const float* beta, float* running_mean,
                                              float* running_var, float* out, float* save_mean, float* save_invstd,
                                              void* workspace, size_t workspace_bytes, int N, int C, long long vol, float momentum,
                                              float eps, int relu, mvsdet_stream_t stream) {
    return mvsdet_bn3d_relu_train_fwd_res_f32(x, gamma, beta, nullptr, running_mean, running_var, out, save_mean, save_invstd, workspace,
                                              workspace_bytes, N, C, vol, momentum, eps, relu, stream);
}

extern "C" int mvsdet_bn3d_relu_bwd_f32(const float* x, const float* grad_out, const float* gamma, const float* beta,
                                        const float* save_mean, const float* save_invstd, float* grad_x, float* grad_gamma,
                                        float* grad_beta, void* workspace, size_t workspace_bytes, int N, int C, long long vol,
                                        int relu, mvsdet_stream_t stream_) {
    hipStream_t stream = (hipStream_t)stream_;
    MVS_REQUIRE(x && grad_out && save_mean && save_invstd && grad_x, "bn3d_relu_bwd: NULL pointer");
    if (int rc = bn_check("bn3d_relu_bwd", N, C, vol, workspace, workspace_bytes)) return rc;
    const BnWs w = bn_ws(workspace, C);
    const bool vec = (vol % 4 == 0) && (((uintptr_t)x | (uintptr_t)grad_out | (uintptr_t)grad_x) & 15u) == 0;
    // the forward's affine again, from the saved statistics (the forward call's workspace need not have survived)
    hipLaunchKernelGGL(bn_affine_kernel, dim3((C + 63) / 64), dim3(64), 0, stream, gamma, beta, save_mean, save_invstd, w.scale,
                       w.shift, C);
    dim3 rgrid((unsigned)C, kBnSplit);
    if (vec)
        hipLaunchKernelGGL(bn_bwd_reduce_kernel<true>, rgrid, dim3(kThreads), 0, stream, x, grad_out, save_mean, save_invstd, w.scale,
                           w.shift, w.partial, N, C, (size_t)vol, relu);
    else
        hipLaunchKernelGGL(bn_bwd_reduce_kernel<false>, rgrid, dim3(kThreads), 0, stream, x, grad_out, save_mean, save_invstd, w.scale,
                           w.shift, w.partial, N, C, (size_t)vol, relu);
    hipLaunchKernelGGL(bn_bwd_finalize_kernel, dim3((C + 63) / 64), dim3(64), 0, stream, w.partial, kBnSplit, grad_gamma, grad_beta,
                       w.sums, C);
    const size_t cnt = (size_t)vol / (vec ? 4 : 1);
    dim3 agrid((unsigned)std::min<size_t>((cnt + kThreads - 1) / kThreads, 64), (unsigned)C, (unsigned)N);
    const float inv_m = (float)(1.0 / ((double)N * (double)vol));
    if (vec)
        hipLaunchKernelGGL(bn_bwd_apply_kernel<true>, agrid, dim3(kThreads), 0, stream, x, grad_out, save_mean, save_invstd, w.scale,
                           w.shift, w.sums, grad_x, C, (size_t)vol, inv_m, relu);
    else
        hipLaunchKernelGGL(bn_bwd_apply_kernel<false>, agrid, dim3(kThreads), 0, stream, x, grad_out, save_mean, save_invstd, w.scale,
                           w.shift, w.sums, grad_x, C, (size_t)vol, inv_m, relu);
    MVS_LAUNCH_CHECK("bn3d_relu_bwd");
    return MVSDET_OK;
}
