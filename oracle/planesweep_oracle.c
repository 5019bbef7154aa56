/*
 * oracle/planesweep_oracle.c -- TEST INFRASTRUCTURE, NOT PRODUCT CODE.
 *
 * Plain-C CPU restatement of the MVSDet probabilistic depth-sampling hot path
 * (SURVEY.md section 8a).  It is the checker the HIP kernels are compared with and
 * the "port" CPU baseline bench.py times beside them.  Only tests/, bench.py's
 * cpu_baseline leg and __graft_entry__.smoke() may load this library; the product
 * (mvsdet_amd/) never does.
 *
 * Parity pin: every function below is checked in tests/test_oracle_golden.py against
 * golden vectors produced by running the reference itself on PyTorch-CPU in the build
 * container (tests/golden/make_goldens.py).  The reference's own test-suite holds no
 * vectors for this path.
 *
 * Reference files are cited relative to projects/NeRF-Det/nerfdet/ of Pixie8888/MVSDet.
 * The arithmetic that the reference delegates to PyTorch (pinned pytorch=2.1.0,
 * environment.yaml:78) -- grid_sample, softmax, topk, bmm, round -- is restated from the
 * published ATen algorithms:
 *   grid_sample 2-D bilinear / zeros padding / align_corners=False
 *     (ATen/native/cpu/GridSamplerKernel.cpp: unnormalise (g+1)*(size/2)-0.5, taps
 *      floor/floor+1, weights from the fractional parts, out-of-range taps read as 0),
 *   softmax = exp(x-max)/sum, sigmoid = 1/(1+exp(-x)), topk = k largest in descending order,
 *   Tensor.round = round-half-to-even.
 *
 * The warp (orc_homo_warp) reproduces the reference's homo_warping BIT FOR BIT on the golden
 * vectors (same rounding points as torch.matmul / ATen-CPU grid_sample: see orc_compute_taps).
 * `mode` of orc_plane_sweep_variance selects how the variance expression is rounded:
 *   mode 0  eager-PyTorch rounding (mvsdet.py:458-467 run as separate tensor ops): every * and +
 *           rounded separately, true division by (K+1).
 *   mode 1  device rounding: fused multiply-adds and a multiply by the fp32 reciprocal of (K+1),
 *           which is what the HIP kernel does (and what ATen's GPU kernels do for tensor/scalar).
 * Both follow the same algorithm; they differ by <= a few ulp per element.  mode 1 lets the
 * GPU tests demand bit-for-bit equality with the HIP kernel.
 *
 * Build: see oracle/Makefile (gcc -O2 -ffp-contract=off -fopenmp).
 */
#include <math.h>
#include <stdint.h>
#include <stdlib.h>
#include <string.h>

#ifdef _OPENMP
#include <omp.h>
#endif

#define ORC_API __attribute__((visibility("default")))

ORC_API int orc_num_threads(void) {
#ifdef _OPENMP
    return omp_get_max_threads();
#else
    return 1;
#endif
}

ORC_API void orc_set_num_threads(int n) {
#ifdef _OPENMP
    if (n > 0) omp_set_num_threads(n);
#else
    (void)n;
#endif
}

/* ------------------------------------------------------------------------------------------
 * Sampling taps of one (reference pixel, depth plane, source view).
 * mvs_models/module.py:116-143 (homo_warping) + ATen grid_sample.
 * `P` is proj = src_proj @ inverse(ref_proj) (module.py:116), row-major 4x4;
 * rot = P[:3,:3], trans = P[:3,3] (module.py:117-118).
 * ---------------------------------------------------------------------------------------- */
typedef struct {
    int off[4];   /* element offset y*W+x of taps nw, ne, sw, se (0 when the tap is outside) */
    float w[4];   /* bilinear weights; an outside tap keeps weight*0 so NaN/Inf propagate as in ATen-CPU */
} orc_taps;

static inline void orc_compute_taps(const float* P, float x, float y, float d, int H, int W, orc_taps* t) {
    /* module.py:126  rot_xyz = rot @ [x, y, 1]^T, summed in k order with fused multiply-adds as a
     * GEMM inner loop does (bit-identical to torch.matmul on the golden-generating host) */
    float rx = fmaf(P[1], y, P[0] * x) + P[2];
    float ry = fmaf(P[5], y, P[4] * x) + P[6];
    float rz = fmaf(P[9], y, P[8] * x) + P[10];
    /* module.py:128,135  rot_depth_xyz = rot_xyz * d ; proj_xyz = rot_depth_xyz + trans */
    float X = rx * d;
    X = X + P[3];
    float Y = ry * d;
    Y = Y + P[7];
    float Z = rz * d;
    Z = Z + P[11];
    /* module.py:136  proj_xy = xy / z  (no z > 0 test: SURVEY D9) */
    float px = X / Z;
    float py = Y / Z;
    /* module.py:137-138  normalise with the align_corners=True formula ... */
    float gx = px / ((float)(W - 1) * 0.5f);
    gx = gx - 1.0f;
    float gy = py / ((float)(H - 1) * 0.5f);
    gy = gy - 1.0f;
    /* ... module.py:142 grid_sample(align_corners=False) un-normalises with (g+1)*size/2-0.5 */
    float ix = fmaf(gx + 1.0f, (float)W * 0.5f, -0.5f);
    float iy = fmaf(gy + 1.0f, (float)H * 0.5f, -0.5f);
    float x0 = floorf(ix), y0 = floorf(iy);
    float wx = ix - x0, wy = iy - y0; /* w, n of ATen */
    float ex = 1.0f - wx, sy = 1.0f - wy; /* e, s */
    float wnw = sy * ex, wne = sy * wx, wsw = wy * ex, wse = wy * wx;
    /* bounds in float so huge / non-finite coordinates never reach an int conversion */
    int x0in = (x0 >= 0.0f) && (x0 <= (float)(W - 1));
    int x1in = (x0 >= -1.0f) && (x0 <= (float)(W - 2));
    int y0in = (y0 >= 0.0f) && (y0 <= (float)(H - 1));
    int y1in = (y0 >= -1.0f) && (y0 <= (float)(H - 2));
    int xi = (x0in || x1in) ? (int)x0 : 0;
    int yi = (y0in || y1in) ? (int)y0 : 0;
    int in[4] = {x0in && y0in, x1in && y0in, x0in && y1in, x1in && y1in};
    int offs[4] = {yi * W + xi, yi * W + xi + 1, (yi + 1) * W + xi, (yi + 1) * W + xi + 1};
    float ws[4] = {wnw, wne, wsw, wse};
    for (int i = 0; i < 4; ++i) {
        t->off[i] = in[i] ? offs[i] : 0;
        t->w[i] = in[i] ? ws[i] : ws[i] * 0.0f;
    }
}

/* ATen-CPU's vectorised bilinear kernel evaluates nw_val*nw + ne_val*ne + sw_val*sw + se_val*se as a
 * chain of fused multiply-adds (observed: this chain is bit-identical to F.grid_sample on the golden
 * host, the separately rounded form is not). */
static inline float orc_sample(const float* plane, const orc_taps* t) {
    float s = plane[t->off[0]] * t->w[0];
    s = fmaf(plane[t->off[1]], t->w[1], s);
    s = fmaf(plane[t->off[2]], t->w[2], s);
    s = fmaf(plane[t->off[3]], t->w[3], s);
    return s;
}

/* a3: homo_warping, mvs_models/module.py:105-146.
 * src (B,C,H,W); proj (B,4,4) = src_proj @ inverse(ref_proj); depth (B,D); out (B,C,D,H,W). */
ORC_API void orc_homo_warp(const float* src, const float* proj, const float* depth, float* out,
                           int B, int C, int D, int H, int W) {
    const size_t HW = (size_t)H * W;
#pragma omp parallel
    {
        orc_taps* row = (orc_taps*)malloc(sizeof(orc_taps) * (size_t)W);
#pragma omp for collapse(2) schedule(static)
        for (int b = 0; b < B; ++b)
            for (int dy = 0; dy < D * H; ++dy) {
                int d = dy / H, y = dy % H;
                for (int x = 0; x < W; ++x)
                    orc_compute_taps(proj + (size_t)b * 16, (float)x, (float)y, depth[(size_t)b * D + d], H, W, &row[x]);
                for (int c = 0; c < C; ++c) {
                    const float* plane = src + ((size_t)b * C + c) * HW;
                    float* o = out + ((((size_t)b * C + c) * D + d) * H + y) * W;
                    for (int x = 0; x < W; ++x) o[x] = orc_sample(plane, &row[x]);
                }
            }
        free(row);
    }
}

/* a3+a4: plane-sweep variance cost volume, mvsdet.py:439-467.
 * feat (N,C,H,W); nbr (N,K) int64 = neighbour view of each reference view; proj (N,K,4,4) =
 * nei_proj_j @ inverse(ref_proj); depth (N,D); out (N,C,D,H,W).
 *   volume_sum    = ref + sum_j warped_j          (mvsdet.py:441,458)
 *   volume_sq_sum = ref^2 + sum_j warped_j^2      (mvsdet.py:442,459)
 *   variance      = volume_sq_sum/(K+1) - (volume_sum/(K+1))^2   (mvsdet.py:467)            */
#define ORC_MAX_K 8
ORC_API int orc_plane_sweep_variance(const float* feat, const int64_t* nbr, const float* proj, const float* depth,
                                     float* out, int N, int K, int C, int D, int H, int W, int mode) {
    if (K < 0 || K > ORC_MAX_K) return 1;
    for (int i = 0; i < N * K; ++i)
        if (nbr[i] < 0 || nbr[i] >= N) return 2;
    const size_t HW = (size_t)H * W;
    const float nviews = (float)(K + 1);
    const float rcp = 1.0f / nviews;
#pragma omp parallel
    {
        orc_taps* row = (orc_taps*)malloc(sizeof(orc_taps) * (size_t)W * (K > 0 ? K : 1));
#pragma omp for collapse(2) schedule(static)
        for (int n = 0; n < N; ++n)
            for (int dy = 0; dy < D * H; ++dy) {
                int d = dy / H, y = dy % H;
                for (int j = 0; j < K; ++j)
                    for (int x = 0; x < W; ++x)
                        orc_compute_taps(proj + ((size_t)n * K + j) * 16, (float)x, (float)y,
                                         depth[(size_t)n * D + d], H, W, &row[(size_t)j * W + x]);
                for (int c = 0; c < C; ++c) {
                    const float* refp = feat + (((size_t)n * C + c) * H + y) * W;
                    float* o = out + ((((size_t)n * C + c) * D + d) * H + y) * W;
                    for (int x = 0; x < W; ++x) {
                        float f = refp[x];
                        float S = f;
                        float Q = f * f;
                        for (int j = 0; j < K; ++j) {
                            const float* plane = feat + ((size_t)nbr[(size_t)n * K + j] * C + c) * HW;
                            float wv = orc_sample(plane, &row[(size_t)j * W + x]);
                            S = S + wv;
                            if (mode == 0) {
                                float w2 = wv * wv;
                                Q = Q + w2;
                            } else {
                                Q = fmaf(wv, wv, Q);
                            }
                        }
                        if (mode == 0) {
                            float q = Q / nviews;
                            float m = S / nviews;
                            float m2 = m * m;
                            o[x] = q - m2;
                        } else {
                            float m = S * rcp;
                            o[x] = fmaf(-m, m, Q * rcp);
                        }
                    }
                }
            }
        free(row);
    }
    return 0;
}

/* backward of a3+a4 w.r.t. feat (the grid is built under no_grad, module.py:115).
 * g (N,C,D,H,W) = dL/dvariance; gfeat (N,C,H,W) is overwritten.
 *   dvar/dv = 2 v/(K+1) - 2 S/(K+1)^2 for each of the K+1 contributing values v. */
ORC_API int orc_plane_sweep_variance_bwd(const float* feat, const int64_t* nbr, const float* proj, const float* depth,
                                         const float* g, float* gfeat, int N, int K, int C, int D, int H, int W) {
    if (K < 0 || K > ORC_MAX_K) return 1;
    const size_t HW = (size_t)H * W;
    const double inv = 1.0 / (K + 1);
    double* acc = (double*)calloc((size_t)N * C * HW, sizeof(double));
    if (!acc) return 3;
    orc_taps taps[ORC_MAX_K];
    float wv[ORC_MAX_K];
    for (int n = 0; n < N; ++n)
        for (int d = 0; d < D; ++d)
            for (int y = 0; y < H; ++y)
                for (int x = 0; x < W; ++x) {
                    for (int j = 0; j < K; ++j)
                        orc_compute_taps(proj + ((size_t)n * K + j) * 16, (float)x, (float)y, depth[(size_t)n * D + d], H, W, &taps[j]);
                    for (int c = 0; c < C; ++c) {
                        float f = feat[(((size_t)n * C + c) * H + y) * W + x];
                        double S = f;
                        for (int j = 0; j < K; ++j) {
                            wv[j] = orc_sample(feat + ((size_t)nbr[(size_t)n * K + j] * C + c) * HW, &taps[j]);
                            S += wv[j];
                        }
                        double go = g[((((size_t)n * C + c) * D + d) * H + y) * W + x];
                        acc[(((size_t)n * C + c) * H + y) * W + x] += go * (2.0 * f * inv - 2.0 * S * inv * inv);
                        for (int j = 0; j < K; ++j) {
                            double gw = go * (2.0 * wv[j] * inv - 2.0 * S * inv * inv);
                            double* ap = acc + ((size_t)nbr[(size_t)n * K + j] * C + c) * HW;
                            for (int t = 0; t < 4; ++t) ap[taps[j].off[t]] += gw * taps[j].w[t];
                        }
                    }
                }
    for (size_t i = 0; i < (size_t)N * C * HW; ++i) gfeat[i] = (float)acc[i];
    free(acc);
    return 0;
}

/* ------------------------------------------------------------------------------------------
 * a5-a7: softmax / sigmoid (mvsdet.py:470-475), sample_depth_prob (mvsdet.py:266-283),
 * compute_avg_depth (mvsdet.py:298-317).
 * cost_reg, off_logit (N,D,H,W) -> prob, off (N,D,H,W); est_depth, est_dens (N,topk,H,W);
 * avg_depth (N,H,W).  near = near_far_range[0], interval = (far-near)/D as fp32 scalars
 * (mvsdet.py:278-280: int64 index * python float -> fp32, + torch.tensor(near) fp32).
 * Top-k ties are broken towards the lower plane index (torch.topk leaves it unspecified).
 * avg_depth sums in plane order (the reference sums in probability order; <= 2e-6 apart). */
ORC_API int orc_depth_prob_topk(const float* cost_reg, const float* off_logit, float* prob, float* off,
                                float* est_depth, float* est_dens, int32_t* est_idx, float* avg_depth,
                                int N, int D, int H, int W, int topk, float near, float interval) {
    if (topk > D || topk > 16) return 1;
    const size_t HW = (size_t)H * W;
#pragma omp parallel for schedule(static)
    for (int n = 0; n < N; ++n) {
        float* e = (float*)malloc(sizeof(float) * (size_t)D);
        for (size_t p = 0; p < HW; ++p) {
            const float* c = cost_reg + (size_t)n * D * HW + p;
            const float* o = off_logit + (size_t)n * D * HW + p;
            float m = c[0];
            for (int d = 1; d < D; ++d) m = c[d * HW] > m ? c[d * HW] : m;
            float s = 0.0f;
            for (int d = 0; d < D; ++d) {
                e[d] = expf(c[d * HW] - m);
                s += e[d];
            }
            float avg = 0.0f;
            for (int d = 0; d < D; ++d) {
                float pd = e[d] / s;
                float od = 1.0f / (1.0f + expf(-o[d * HW]));
                prob[(size_t)n * D * HW + d * HW + p] = pd;
                off[(size_t)n * D * HW + d * HW + p] = od;
                e[d] = pd;
                float dep = (float)d * interval;
                dep = dep + near;
                float od_i = od * interval;
                dep = dep + od_i;
                float t = dep * pd;
                avg = avg + t;
            }
            avg_depth[(size_t)n * HW + p] = avg;
            int chosen[16];
            for (int k = 0; k < topk; ++k) {
                int best = -1;
                for (int d = 0; d < D; ++d) {
                    int used = 0;
                    for (int q = 0; q < k; ++q) used |= (chosen[q] == d);
                    if (used) continue;
                    if (best < 0 || e[d] > e[best]) best = d;
                }
                chosen[k] = best;
                float od = off[(size_t)n * D * HW + best * HW + p];
                float dep = (float)best * interval;
                dep = dep + near;
                float od_i = od * interval;
                dep = dep + od_i;
                est_depth[((size_t)n * topk + k) * HW + p] = dep;
                est_dens[((size_t)n * topk + k) * HW + p] = e[best];
                if (est_idx) est_idx[((size_t)n * topk + k) * HW + p] = best;
            }
        }
        free(e);
    }
    return 0;
}

/* backward of a5-a7: gradients of prob, est_depth, est_dens, avg_depth into cost_reg, off_logit.
 * The top-k indices carry no gradient.  Any g* pointer may be NULL (= zero gradient). */
ORC_API int orc_depth_prob_topk_bwd(const float* prob, const float* off, const int32_t* est_idx,
                                    const float* g_prob, const float* g_depth, const float* g_dens, const float* g_avg,
                                    float* g_cost, float* g_offlogit, int N, int D, int H, int W, int topk,
                                    float near, float interval) {
    const size_t HW = (size_t)H * W;
#pragma omp parallel for schedule(static)
    for (int n = 0; n < N; ++n) {
        double* gp = (double*)malloc(sizeof(double) * (size_t)D);
        double* go = (double*)malloc(sizeof(double) * (size_t)D);
        for (size_t p = 0; p < HW; ++p) {
            for (int d = 0; d < D; ++d) {
                size_t i = (size_t)n * D * HW + d * HW + p;
                double dep = (double)d * interval + near + (double)off[i] * interval;
                gp[d] = (g_prob ? g_prob[i] : 0.0) + (g_avg ? (double)g_avg[(size_t)n * HW + p] * dep : 0.0);
                go[d] = g_avg ? (double)g_avg[(size_t)n * HW + p] * prob[i] * interval : 0.0;
            }
            for (int k = 0; k < topk; ++k) {
                int d = est_idx[((size_t)n * topk + k) * HW + p];
                if (g_dens) gp[d] += g_dens[((size_t)n * topk + k) * HW + p];
                if (g_depth) go[d] += (double)g_depth[((size_t)n * topk + k) * HW + p] * interval;
            }
            double dot = 0.0;
            for (int d = 0; d < D; ++d) dot += gp[d] * prob[(size_t)n * D * HW + d * HW + p];
            for (int d = 0; d < D; ++d) {
                size_t i = (size_t)n * D * HW + d * HW + p;
                g_cost[i] = (float)(prob[i] * (gp[d] - dot));
                g_offlogit[i] = (float)(go[d] * off[i] * (1.0 - off[i]));
            }
        }
        free(gp);
        free(go);
    }
    return 0;
}

/* ------------------------------------------------------------------------------------------
 * a9: backproject_Weigh, mvsdet.py:1372-1492 (gt_depth=None branch).
 * features: element (i,c,y,x) at feat[i*fs[0] + c*fs[1] + y*fs[2] + x*fs[3]] (the reference passes the
 *           non-contiguous crop feature[:, :, :h, :w], mvsdet.py:499).
 * points (3,V) voxel coordinates from get_points (mvsdet.py:1316-1327); projection (N,3,4).
 * depth/dens: candidate j of pixel (y,x) of view i at ptr[i*ds[0] + j*ds[1] + y*ds[2] + x*ds[3]]
 *           (the reference's (N, h*w, 1, J) tensors are transposed views of (N,J,h,w)).
 * vz = voxel_size[-1].
 * Outputs: volume (N,C,V) fp32, valid (N,V) u8; optional xi, yi (N,V) int32 rounded pixel
 * coordinates (clamped to +-2^30 when out of int range) and zf (N,V) for the index tests.
 *
 * q = projection @ [p;1] is summed in k order with fused multiply-adds, ((P0*x (+) P1*y) (+) P2*z) + P3,
 * as a GEMM inner loop does.  torch.bmm leaves the fp32 summation order to the BLAS; this
 * restatement and the HIP kernel fix this one (SURVEY.md section 7 hard part ii). */
static inline void orc_project(const float* P, float px, float py, float pz, int h, int w,
                               float* xr, float* yr, float* z, int* valid) {
    float q0 = fmaf(P[2], pz, fmaf(P[1], py, P[0] * px)) + P[3];
    float q1 = fmaf(P[6], pz, fmaf(P[5], py, P[4] * px)) + P[7];
    float q2 = fmaf(P[10], pz, fmaf(P[9], py, P[8] * px)) + P[11];
    /* mvsdet.py:1388-1390  x = (q0/q2).round().long()  -- round half to even */
    *xr = rintf(q0 / q2);
    *yr = rintf(q1 / q2);
    *z = q2;
    /* mvsdet.py:1391 */
    *valid = (*xr >= 0.0f) && (*yr >= 0.0f) && (*xr < (float)w) && (*yr < (float)h) && (q2 > 0.0f);
}

static inline int orc_clamp_i32(float v) {
    if (!(v == v)) return INT32_MIN;
    if (v > 1073741824.0f) return 1073741824;
    if (v < -1073741824.0f) return -1073741824;
    return (int)v;
}

/* weight and valid' of one (view, voxel) pair: mvsdet.py:1393-1428 */
static inline int orc_depth_window(const float* depth, const float* dens, const int64_t* ds, int J,
                                   int yi, int xi, float z, float vz, float* weight) {
    /* mvsdet.py:1395-1396  prob_norm = prob / prob.sum(-1) */
    float psum = 0.0f;
    for (int j = 0; j < J; ++j) psum = psum + dens[j * ds[1] + yi * ds[2] + xi * ds[3]];
    float wmax = 0.0f;
    int any = 0;
    for (int j = 0; j < J; ++j) {
        float dj = depth[j * ds[1] + yi * ds[2] + xi * ds[3]];
        float lo = dj - vz, hi = dj + vz;
        int m = (z > lo) && (z < hi); /* mvsdet.py:1407-1408, open window */
        float pn = dens[j * ds[1] + yi * ds[2] + xi * ds[3]] / psum;
        float cand = m ? pn : 0.0f;  /* mvsdet.py:1410-1411 */
        wmax = cand > wmax ? cand : wmax; /* mvsdet.py:1421-1422 */
        if (cand != cand) wmax = cand;    /* torch.max propagates NaN */
        any |= m;                         /* mvsdet.py:1415-1418 */
    }
    *weight = wmax;
    return any;
}

ORC_API int orc_backproject_weigh(const float* feat, const int64_t* fs, const float* points, const float* projection,
                                  const float* depth, const float* dens, const int64_t* ds,
                                  float* volume, uint8_t* valid_out, int32_t* xi_out, int32_t* yi_out, float* z_out,
                                  int N, int C, int h, int w, int V, int J, float vz) {
#pragma omp parallel for schedule(static)
    for (int i = 0; i < N; ++i) {
        const float* P = projection + (size_t)i * 12;
        for (int v = 0; v < V; ++v) {
            float xr, yr, z, wgt = 0.0f;
            int ok;
            orc_project(P, points[v], points[(size_t)V + v], points[2 * (size_t)V + v], h, w, &xr, &yr, &z, &ok);
            if (xi_out) xi_out[(size_t)i * V + v] = orc_clamp_i32(xr);
            if (yi_out) yi_out[(size_t)i * V + v] = orc_clamp_i32(yr);
            if (z_out) z_out[(size_t)i * V + v] = z;
            int xi = 0, yi = 0;
            if (ok) {
                xi = (int)xr;
                yi = (int)yr;
                ok = orc_depth_window(depth + (size_t)i * ds[0], dens + (size_t)i * ds[0], ds, J, yi, xi, z, vz, &wgt);
            }
            valid_out[(size_t)i * V + v] = (uint8_t)ok;
            /* mvsdet.py:1457-1460  volume[i,:,valid] = features[i,:,y,x]; volume[i] *= prob_volume[i] */
            for (int c = 0; c < C; ++c) {
                float val = 0.0f;
                if (ok) val = feat[(size_t)i * fs[0] + (size_t)c * fs[1] + (size_t)yi * fs[2] + (size_t)xi * fs[3]] * wgt;
                volume[((size_t)i * C + c) * V + v] = val;
            }
        }
    }
    return 0;
}

/* a9+a10 fused: mvsdet.py:511-515  volume_mean = sum_i volume_i / (sum_i valid_i + 1e-8), 0 where no view
 * is valid.  mean (C,V) fp32; count (V) int32.  Views are added in index order. */
ORC_API int orc_backproject_weigh_mean(const float* feat, const int64_t* fs, const float* points, const float* projection,
                                       const float* depth, const float* dens, const int64_t* ds,
                                       float* mean, int32_t* count, int N, int C, int h, int w, int V, int J, float vz) {
#pragma omp parallel for schedule(static)
    for (int v = 0; v < V; ++v) {
        int cnt = 0;
        for (int c = 0; c < C; ++c) mean[(size_t)c * V + v] = 0.0f;
        for (int i = 0; i < N; ++i) {
            float xr, yr, z, wgt = 0.0f;
            int ok;
            orc_project(projection + (size_t)i * 12, points[v], points[(size_t)V + v], points[2 * (size_t)V + v], h, w,
                        &xr, &yr, &z, &ok);
            if (!ok) continue;
            int xi = (int)xr, yi = (int)yr;
            ok = orc_depth_window(depth + (size_t)i * ds[0], dens + (size_t)i * ds[0], ds, J, yi, xi, z, vz, &wgt);
            if (!ok) continue;
            ++cnt;
            for (int c = 0; c < C; ++c) {
                float t = feat[(size_t)i * fs[0] + (size_t)c * fs[1] + (size_t)yi * fs[2] + (size_t)xi * fs[3]] * wgt;
                mean[(size_t)c * V + v] = mean[(size_t)c * V + v] + t;
            }
        }
        count[v] = cnt;
        if (cnt > 0) {
            float den = (float)cnt + 1e-8f;
            for (int c = 0; c < C; ++c) mean[(size_t)c * V + v] = mean[(size_t)c * V + v] / den;
        }
    }
    return 0;
}

/* backward of a9 (per-view volume) w.r.t. features and prob (mvsdet.py:1372 docstring; depth has no
 * gradient).  g (N,C,V) = dL/dvolume; gfeat has the (N,C,h,w) contiguous layout and is overwritten;
 * gdens (N,J,h,w) contiguous, overwritten. */
ORC_API int orc_backproject_weigh_bwd(const float* feat, const int64_t* fs, const float* points, const float* projection,
                                      const float* depth, const float* dens, const int64_t* ds, const float* g,
                                      float* gfeat, float* gdens, int N, int C, int h, int w, int V, int J, float vz) {
    const size_t hw = (size_t)h * w;
    double* af = (double*)calloc((size_t)N * C * hw, sizeof(double));
    double* ad = (double*)calloc((size_t)N * J * hw, sizeof(double));
    if (!af || !ad) return 3;
    for (int i = 0; i < N; ++i)
        for (int v = 0; v < V; ++v) {
            float xr, yr, z;
            int ok;
            orc_project(projection + (size_t)i * 12, points[v], points[(size_t)V + v], points[2 * (size_t)V + v], h, w,
                        &xr, &yr, &z, &ok);
            if (!ok) continue;
            int xi = (int)xr, yi = (int)yr;
            const float* dp = depth + (size_t)i * ds[0];
            const float* pp = dens + (size_t)i * ds[0];
            float psum = 0.0f;
            for (int j = 0; j < J; ++j) psum = psum + pp[j * ds[1] + yi * ds[2] + xi * ds[3]];
            int any = 0, arg = -1;
            float wmax = 0.0f;
            for (int j = 0; j < J; ++j) {
                float dj = dp[j * ds[1] + yi * ds[2] + xi * ds[3]];
                int m = (z > dj - vz) && (z < dj + vz);
                float pn = pp[j * ds[1] + yi * ds[2] + xi * ds[3]] / psum;
                any |= m;
                /* torch.max(dim) backward sends the gradient to the first maximal entry */
                if (m && pn > wmax) { wmax = pn; arg = j; }
            }
            if (!any) continue;
            double gw = 0.0;
            for (int c = 0; c < C; ++c) {
                double go = g[((size_t)i * C + c) * V + v];
                double f = feat[(size_t)i * fs[0] + (size_t)c * fs[1] + (size_t)yi * fs[2] + (size_t)xi * fs[3]];
                af[((size_t)i * C + c) * hw + (size_t)yi * w + xi] += go * wmax;
                gw += go * f;
            }
            if (arg >= 0) {
                /* w = p_arg / psum: dw/dp_j = (delta_j,arg - p_arg/psum) / psum */
                for (int j = 0; j < J; ++j) {
                    double dj = ((j == arg) ? 1.0 : 0.0) - (double)wmax;
                    ad[((size_t)i * J + j) * hw + (size_t)yi * w + xi] += gw * dj / psum;
                }
            }
        }
    for (size_t k = 0; k < (size_t)N * C * hw; ++k) gfeat[k] = (float)af[k];
    for (size_t k = 0; k < (size_t)N * J * hw; ++k) gdens[k] = (float)ad[k];
    free(af);
    free(ad);
    return 0;
}
