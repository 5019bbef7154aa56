// Stage 1 of the MVSDet hot path on gfx950: feature packing, homography warp (a3) and the fused
// plane-sweep variance cost volume (a3+a4).  Reference: mvs_models/module.py:105-146 and
// mvsdet.py:439-467 of Pixie8888/MVSDet (projects/NeRF-Det/nerfdet/).
//
// Roofline: HBM.  Per cost volume the kernel must write C*D*H*W*4 B and read (K+1)*C*H*W*4 B; there
// is no dense contraction, so MFMA does not apply.  The design problem is the gather:
// 4*K bilinear taps per output element.  Reading them from NCHW planes costs one uncoalesced dword
// load per tap per channel; instead the maps are re-laid channel-last ("packed", include/mvsdet_hip.h)
// so that ONE 16-byte-per-lane wave load fetches a tap for 256 channels as a contiguous 1 KiB run,
// and the results are transposed through LDS so the (N,C,D,H,W) output is still written as full
// 256-byte rows along W.
#include "common.h"
#include "pack.h"

namespace mvsdet {

// ---------------------------------------------------------------------------------------------
// a3: homo_warping with the reference's own layout (NCHW source, one source view per batch row).
// Compatibility operator: one thread per output pixel, loop over channels; stores are coalesced
// along W, taps are per-lane dword gathers.  The fused kernel below is the fast path.
// ---------------------------------------------------------------------------------------------
__global__ __launch_bounds__(kThreads) void homo_warp_kernel(const float* __restrict__ src,
                                                              const float* __restrict__ proj,
                                                              const float* __restrict__ depth,
                                                              float* __restrict__ out, int C, int D, int H, int W) {
    const int HW = H * W;
    const int pix = blockIdx.x * kThreads + threadIdx.x;
    const int d = blockIdx.y, b = blockIdx.z;
    if (pix >= HW) return;
    const int y = pix / W, x = pix - y * W;
    int4 off;
    float4 w;
    compute_taps(proj + (size_t)b * 16, (float)x, (float)y, depth[(size_t)b * D + d], H, W, 1, off, w);
    const float* plane = src + (size_t)b * C * HW;
    float* o = out + ((size_t)b * C * D + d) * HW + pix;
    for (int c = 0; c < C; ++c) {
        float s = plane[off.x] * w.x;
        s = fmaf(plane[off.y], w.y, s);
        s = fmaf(plane[off.z], w.z, s);
        s = fmaf(plane[off.w], w.w, s);
        *o = s;
        plane += HW;
        o += (size_t)D * HW;
    }
}

// ---------------------------------------------------------------------------------------------
// a3+a4 fused: plane-sweep variance.
//
// Work decomposition
//   block  = (reference view n, tile of TP consecutive pixels of the flattened H*W plane, depth chunk)
//            -- out[n,c,d,:,:] is contiguous over (h,w), so a tile is one contiguous run per (c,d)
//   per depth plane:
//     phase 1  K*TP threads build the tap table (4 offsets + 4 weights per pixel and neighbour) in LDS
//     phase 2  each wave walks its TP/4 pixels; lane = (pixel slot, channel group g): 1 + 4K
//              16-byte loads per pixel and lane, each wave-instruction reading contiguous runs of
//              4*LP floats of the packed maps; variance for 4 channels -> LDS tile [channel][pixel]
//     phase 3  LDS rows -> global: one 4*TP-byte contiguous run per (channel, plane)
//   LP (lanes per pixel) = min(64, pow2ceil(G)), so C=256 -> one pixel per wave-instruction.
//
// Arithmetic (device rounding, oracle mode 1): warped = fma chain over the 4 taps; S = f + w1 + ..;
// Q = fma(w,w,Q); var = fma(-m, m, Q*r) with m = S*r, r = 1/(K+1).
// ---------------------------------------------------------------------------------------------
template <int K, int TP, bool NT>
__global__ __launch_bounds__(kThreads) void plane_sweep_variance_kernel(
    const float* __restrict__ packed, const int64_t* __restrict__ nbr, const float* __restrict__ proj,
    const float* __restrict__ depth, float* __restrict__ var, int N, int C, int G, int D, int H, int W, int tiles,
    int d_per_block, int lp_log2) {
    constexpr int KK = K > 0 ? K : 1;
    constexpr int PW = TP / 4;  // pixels per wave
    __shared__ float s_tile[256 * (TP + 1)];
    __shared__ int4 s_off[KK][TP];
    __shared__ float4 s_w[KK][TP];

    const int HW = H * W;
    const int L = xcd_contiguous_id(blockIdx.x, gridDim.x);
    const int n = L / tiles, tile = L - n * tiles;
    const int pix0 = tile * TP;
    const int d_begin = blockIdx.y * d_per_block;
    const int d_end = min(D, d_begin + d_per_block);

    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int LP = 1 << lp_log2, PPI = 64 >> lp_log2;
    const int gl = lane & (LP - 1), ps = lane >> lp_log2;
    const int G4 = 4 * G;
    const size_t view_stride = (size_t)HW * G4;
    const float* ref_base = packed + (size_t)n * view_stride;
    const float* nb_base[KK];
#pragma unroll
    for (int j = 0; j < K; ++j) {
        int64_t v = nbr[(size_t)n * K + j];
        v = v < 0 ? 0 : (v >= N ? N - 1 : v);  // never read outside the packed maps
        nb_base[j] = packed + (size_t)v * view_stride;
    }
    const float rcp = 1.0f / (float)(K + 1);
    const int chunks = (G + 63) / 64;

    for (int d = d_begin; d < d_end; ++d) {
        // ---- phase 1: tap table
        if (K > 0) {
            const float dval = depth[(size_t)n * D + d];
            for (int idx = threadIdx.x; idx < K * TP; idx += kThreads) {
                const int j = idx / TP, p = idx - j * TP;
                const int pix = pix0 + p;
                int4 o = make_int4(0, 0, 0, 0);
                float4 w = make_float4(0.f, 0.f, 0.f, 0.f);
                if (pix < HW) {
                    const int y = pix / W, x = pix - y * W;
                    compute_taps(proj + ((size_t)n * K + j) * 16, (float)x, (float)y, dval, H, W, G4, o, w);
                }
                s_off[j][p] = o;
                s_w[j][p] = w;
            }
        }
        __syncthreads();
        for (int ci = 0; ci < chunks; ++ci) {
            const int rg = min(64, G - ci * 64);  // channel groups in this chunk
            const bool gvalid = gl < rg;
            const int g = ci * 64 + (gvalid ? gl : 0);
            // ---- phase 2: gather + variance -> LDS tile
#pragma unroll 2
            for (int s = 0; s < PW / PPI; ++s) {
                const int p = wave * PW + s * PPI + ps;
                const int pix = min(pix0 + p, HW - 1);
                const float4 f = *reinterpret_cast<const float4*>(ref_base + (size_t)pix * G4 + 4 * g);
                float S0 = f.x, S1 = f.y, S2 = f.z, S3 = f.w;
                float Q0 = f.x * f.x, Q1 = f.y * f.y, Q2 = f.z * f.z, Q3 = f.w * f.w;
#pragma unroll
                for (int j = 0; j < K; ++j) {
                    const int4 o = s_off[j][p];
                    const float4 w = s_w[j][p];
                    const float* b = nb_base[j] + 4 * g;
                    const float4 t0 = *reinterpret_cast<const float4*>(b + o.x);
                    const float4 t1 = *reinterpret_cast<const float4*>(b + o.y);
                    const float4 t2 = *reinterpret_cast<const float4*>(b + o.z);
                    const float4 t3 = *reinterpret_cast<const float4*>(b + o.w);
                    float v0 = t0.x * w.x, v1 = t0.y * w.x, v2 = t0.z * w.x, v3 = t0.w * w.x;
                    v0 = fmaf(t1.x, w.y, v0); v1 = fmaf(t1.y, w.y, v1); v2 = fmaf(t1.z, w.y, v2); v3 = fmaf(t1.w, w.y, v3);
                    v0 = fmaf(t2.x, w.z, v0); v1 = fmaf(t2.y, w.z, v1); v2 = fmaf(t2.z, w.z, v2); v3 = fmaf(t2.w, w.z, v3);
                    v0 = fmaf(t3.x, w.w, v0); v1 = fmaf(t3.y, w.w, v1); v2 = fmaf(t3.z, w.w, v2); v3 = fmaf(t3.w, w.w, v3);
                    S0 = S0 + v0; S1 = S1 + v1; S2 = S2 + v2; S3 = S3 + v3;
                    Q0 = fmaf(v0, v0, Q0); Q1 = fmaf(v1, v1, Q1); Q2 = fmaf(v2, v2, Q2); Q3 = fmaf(v3, v3, Q3);
                }
                const float m0 = S0 * rcp, m1 = S1 * rcp, m2 = S2 * rcp, m3 = S3 * rcp;
                if (gvalid) {
                    float* t = s_tile + gl * (TP + 1) + p;
                    t[0] = fmaf(-m0, m0, Q0 * rcp);
                    t[rg * (TP + 1)] = fmaf(-m1, m1, Q1 * rcp);
                    t[2 * rg * (TP + 1)] = fmaf(-m2, m2, Q2 * rcp);
                    t[3 * rg * (TP + 1)] = fmaf(-m3, m3, Q3 * rcp);
                }
            }
            __syncthreads();
            // ---- phase 3: rows of the tile -> (N,C,D,H,W)
            {
                constexpr int RPI = 64 / TP;  // rows per wave-instruction
                const int pp = lane % TP, rsub = lane / TP;
                const int rows = 4 * rg;
                const bool pvalid = pix0 + pp < HW;
                for (int r = wave * RPI + rsub; r < rows; r += 4 * RPI) {
                    const int i = r / rg, gg = r - i * rg;
                    const int c = i * G + ci * 64 + gg;
                    if (c < C && pvalid) {
                        float* dst = var + (((size_t)n * C + c) * D + d) * HW + pix0 + pp;
                        // the cost volume is written once and not re-read by this kernel: a non-temporal store
                        // keeps the output stream from evicting the source maps out of L2 / Infinity Cache
                        if (NT) __builtin_nontemporal_store(s_tile[r * (TP + 1) + pp], dst);
                        else *dst = s_tile[r * (TP + 1) + pp];
                    }
                }
            }
            __syncthreads();
        }
    }
}

}  // namespace mvsdet

using namespace mvsdet;

static int pow2ceil_log2(int v) {
    int l = 0;
    while ((1 << l) < v) ++l;
    return l;
}

extern "C" size_t mvsdet_packed_bytes(int N, int C, int H, int W) {
    if (N <= 0 || C <= 0 || H <= 0 || W <= 0) return 0;
    const size_t G = (size_t)(C + 3) / 4;
    return (size_t)N * H * W * 4 * G * sizeof(float);
}

extern "C" int mvsdet_pack_features_f32(const float* feat, const int64_t* fs, float* packed, int N, int C, int H, int W,
                                        mvsdet_stream_t stream) {
    MVS_REQUIRE(feat && fs && packed, "pack_features: NULL pointer");
    MVS_REQUIRE(N > 0 && C > 0 && H > 0 && W > 0, "pack_features: bad shape N=%d C=%d H=%d W=%d", N, C, H, W);
    MVS_REQUIRE((size_t)H * W * 4 * ((C + 3) / 4) < (size_t)INT32_MAX, "pack_features: one view exceeds 2^31 elements");
    MVS_REQUIRE(N <= 65535, "pack_features: N > 65535");
    const int G = (C + 3) / 4;
    dim3 grid((H * W + 63) / 64, (G + 15) / 16, N);
    hipLaunchKernelGGL(pack_features_kernel, grid, dim3(kThreads), 0, (hipStream_t)stream, feat, fs[0], fs[1], fs[2],
                       fs[3], packed, C, G, H, W);
    MVS_LAUNCH_CHECK("pack_features");
    return MVSDET_OK;
}

extern "C" int mvsdet_homo_warp_f32(const float* src, const float* proj, const float* depth, float* out, int B, int C,
                                    int D, int H, int W, mvsdet_stream_t stream) {
    MVS_REQUIRE(src && proj && depth && out, "homo_warp: NULL pointer");
    MVS_REQUIRE(B > 0 && C > 0 && D > 0 && H > 1 && W > 1, "homo_warp: bad shape B=%d C=%d D=%d H=%d W=%d", B, C, D, H, W);
    MVS_REQUIRE(B <= 65535 && D <= 65535, "homo_warp: B or D > 65535");
    MVS_REQUIRE((size_t)C * H * W < (size_t)INT32_MAX, "homo_warp: one view exceeds 2^31 elements");
    dim3 grid((H * W + kThreads - 1) / kThreads, D, B);
    hipLaunchKernelGGL(homo_warp_kernel, grid, dim3(kThreads), 0, (hipStream_t)stream, src, proj, depth, out, C, D, H, W);
    MVS_LAUNCH_CHECK("homo_warp");
    return MVSDET_OK;
}

namespace {
int g_tile_pixels = 0;  // 0 = heuristic; set through MVSDET_SWEEP_TILE for tuning runs

template <int TP, bool NT>
int launch_sweep(const float* packed, const int64_t* nbr, const float* proj, const float* depth, float* var, int N,
                 int K, int C, int D, int H, int W, hipStream_t stream) {
    const int G = (C + 3) / 4;
    const int HW = H * W;
    const int tiles = (HW + TP - 1) / TP;
    int lp_log2 = pow2ceil_log2(G < 64 ? G : 64);
    // a wave must own at least one whole pixel step: PPI = 64/LP <= TP/4
    while ((64 >> lp_log2) > TP / 4) ++lp_log2;
    const long long nblocks = (long long)N * tiles;
    if (nblocks > INT32_MAX) {
        set_error("plane_sweep_variance: grid too large");
        return MVSDET_ERR_INVALID_ARG;
    }
    // every block sweeps all its planes (taps of neighbouring planes overlap -> L1/L2 reuse) unless the
    // grid would be too small to fill 256 CUs
    int dsplit = 1;
    while (nblocks * dsplit < 2048 && dsplit < D) dsplit *= 2;
    const int d_per_block = (D + dsplit - 1) / dsplit;
    dim3 grid((unsigned)nblocks, (D + d_per_block - 1) / d_per_block);
#define MVS_SWEEP_CASE(KV)                                                                                           \
    case KV:                                                                                                         \
        hipLaunchKernelGGL((plane_sweep_variance_kernel<KV, TP, NT>), grid, dim3(kThreads), 0, stream, packed, nbr, proj, \
                           depth, var, N, C, G, D, H, W, tiles, d_per_block, lp_log2);                                \
        break;
    switch (K) {
        MVS_SWEEP_CASE(0)
        MVS_SWEEP_CASE(1)
        MVS_SWEEP_CASE(2)
        MVS_SWEEP_CASE(3)
        MVS_SWEEP_CASE(4)
    }
#undef MVS_SWEEP_CASE
    MVS_LAUNCH_CHECK("plane_sweep_variance");
    return MVSDET_OK;
}
}  // namespace

extern "C" int mvsdet_plane_sweep_variance_packed_f32(const float* packed, const int64_t* nbr, const float* proj,
                                                      const float* depth, float* var, int N, int K, int C, int D, int H,
                                                      int W, mvsdet_stream_t stream) {
    MVS_REQUIRE(packed && depth && var, "plane_sweep_variance: NULL pointer");
    MVS_REQUIRE(K == 0 || (nbr && proj), "plane_sweep_variance: NULL neighbour arrays with K=%d", K);
    MVS_REQUIRE(N > 0 && C > 0 && D > 0 && H > 1 && W > 1, "plane_sweep_variance: bad shape N=%d C=%d D=%d H=%d W=%d", N,
                C, D, H, W);
    MVS_REQUIRE(K >= 0 && K <= MVSDET_MAX_NEIGHBORS, "plane_sweep_variance: K=%d outside [0,%d]", K, MVSDET_MAX_NEIGHBORS);
    MVS_REQUIRE(D <= 65535, "plane_sweep_variance: D > 65535");
    MVS_REQUIRE((size_t)H * W * 4 * ((C + 3) / 4) < (size_t)INT32_MAX, "plane_sweep_variance: one view exceeds 2^31 elements");
    {  // tuning knobs, read per call so A/B runs can flip them inside one process
        const char* e = getenv("MVSDET_SWEEP_TILE");
        g_tile_pixels = e ? atoi(e) : -1;
    }
    const int tp = (g_tile_pixels == 32 || g_tile_pixels == 64) ? g_tile_pixels : 32;
    const char* ent = getenv("MVSDET_SWEEP_NT");
    const bool nt = ent ? atoi(ent) != 0 : true;
    hipStream_t st = (hipStream_t)stream;
    if (tp == 32) return nt ? launch_sweep<32, true>(packed, nbr, proj, depth, var, N, K, C, D, H, W, st)
                            : launch_sweep<32, false>(packed, nbr, proj, depth, var, N, K, C, D, H, W, st);
    return nt ? launch_sweep<64, true>(packed, nbr, proj, depth, var, N, K, C, D, H, W, st)
              : launch_sweep<64, false>(packed, nbr, proj, depth, var, N, K, C, D, H, W, st);
}

extern "C" int mvsdet_plane_sweep_variance_f32(const float* feat, const int64_t* nbr, const float* proj,
                                               const float* depth, float* var, void* workspace, size_t workspace_bytes,
                                               int N, int K, int C, int D, int H, int W, mvsdet_stream_t stream) {
    MVS_REQUIRE(feat && workspace, "plane_sweep_variance: NULL pointer");
    MVS_REQUIRE(N > 0 && C > 0 && H > 1 && W > 1, "plane_sweep_variance: bad shape");
    if (workspace_bytes < mvsdet_packed_bytes(N, C, H, W)) {
        set_error("plane_sweep_variance: workspace %zu B < %zu B", workspace_bytes, mvsdet_packed_bytes(N, C, H, W));
        return MVSDET_ERR_WORKSPACE;
    }
    const int64_t fs[4] = {(int64_t)C * H * W, (int64_t)H * W, W, 1};
    int rc = mvsdet_pack_features_f32(feat, fs, (float*)workspace, N, C, H, W, stream);
    if (rc) return rc;
    return mvsdet_plane_sweep_variance_packed_f32((const float*)workspace, nbr, proj, depth, var, N, K, C, D, H, W, stream);
}
