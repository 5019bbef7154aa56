// The 3-D neck's two GEMM-shaped layers on the bf16 matrix cores with three-term split operands:
//
//   * the 1x1x1 stride-2 shortcut of a down-sampling ResModule (mmdet3d/models/necks/imvoxel_neck.py:196-217 `downsample`):
//       out[n][o][d][h][w] = sum_c W[o][c] x[n][c][2d][2h][2w] + bias[o]                        (BatchNorm folded into W, bias)
//   * the kernel-2 stride-2 transposed convolution of an up block (imvoxel_neck.py:166-180, first three layers):
//       out[n][o][2d+p][2h+q][2w+r] = relu(sum_c x[n][c][d][h][w] W[c][o][p][q][r] + bias[o])   -- 8 single-tap classes
//
// Both are C[M][v] = A[M][K] B[K][v] with B = the activations as they lie in memory (NCDHW: a channel is a row of voxels).
// Rounds 2-4 ran them as rocBLAS fp32 GEMMs plus ATen glue around them (the strided sub-sampling copy, the bias broadcast, a
// permuting clamp for the 2x2x2 interleave: four Cijk kernels, ~20 small launches and ~0.35 ms of a 2.2 ms neck).  Here:
// one kernel per layer, bias, ReLU and the interleave in its epilogue, the sub-sampling in its gather.
//
// Block = 128 rows (M) x 64 voxels, 4 waves; wave w = rows 32w .. 32w+31 x both 32-voxel column tiles (2 accumulators of
// v_mfma_f32_32x32x16_bf16); K in steps of 32 channels.  A: the weight matrix cut into bf16 hi / mid ONCE per weight version
// (mvsdet_gemm_split_weight, kept on the module) in fragment order [M/32][K/16][piece][64 lanes][8]: a wave's fragment is 1 KiB
// of one coalesced load from L2, no LDS.  B: a thread fetches the 8 channels of its voxel (coalesced along the voxels), cuts
// them and writes two 16-byte units; fragments are conflict-free ds_read_b128.  Double-buffered LDS (16 KiB), the next step's
// channel values fetched into registers before this step's MFMAs.
//
// Row order for the transposed layer: m = 8 o + 4 p + 2 q + r, so that the 32x32 accumulator's rows (reg & 3) + 4 (lane >> 5)
// + 8 (reg >> 2) give a lane four output channels (reg >> 2) at depth parity p = lane >> 5 with (q, r) = reg & 3: a float2
// store {r = 0, 1} per (channel, q), contiguous along w across the lanes of a row.
#include "common.h"

namespace mvsdet {

typedef short ng_bf16x8 __attribute__((ext_vector_type(8)));
typedef float ng_f32x16 __attribute__((ext_vector_type(16)));
typedef unsigned ng_u32x4 __attribute__((ext_vector_type(4)));

constexpr int kNgBM = 128, kNgBN = 64, kNgBK = 32;

__device__ __forceinline__ unsigned ng_pack(float a, float b) {
    typedef __bf16 bf2 __attribute__((ext_vector_type(2)));
    const bf2 v = {(__bf16)a, (__bf16)b};
    return __builtin_bit_cast(unsigned, v);
}
// 8 floats -> their bf16 roundings (hi) and the roundings of the exact remainders (mid)
__device__ __forceinline__ void ng_cut8(const float (&f)[8], ng_u32x4& hi, ng_u32x4& mid) {
#pragma unroll
    for (int p = 0; p < 4; ++p) {
        hi[p] = ng_pack(f[2 * p], f[2 * p + 1]);
        mid[p] = ng_pack(f[2 * p] - __uint_as_float(hi[p] << 16), f[2 * p + 1] - __uint_as_float(hi[p] & 0xffff0000u));
    }
}

// wmat (M, K) fp32 row-major -> [M/32][K/16][2 pieces][64 lanes][8 bf16]: lane = 32 * (k-group of 8) + row of the tile
__global__ __launch_bounds__(kThreads) void gemm_split_weight_kernel(const float* __restrict__ wmat, uint4* __restrict__ ws, int M, int K) {
    const size_t u = (size_t)blockIdx.x * kThreads + threadIdx.x;
    const size_t units = (size_t)(M / 32) * (K / 16) * 64;
    if (u >= units) return;
    const int lane = (int)(u & 63);
    const size_t t = u >> 6;
    const int k16 = (int)(t % (K / 16)), rt = (int)(t / (K / 16));
    const float* src = wmat + (size_t)(rt * 32 + (lane & 31)) * K + k16 * 16 + 8 * (lane >> 5);
    float f[8];
#pragma unroll
    for (int j = 0; j < 8; ++j) f[j] = src[j];
    ng_u32x4 hi, mid;
    ng_cut8(f, hi, mid);
    ws[(t * 2 + 0) * 64 + lane] = make_uint4(hi[0], hi[1], hi[2], hi[3]);
    ws[(t * 2 + 1) * 64 + lane] = make_uint4(mid[0], mid[1], mid[2], mid[3]);
}

// MODE 0: pointwise stride-2 convolution (x (N,K,D,H,W), out (N,M,D/2,H/2,W/2), voxel = coarse voxel, gather at 2x);
// MODE 1: transposed k2 s2 (x (N,K,D,H,W), rows m = 8 o + 4 p + 2 q + r, out (N,M/8,2D,2H,2W))
template <int MODE>
__global__ __launch_bounds__(kThreads) void neck_gemm_bf16x3_kernel(const float* __restrict__ x, const uint4* __restrict__ ws,
                                                                    const float* __restrict__ bias, float* __restrict__ out, int M,
                                                                    int K, int D, int H, int W, int relu) {
    __shared__ uint4 s_b[2][2][4][kNgBN];   // [stage][piece][k-group of 8][voxel]
    const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int n = blockIdx.z, m0 = blockIdx.y * kNgBM, v0 = blockIdx.x * kNgBN;
    // the voxel grid the columns run over: MODE 0 the coarse output grid, MODE 1 the input grid
    const int Dv = MODE == 0 ? D / 2 : D, Hv = MODE == 0 ? H / 2 : H, Wv = MODE == 0 ? W / 2 : W;
    const int V = Dv * Hv * Wv;
    const size_t plane = (size_t)D * H * W;

    // staging duty: voxel v0 + (tid & 63), channels 8 * (tid >> 6) .. + 7 of the step
    const int sv = v0 + (tid & 63), skq = tid >> 6;
    const bool sv_ok = sv < V;
    size_t soff = 0;
    if (sv_ok) {
        if (MODE == 0) {
            const int d = sv / (Hv * Wv), r = sv - d * Hv * Wv, h = r / Wv, w = r - h * Wv;
            soff = ((size_t)2 * d * H + 2 * h) * W + 2 * w;
        } else {
            soff = (size_t)sv;
        }
    }
    const float* xs = x + ((size_t)n * K + 8 * skq) * plane + soff;
    float f[8];
    auto fetch = [&](int ks) {
#pragma unroll
        for (int j = 0; j < 8; ++j) f[j] = sv_ok ? xs[(size_t)(ks * kNgBK + j) * plane] : 0.0f;
    };
    auto put = [&](int buf) {
        ng_u32x4 hi, mid;
        ng_cut8(f, hi, mid);
        s_b[buf][0][skq][tid & 63] = make_uint4(hi[0], hi[1], hi[2], hi[3]);
        s_b[buf][1][skq][tid & 63] = make_uint4(mid[0], mid[1], mid[2], mid[3]);
    };

    ng_f32x16 acc[2];
#pragma unroll
    for (int c = 0; c < 2; ++c)
#pragma unroll
        for (int i = 0; i < 16; ++i) acc[c][i] = 0.0f;

    const int rt = (m0 >> 5) + wave;                     // the wave's row tile
    const uint4* wa = ws + (size_t)rt * (K / 16) * 2 * 64 + lane;
    const int nks = K / kNgBK;
    fetch(0);
    put(0);
    __syncthreads();
    for (int ks = 0; ks < nks; ++ks) {
        const int buf = ks & 1;
        if (ks + 1 < nks) fetch(ks + 1);                 // in flight under the MFMAs below
        uint4 a_hi[2], a_mid[2];
#pragma unroll
        for (int s = 0; s < 2; ++s) {
            a_hi[s] = wa[((size_t)(2 * ks + s) * 2 + 0) * 64];
            a_mid[s] = wa[((size_t)(2 * ks + s) * 2 + 1) * 64];
        }
#pragma unroll
        for (int s = 0; s < 2; ++s) {
            const ng_bf16x8 Ah = __builtin_bit_cast(ng_bf16x8, a_hi[s]), Am = __builtin_bit_cast(ng_bf16x8, a_mid[s]);
#pragma unroll
            for (int c = 0; c < 2; ++c) {
                const int kq = 2 * s + (lane >> 5), col = 32 * c + (lane & 31);
                const ng_bf16x8 Bh = __builtin_bit_cast(ng_bf16x8, s_b[buf][0][kq][col]);
                const ng_bf16x8 Bm = __builtin_bit_cast(ng_bf16x8, s_b[buf][1][kq][col]);
                acc[c] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(Am, Bh, acc[c], 0, 0, 0);   // small terms first
                acc[c] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(Ah, Bm, acc[c], 0, 0, 0);
                acc[c] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(Ah, Bh, acc[c], 0, 0, 0);
            }
        }
        if (ks + 1 < nks) put(buf ^ 1);                  // the other stage: its readers finished before the last barrier
        __syncthreads();
    }

    // ---- epilogue
    const int half = lane >> 5;
#pragma unroll
    for (int c = 0; c < 2; ++c) {
        const int v = v0 + 32 * c + (lane & 31);
        if (v >= V) continue;
        if (MODE == 0) {
            float* o = out + ((size_t)n * M + m0 + 32 * wave) * V + v;
#pragma unroll
            for (int i = 0; i < 16; ++i) {
                const int row = (i & 3) + 8 * (i >> 2) + 4 * half;
                float val = acc[c][i] + bias[m0 + 32 * wave + row];
                if (relu) val = fmaxf(val, 0.0f);
                o[(size_t)row * V] = val;
            }
        } else {
            const int d = v / (Hv * Wv), r = v - d * Hv * Wv, h = r / Wv, w = r - h * Wv;
            const int Co = M / 8, H2 = 2 * H, W2 = 2 * W;
            const int o0 = (m0 + 32 * wave) / 8;          // four output channels per row tile
#pragma unroll
            for (int ch = 0; ch < 4; ++ch) {
                const float b = bias[o0 + ch];
                float* o = out + (((size_t)n * Co + o0 + ch) * (2 * D) + 2 * d + half) * H2 * W2 + (size_t)(2 * h) * W2 + 2 * w;
#pragma unroll
                for (int q = 0; q < 2; ++q) {
                    float a0 = acc[c][4 * ch + 2 * q] + b, a1 = acc[c][4 * ch + 2 * q + 1] + b;
                    if (relu) { a0 = fmaxf(a0, 0.0f); a1 = fmaxf(a1, 0.0f); }
                    *reinterpret_cast<float2*>(o + (size_t)q * W2) = make_float2(a0, a1);
                }
            }
        }
    }
}

}  // namespace mvsdet

using namespace mvsdet;

extern "C" size_t mvsdet_gemm_split_weight_bytes(int M, int K) {
    if (M <= 0 || K <= 0 || M % kNgBM || K % kNgBK) return 0;
    return (size_t)(M / 32) * (K / 16) * 2 * 64 * 16;
}

extern "C" int mvsdet_gemm_split_weight(const float* wmat, void* wsplit, int M, int K, mvsdet_stream_t stream) {
    MVS_REQUIRE(wmat && wsplit, "gemm_split_weight: NULL pointer");
    MVS_REQUIRE(M > 0 && K > 0 && M % kNgBM == 0 && K % kNgBK == 0, "gemm_split_weight: M=%d must be a multiple of %d, K=%d of %d", M, kNgBM, K, kNgBK);
    MVS_REQUIRE(((uintptr_t)wsplit & 15u) == 0, "gemm_split_weight: wsplit must be 16-byte aligned");
    const size_t units = (size_t)(M / 32) * (K / 16) * 64;
    hipLaunchKernelGGL(gemm_split_weight_kernel, dim3((unsigned)((units + kThreads - 1) / kThreads)), dim3(kThreads), 0, (hipStream_t)stream,
                       wmat, static_cast<uint4*>(wsplit), M, K);
    MVS_LAUNCH_CHECK("gemm_split_weight");
    return MVSDET_OK;
}

static int neck_gemm_check(const char* name, const void* x, const void* ws, const void* bias, const void* out, int N, int K, int M,
                           int D, int H, int W) {
    MVS_REQUIRE(x && ws && bias && out, "%s: NULL pointer", name);
    MVS_REQUIRE(N > 0 && N <= 65535 && D > 0 && H > 0 && W > 0, "%s: bad shape N=%d D=%d H=%d W=%d", name, N, D, H, W);
    MVS_REQUIRE(K > 0 && K % kNgBK == 0 && M > 0 && M % kNgBM == 0, "%s: Cin=%d must be a multiple of %d and the row count %d of %d", name, K,
                kNgBK, M, kNgBM);
    MVS_REQUIRE(M / kNgBM <= 65535, "%s: too many rows", name);
    MVS_REQUIRE(((uintptr_t)ws & 15u) == 0 && ((uintptr_t)out & 7u) == 0, "%s: wsplit must be 16-byte, out 8-byte aligned", name);
    MVS_REQUIRE((long long)D * H * W < INT32_MAX / 8, "%s: volume too large", name);
    return MVSDET_OK;
}

extern "C" int mvsdet_conv3d_k1_s2_bf16x3(const float* x, const void* wsplit, const float* bias, float* out, int N, int Cin, int Cout,
                                          int D, int H, int W, int relu, mvsdet_stream_t stream) {
    if (int rc = neck_gemm_check("conv3d_k1_s2_bf16x3", x, wsplit, bias, out, N, Cin, Cout, D, H, W)) return rc;
    MVS_REQUIRE(D % 2 == 0 && H % 2 == 0 && W % 2 == 0, "conv3d_k1_s2_bf16x3: D, H, W must be even");
    const int V = (D / 2) * (H / 2) * (W / 2);
    dim3 grid((unsigned)((V + kNgBN - 1) / kNgBN), (unsigned)(Cout / kNgBM), (unsigned)N);
    hipLaunchKernelGGL(neck_gemm_bf16x3_kernel<0>, grid, dim3(kThreads), 0, (hipStream_t)stream, x, static_cast<const uint4*>(wsplit), bias,
                       out, Cout, Cin, D, H, W, relu);
    MVS_LAUNCH_CHECK("conv3d_k1_s2_bf16x3");
    return MVSDET_OK;
}

extern "C" int mvsdet_convT3d_k2_s2_bf16x3(const float* x, const void* wsplit, const float* bias, float* out, int N, int Cin, int Cout,
                                           int D, int H, int W, int relu, mvsdet_stream_t stream) {
    MVS_REQUIRE(Cout > 0 && Cout <= INT32_MAX / 8, "convT3d_k2_s2_bf16x3: bad Cout");
    if (int rc = neck_gemm_check("convT3d_k2_s2_bf16x3", x, wsplit, bias, out, N, Cin, 8 * Cout, D, H, W)) return rc;
    const int V = D * H * W;
    dim3 grid((unsigned)((V + kNgBN - 1) / kNgBN), (unsigned)(8 * Cout / kNgBM), (unsigned)N);
    hipLaunchKernelGGL(neck_gemm_bf16x3_kernel<1>, grid, dim3(kThreads), 0, (hipStream_t)stream, x, static_cast<const uint4*>(wsplit), bias,
                       out, 8 * Cout, Cin, D, H, W, relu);
    MVS_LAUNCH_CHECK("convT3d_k2_s2_bf16x3");
    return MVSDET_OK;
}
