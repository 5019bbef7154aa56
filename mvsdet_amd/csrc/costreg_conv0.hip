// First layer of the cost regularisation network (mvs_models/mvsnet.py:76,105): Conv3d(Cin -> 64, kernel 3, padding 1,
// no bias) on the variance volume (N,Cin,D,H,W), optionally followed by a per-channel affine (eval-mode BatchNorm
// folded) and ReLU -- 72 % of the network's 2.8 TFLOP.  Implicit GEMM on the fp32 matrix cores:
//
//   D[o][w] += sum over (channel pair, tap) of A[o][k] * B[k][w]          v_mfma_f32_32x32x2_f32, k = channel in pair
//     A = weights, pre-permuted by the caller to [c][kd][kh][kw][o] so that a channel pair's 2 x 27 x 64 block is one
//         contiguous 13.8 KB run;   B = input row of 32 voxels along w, shifted by the tap
//   block = (view, tile of 2 x 4 x 32 voxels (d,h,w)) x all 64 output channels; wave = 2 rows of 32 voxels x 2 blocks of
//           32 channels = 4 accumulators of 32x32 (64 VGPRs); per channel pair and tap 2 A reads + 2 B reads
//           (ds_read_b32, conflict-free) feed 4 MFMAs (256 matrix-core cycles)
//   per channel pair the halo tile (2 x 4x6x34 floats) and the weight block are staged in LDS; the next pair's values
//   travel global -> registers while the current pair is multiplied.
//   Numerics: an fp32 MFMA is bit for bit a k-ordered fmaf chain (cdna_hip_programming.md), so the result is an fp32
//   FMA sum in (channel, tap) order -- the same products as ATen's convolution in another order.
// Bound: fp32 MFMA, 64 FLOP/clk/SIMD = 157 TFLOP/s; 2.04 TFLOP at (40,256,12,60,80) = 13 ms at peak.
// Measured 15.1 ms = 135 TFLOP/s (MIOpen: 33.6 ms).
#include "common.h"

#include <algorithm>

namespace mvsdet {

typedef float f32x16 __attribute__((ext_vector_type(16)));

// The 32 B-columns of an MFMA are 32 voxels along w (TWC = 32: tile 2 x 4 x 32) or two h-rows of 16 (TWC = 16: tile
// 2 x 8 x 16): the launcher picks the one that pads W less (W = 80: 96 vs 80).
constexpr int kC0D = 2;
constexpr int kC0Out = 64;
constexpr int kC0WPair = 2 * 27 * kC0Out;                            // 3456 floats of weights per channel pair
constexpr int kC0WStage = (kC0WPair / 4 + kThreads - 1) / kThreads;  // 4 float4 of weights per thread and pair

// ST = stride (1 or 2, padding 1): input voxel = ST * output voxel - 1 + tap; (Di,Hi,Wi) input, (D,H,W) output extents.
// TD = tile planes (2 or 4): the 256 output voxels of a block are TD x (256 / (TD * TWC)) x TWC.
template <int TWC, int ST, int TD>
__global__ __launch_bounds__(kThreads, 2) void conv3d_k3_mfma_kernel(
    const float* __restrict__ x, const float4* __restrict__ wperm, const float* __restrict__ residual, const float* __restrict__ scale,
    const float* __restrict__ shift, float* __restrict__ out, int Cin, int Cout, int Di, int Hi, int Wi, int D, int H,
    int W, int tiles_w, int tiles_h, int relu, int nsplit, float* __restrict__ partial, size_t partial_stride) {
    constexpr int kC0W = TWC, kC0H = 256 / (TD * TWC);
    // halo of the input tile: ST*(T-1) + 3 per dimension (stride 1, TD = 2: 4 x 6 x 34 or 4 x 10 x 18)
    constexpr int kC0HD = ST * (TD - 1) + 3, kC0HH = ST * (kC0H - 1) + 3, kC0HW = ST * (kC0W - 1) + 3;
    constexpr int kC0Halo = kC0HD * kC0HH * kC0HW;                       // floats per channel
    constexpr int kC0InStage = (2 * kC0Halo + kThreads - 1) / kThreads;  // input values per thread and pair
    constexpr int kRowsPerCol = 32 / TWC;                                // h-rows covered by the 32 MFMA columns
    // two LDS stages: the next channel pair is written while the current one is multiplied, one barrier per pair
    __shared__ float s_in2[2][2 * kC0Halo];
    __shared__ float4 s_w42[2][kC0WPair / 4];

    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int bw = blockIdx.x % tiles_w, bh = blockIdx.x / tiles_w;
    // blockIdx.z = (view, block of 64 output channels, split of the input channels): Cout > 64 runs as Cout/64
    // independent slices of the weights; a small volume (the 3-D neck: 2-100 tiles) is also split over the channel pairs
    // so that the grid fills the chip, each split writing raw partial sums that splitk_epilogue_kernel adds up
    const int nob = Cout / kC0Out;
    const int split = blockIdx.z % nsplit, zo = blockIdx.z / nsplit;
    const int w0 = bw * kC0W, h0 = bh * kC0H, d0 = blockIdx.y * TD, n = zo / nob, ob64 = zo % nob;
    const size_t plane = (size_t)H * W, vol = (size_t)D * plane;        // output
    const size_t iplane = (size_t)Hi * Wi, ivol = (size_t)Di * iplane;  // input
    const float* xn = x + (size_t)n * Cin * ivol;
    const int npairs = (Cin + 1) / 2;
    const int cp_begin = (int)((long long)npairs * split / nsplit), cp_end = (int)((long long)npairs * (split + 1) / nsplit);

    // this wave's two MFMA column groups g = 2*wave, 2*wave+1 of the tile's 8 (8/TD per plane, both in one plane):
    // plane dz0 = g / (8/TD), h-rows (g % (8/TD))*kRowsPerCol .. ; inside a group column c is voxel
    // (h-row c / TWC, w = c % TWC)
    constexpr int kGroupsPerPlane = 8 / TD;  // 4 or 2
    const int dz0 = (2 * wave) / kGroupsPerPlane, hy0 = ((2 * wave) % kGroupsPerPlane) * kRowsPerCol;
    const int col = lane & 31, kk = lane >> 5;  // MFMA operand lane map: A[i=col][k=kk], B[k=kk][j=col]
    const int chy = col / TWC, cw = col % TWC;

    // staging plan (identical for every channel pair): input element e = tid + 256*k of the 2 x halo values
    int in_off[kC0InStage];
#pragma unroll
    for (int k = 0; k < kC0InStage; ++k) {
        const int e = tid + k * kThreads;
        const int kc = e / kC0Halo, r = e - kc * kC0Halo;
        const int dz = r / (kC0HH * kC0HW), r2 = r - dz * (kC0HH * kC0HW);
        const int hy = r2 / kC0HW, wx = r2 - hy * kC0HW;
        const int d = ST * d0 + dz - 1, h = ST * h0 + hy - 1, w = ST * w0 + wx - 1;
        const bool ok = e < 2 * kC0Halo && d >= 0 && d < Di && h >= 0 && h < Hi && w >= 0 && w < Wi;
        // bit 30 carries the channel of the pair; offsets stay below 2^30 (checked by the launcher)
        in_off[k] = ok ? (int)((size_t)d * iplane + (size_t)h * Wi + w) | (kc << 30) : -1;
    }

    float in_reg[kC0InStage];
    float4 w_reg[kC0WStage];
    auto fetch = [&](int cp) {
        const float* x0 = xn + (size_t)(2 * cp) * ivol;
        const bool has1 = 2 * cp + 1 < Cin;
#pragma unroll
        for (int k = 0; k < kC0InStage; ++k) {
            const int o = in_off[k];
            const int kc = (o >> 30) & 1;
            float v = 0.0f;
            if (o >= 0 && (kc == 0 || has1)) v = x0[(size_t)kc * ivol + (o & 0x3fffffff)];
            in_reg[k] = v;
        }
#pragma unroll
        for (int k = 0; k < kC0WStage; ++k) {
            const int e = tid + k * kThreads;
            // row = (channel of the pair, tap), 16 float4 = this block's 64 of the Cout output channels
            w_reg[k] = e < kC0WPair / 4 ? wperm[((size_t)cp * 54 + (e >> 4)) * (Cout / 4) + ob64 * 16 + (e & 15)]
                                        : make_float4(0.f, 0.f, 0.f, 0.f);
        }
    };

    f32x16 acc[2][2];  // [output-channel block][voxel row]
#pragma unroll
    for (int a = 0; a < 2; ++a)
#pragma unroll
        for (int b = 0; b < 2; ++b)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[a][b][r] = 0.0f;

    auto stage = [&](int buf) {
#pragma unroll
        for (int k = 0; k < kC0InStage; ++k) {
            const int e = tid + k * kThreads;
            if (e < 2 * kC0Halo) s_in2[buf][e] = in_reg[k];
        }
#pragma unroll
        for (int k = 0; k < kC0WStage; ++k) {
            const int e = tid + k * kThreads;
            if (e < kC0WPair / 4) s_w42[buf][e] = w_reg[k];
        }
    };
    fetch(cp_begin);
    stage(0);
    __syncthreads();
    if (cp_begin + 1 < cp_end) fetch(cp_begin + 1);
    for (int cp = cp_begin; cp < cp_end; ++cp) {
        const int buf = (cp - cp_begin) & 1;
        const float* bin = s_in2[buf] + kk * kC0Halo + (ST * dz0 * kC0HH + ST * (hy0 + chy)) * kC0HW + ST * cw;
        const float* ain = reinterpret_cast<const float*>(s_w42[buf]) + kk * 27 * kC0Out + col;
#pragma unroll
        for (int kd = 0; kd < 3; ++kd)
#pragma unroll
            for (int kh = 0; kh < 3; ++kh)
#pragma unroll
                for (int kw = 0; kw < 3; ++kw) {
                    const int tap = (kd * 3 + kh) * 3 + kw;
                    const float a0 = ain[tap * kC0Out], a1 = ain[tap * kC0Out + 32];
                    const float b0 = bin[(kd * kC0HH + kh) * kC0HW + kw];
                    const float b1 = bin[(kd * kC0HH + kh + ST * kRowsPerCol) * kC0HW + kw];
                    acc[0][0] = __builtin_amdgcn_mfma_f32_32x32x2f32(a0, b0, acc[0][0], 0, 0, 0);
                    acc[0][1] = __builtin_amdgcn_mfma_f32_32x32x2f32(a0, b1, acc[0][1], 0, 0, 0);
                    acc[1][0] = __builtin_amdgcn_mfma_f32_32x32x2f32(a1, b0, acc[1][0], 0, 0, 0);
                    acc[1][1] = __builtin_amdgcn_mfma_f32_32x32x2f32(a1, b1, acc[1][1], 0, 0, 0);
                }
        if (cp + 1 < cp_end) {
            // the other stage was last read in iteration cp - 1, which every wave left through the barrier below
            stage(buf ^ 1);
            __syncthreads();
            if (cp + 2 < cp_end) fetch(cp + 2);
        }
    }

    // epilogue: C/D map of the 32x32 MFMA: column = lane & 31 (voxel along w), row = (reg & 3) + 8*(reg >> 2) + 4*(lane >> 5)
    const int w = w0 + cw;
#pragma unroll
    for (int rb = 0; rb < 2; ++rb) {
        const int d = d0 + dz0, h = h0 + hy0 + rb * kRowsPerCol + chy;
        if (d >= D || h >= H || w >= W) continue;
#pragma unroll
        for (int ob = 0; ob < 2; ++ob) {
            // affine and residual of the 16 outputs are requested together, ahead of the stores: loads and stores share one
            // in-order counter, so a load issued behind a store waits for the store's round trip as well
            float sc[16], sh[16], rv[16];
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const int o = ob64 * kC0Out + ob * 32 + (r & 3) + 8 * (r >> 2) + 4 * kk;
                const size_t idx = ((size_t)n * Cout + o) * vol + (size_t)d * plane + (size_t)h * W + w;
                const bool fin = nsplit == 1;
                sc[r] = (fin && scale) ? scale[o] : 1.0f;
                sh[r] = (fin && scale) ? shift[o] : 0.0f;
                rv[r] = (fin && residual) ? residual[idx] : 0.0f;
            }
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const int o = ob64 * kC0Out + ob * 32 + (r & 3) + 8 * (r >> 2) + 4 * kk;
                float v = acc[ob][rb][r];
                const size_t idx = ((size_t)n * Cout + o) * vol + (size_t)d * plane + (size_t)h * W + w;
                if (nsplit > 1) {
                    partial[(size_t)split * partial_stride + idx] = v;
                    continue;
                }
                if (scale) v = fmaf(v, sc[r], sh[r]);
                if (residual) v = v + rv[r];   // ResModule: x = act(bn(conv(h)) + identity), imvoxel_neck.py:227-229
                if (relu) v = fmaxf(v, 0.0f);
                out[idx] = v;
            }
        }
    }
}

// out = [relu]([scale * ] sum over splits (ascending) of partial [+ shift] [+ residual]): the epilogue of the split form
__global__ __launch_bounds__(kThreads) void splitk_epilogue_kernel(const float* __restrict__ partial, int nsplit, size_t total,
                                                                   const float* __restrict__ scale, const float* __restrict__ shift,
                                                                   const float* __restrict__ residual, float* __restrict__ out,
                                                                   int Cout, size_t vol, int relu) {
    const size_t i = (size_t)blockIdx.x * kThreads + threadIdx.x;
    if (i >= total) return;
    float v = partial[i];
    for (int s = 1; s < nsplit; ++s) v += partial[(size_t)s * total + i];
    const int o = (int)((i / vol) % Cout);
    if (scale) v = fmaf(v, scale[o], shift[o]);
    if (residual) v = v + residual[i];
    if (relu) v = fmaxf(v, 0.0f);
    out[i] = v;
}

// host-side launch for the other translation units (costreg_bf16.hip)
void launch_splitk_epilogue(const float* partial, int nsplit, size_t total, const float* scale, const float* shift,
                            const float* residual, float* out, int Cout, size_t vol, int relu, hipStream_t st) {
    hipLaunchKernelGGL(splitk_epilogue_kernel, dim3((unsigned)((total + kThreads - 1) / kThreads)), dim3(kThreads), 0, st, partial,
                       nsplit, total, scale, shift, residual, out, Cout, vol, relu);
}

}  // namespace mvsdet

using namespace mvsdet;

namespace {
struct ConvPlan {
    int D, H, W, twc, td, tiles_w, tiles_h, tiles_d, nsplit;
};

// tile shape (TWC along w, TD planes, 256/(TD*TWC) rows): the one that pads (D, H, W) least (conv_plan); the 4-plane tile with 16-wide groups only exists
// for stride 1 and 16-wide column groups.  nsplit: splits of the channel pairs so that a small grid reaches ~2 blocks
// per CU, each split at least 8 pairs long.
ConvPlan conv_plan(int N, int Cin, int Cout, int Di, int Hi, int Wi, int stride) {
    ConvPlan p;
    // output extent of kernel 3, padding 1: floor((in - 1) / stride) + 1
    p.D = (Di - 1) / stride + 1; p.H = (Hi - 1) / stride + 1; p.W = (Wi - 1) / stride + 1;
    auto padded = [&](int twc_, int td_) {
        const int th_ = 256 / (td_ * twc_);
        return (long long)((p.W + twc_ - 1) / twc_ * twc_) * ((p.H + th_ - 1) / th_ * th_) * ((p.D + td_ - 1) / td_ * td_);
    };
    // candidates in order of preference at equal padding: wide column groups first (fewer, longer LDS rows).  The 8- and
    // 4-wide groups (and, at stride 2, their 4-plane forms: a 9 x 17 x 17 halo is no larger than 5 x 33 x 17) serve the small
    // grids (the neck's 20x20x8 and 10x10x4, the cost network's 3x15x20 quarter resolution)
    static const int cand1[][2] = {{32, 2}, {16, 2}, {16, 4}, {8, 2}, {8, 4}, {4, 4}};
    static const int cand2[][2] = {{32, 2}, {16, 2}, {8, 2}, {8, 4}, {4, 4}};
    const int (*cand)[2] = stride == 1 ? cand1 : cand2;
    const int ncand = stride == 1 ? 6 : 5;
    p.twc = cand[0][0]; p.td = cand[0][1];
    long long best = padded(p.twc, p.td);
    for (int i = 1; i < ncand; ++i)
        if (padded(cand[i][0], cand[i][1]) < best) { best = padded(cand[i][0], cand[i][1]); p.twc = cand[i][0]; p.td = cand[i][1]; }
    const int th = 256 / (p.td * p.twc);
    p.tiles_w = (p.W + p.twc - 1) / p.twc; p.tiles_h = (p.H + th - 1) / th; p.tiles_d = (p.D + p.td - 1) / p.td;
    const long long blocks = (long long)p.tiles_w * p.tiles_h * p.tiles_d * N * (Cout / kC0Out);
    const int npairs = (Cin + 1) / 2;
    p.nsplit = 1;
    if (blocks < 384) p.nsplit = (int)std::max(1LL, std::min<long long>((512 + blocks - 1) / blocks, npairs / 8));
    else if (blocks < 2048) p.nsplit = (int)std::max(1LL, std::min<long long>((2048 + blocks - 1) / blocks, npairs / 8));   // under 4 rounds of 2 blocks per CU: even out the CUs
    return p;
}
}  // namespace

// Bytes of workspace the convolution wants for its partial sums at this shape (0: the grid is large enough unsplit).
extern "C" size_t mvsdet_conv3d_k3_mfma_workspace_bytes(int N, int Cin, int Cout, int D, int H, int W, int stride) {
    if (N <= 0 || Cin <= 0 || Cout <= 0 || Cout % kC0Out || D <= 0 || H <= 0 || W <= 0 || (stride != 1 && stride != 2)) return 0;
    const ConvPlan p = conv_plan(N, Cin, Cout, D, H, W, stride);
    return p.nsplit > 1 ? (size_t)p.nsplit * N * Cout * p.D * p.H * p.W * sizeof(float) : 0;
}

static int launch_conv_mfma(const char* name, const float* x, const float* weight_perm, const float* scale,
                            const float* shift, float* out, int N, int Cin, int Cout, int Di, int Hi, int Wi, int stride,
                            int relu, mvsdet_stream_t stream, const float* residual = nullptr, float* workspace = nullptr,
                            size_t workspace_bytes = 0) {
    MVS_REQUIRE(x && weight_perm && out, "%s: NULL pointer", name);
    MVS_REQUIRE((scale == nullptr) == (shift == nullptr), "%s: scale and shift come together", name);
    MVS_REQUIRE(N > 0 && Cin > 0 && Di > 0 && Hi > 0 && Wi > 0, "%s: bad shape N=%d Cin=%d D=%d H=%d W=%d", name, N, Cin, Di, Hi,
                Wi);
    MVS_REQUIRE(stride == 1 || stride == 2, "%s: stride %d not in {1,2}", name, stride);
    MVS_REQUIRE(((uintptr_t)weight_perm & 15u) == 0, "%s: weights must be 16-byte aligned", name);
    MVS_REQUIRE((size_t)Di * Hi * Wi < ((size_t)1 << 30), "%s: one channel volume exceeds 2^30 elements", name);
    MVS_REQUIRE(Cout > 0 && Cout % kC0Out == 0, "%s: Cout=%d must be a multiple of 64", name, Cout);
    ConvPlan p = conv_plan(N, Cin, Cout, Di, Hi, Wi, stride);
    const int D = p.D, H = p.H, W = p.W, twc = p.twc, td = p.td, tiles_w = p.tiles_w, tiles_h = p.tiles_h, tiles_d = p.tiles_d;
    const size_t total = (size_t)N * Cout * D * H * W;
    // without (enough) workspace the convolution runs unsplit
    const int nsplit = (workspace && workspace_bytes >= (size_t)p.nsplit * total * sizeof(float)) ? p.nsplit : 1;
    MVS_REQUIRE((long long)N * (Cout / kC0Out) * nsplit <= 65535 && tiles_d <= 65535, "%s: N*Cout/64 or D too large", name);
    dim3 grid((unsigned)(tiles_w * tiles_h), (unsigned)tiles_d, (unsigned)(N * (Cout / kC0Out) * nsplit));
    const float4* w4 = reinterpret_cast<const float4*>(weight_perm);
    hipStream_t st = (hipStream_t)stream;
#define MVS_CONV_CASE(TW_, ST_, TD_)                                                                                          \
    hipLaunchKernelGGL((conv3d_k3_mfma_kernel<TW_, ST_, TD_>), grid, dim3(kThreads), 0, st, x, w4, residual, scale, shift, out, Cin, \
                       Cout, Di, Hi, Wi, D, H, W, tiles_w, tiles_h, relu, nsplit, workspace, total)
    if (stride == 1) {
        if (twc == 4) MVS_CONV_CASE(4, 1, 4);
        else if (twc == 8 && td == 2) MVS_CONV_CASE(8, 1, 2);
        else if (twc == 8) MVS_CONV_CASE(8, 1, 4);
        else if (td == 4) MVS_CONV_CASE(16, 1, 4);
        else if (twc == 16) MVS_CONV_CASE(16, 1, 2);
        else MVS_CONV_CASE(32, 1, 2);
    } else {
        if (twc == 4) MVS_CONV_CASE(4, 2, 4);
        else if (twc == 8 && td == 4) MVS_CONV_CASE(8, 2, 4);
        else if (twc == 8) MVS_CONV_CASE(8, 2, 2);
        else if (twc == 16) MVS_CONV_CASE(16, 2, 2);
        else MVS_CONV_CASE(32, 2, 2);
    }
#undef MVS_CONV_CASE
    MVS_LAUNCH_CHECK(name);
    if (nsplit > 1) {
        hipLaunchKernelGGL(splitk_epilogue_kernel, dim3((unsigned)((total + kThreads - 1) / kThreads)), dim3(kThreads), 0, st,
                           workspace, nsplit, total, scale, shift, residual, out, Cout, (size_t)D * H * W, relu);
        MVS_LAUNCH_CHECK(name);
    }
    return MVSDET_OK;
}

// The general form: stride 1 or 2, optional affine, residual (added between the affine and the ReLU; stride 1 only) and
// workspace (mvsdet_conv3d_k3_mfma_workspace_bytes; NULL or too small: unsplit).
extern "C" int mvsdet_conv3d_k3_mfma_ws_f32(const float* x, const float* weight_perm, const float* scale, const float* shift,
                                            const float* residual, float* out, void* workspace, size_t workspace_bytes, int N,
                                            int Cin, int Cout, int D, int H, int W, int stride, int relu,
                                            mvsdet_stream_t stream) {
    MVS_REQUIRE(residual == nullptr || stride == 1, "conv3d_k3_mfma_ws: a residual needs stride 1");
    return launch_conv_mfma("conv3d_k3_mfma_ws", x, weight_perm, scale, shift, out, N, Cin, Cout, D, H, W, stride, relu, stream,
                            residual, (float*)workspace, workspace_bytes);
}

extern "C" int mvsdet_conv3d_k3_mfma_f32(const float* x, const float* weight_perm, const float* scale, const float* shift,
                                         float* out, int N, int Cin, int Cout, int D, int H, int W, int relu,
                                         mvsdet_stream_t stream) {
    return launch_conv_mfma("conv3d_k3_mfma", x, weight_perm, scale, shift, out, N, Cin, Cout, D, H, W, 1, relu, stream);
}

extern "C" int mvsdet_conv3d_k3_res_mfma_f32(const float* x, const float* weight_perm, const float* scale, const float* shift,
                                             const float* residual, float* out, int N, int Cin, int Cout, int D, int H, int W,
                                             int relu, mvsdet_stream_t stream) {
    return launch_conv_mfma("conv3d_k3_res_mfma", x, weight_perm, scale, shift, out, N, Cin, Cout, D, H, W, 1, relu, stream, residual);
}

extern "C" int mvsdet_conv3d_k3_s2_mfma_f32(const float* x, const float* weight_perm, const float* scale,
                                            const float* shift, float* out, int N, int Cin, int Cout, int D, int H, int W,
                                            int relu, mvsdet_stream_t stream) {
    return launch_conv_mfma("conv3d_k3_s2_mfma", x, weight_perm, scale, shift, out, N, Cin, Cout, D, H, W, 2, relu, stream);
}

// ---------------------------------------------------------------------------------------------------------------
// The two up-sampling layers (mvsnet.py:92-100,110-111): ConvTranspose3d(kernel 3, stride 2, padding 1,
// output_padding 1, no bias) + BatchNorm + ReLU, then the skip connection is added.  out[2i + p] per dimension only
// sees tap k = 1 (input i) when p = 0 and taps k = 0 (input i + 1), k = 2 (input i) when p = 1, so each of the 8 output
// parity classes (PD,PH,PW) is a small stride-1 convolution over the INPUT grid with 1, 2, 4 or 8 taps: the same
// implicit GEMM as above, B rows contiguous in the input, results written to the strided output positions.
// weight_perm: the (Cin,Cout,3,3,3) ConvTranspose3d weight permuted to [c][kd][kh][kw][o].
// ---------------------------------------------------------------------------------------------------------------
namespace mvsdet {

// One kernel instance per (PD, PH) output parity in d and h; BOTH w parities are computed by the same block -- output
// 2i (tap kw = 1, input i) and output 2i + 1 (taps kw = 0, input i + 1, and kw = 2, input i) -- so that a lane writes
// the two neighbouring outputs as one float2 (32 lanes x 8 B = 256 contiguous bytes; reads the skip tensor the same way)
// instead of two launches writing every other float.  G = ND * NH tap groups over (d, h), each with the 3 kw taps; a
// stage carries P = 8 / G channel pairs = 96 MFMAs per wave behind one barrier pair (48 for the single-group class; the
// 8-launch form had 32: shorter than the latency of the next stage's global loads).
template <int TWC, int PD, int PH>
__global__ __launch_bounds__(kThreads, 2) void convT3d_k3_s2_mfma_kernel(
    const float* __restrict__ x, const float* __restrict__ wperm, const float* __restrict__ scale,
    const float* __restrict__ shift, const float* __restrict__ residual, float* __restrict__ out, int Cin, int Cout,
    int Di, int Hi, int Wi, int tiles_w, int tiles_h, int relu) {
    constexpr int kTW = TWC, kTH = 4 * (32 / TWC);
    constexpr int kHD = kC0D + 1, kHH = kTH + 1, kHW = kTW + 1;   // +1 on the high side: input i + 1 for tap 0
    constexpr int kHalo = kHD * kHH * kHW;                         // per channel
    constexpr int kRowsPerCol = 32 / TWC;
    constexpr int ND = PD ? 2 : 1, NH = PH ? 2 : 1, G = ND * NH, T = 3 * G;
    constexpr int P = G == 1 ? 4 : 8 / G;                          // channel pairs per stage (8 would spill at G = 1)
    constexpr int kInFloats = P * 2 * kHalo;                       // staged input values
    constexpr int kInStage = (kInFloats + kThreads - 1) / kThreads;
    constexpr int kWFloats = P * 2 * T * kC0Out;                   // staged weights: [pair][k][group][kw][64] = 3072 (1536)
    constexpr int kWStage = (kWFloats + kThreads - 1) / kThreads;  // 12
    __shared__ float s_in[kInFloats];
    __shared__ float s_w[kWFloats];

    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int bw = blockIdx.x % tiles_w, bh = blockIdx.x / tiles_w;
    const int nob = Cout / kC0Out;
    const int w0 = bw * kTW, h0 = bh * kTH, d0 = blockIdx.y * kC0D, n = blockIdx.z / nob, ob64 = blockIdx.z % nob;
    const size_t iplane = (size_t)Hi * Wi, ivol = (size_t)Di * iplane;
    const int Do = 2 * Di, Ho = 2 * Hi, Wo = 2 * Wi;
    const size_t oplane = (size_t)Ho * Wo, ovol = (size_t)Do * oplane;
    const float* xn = x + (size_t)n * Cin * ivol;
    const int npairs = (Cin + 1) / 2;
    const int nstages = (npairs + P - 1) / P;

    const int dz0 = wave >> 1, hy0 = (wave & 1) * 2 * kRowsPerCol;
    const int col = lane & 31, kk = lane >> 5;
    const int chy = col / TWC, cw = col % TWC;

    // staging plans.  Input element e -> (channel q of the stage's 2P, halo position); weight element e ->
    // (channel q = 2*pair + k, tap group, kw, output channel o)
    int in_off[kInStage];
#pragma unroll
    for (int k = 0; k < kInStage; ++k) {
        const int e = tid + k * kThreads;
        const int q = e / kHalo, r = e - q * kHalo;
        const int dz = r / (kHH * kHW), r2 = r - dz * (kHH * kHW);
        const int hy = r2 / kHW, wx = r2 - hy * kHW;
        const int d = d0 + dz, h = h0 + hy, w = w0 + wx;
        const bool ok = e < kInFloats && d < Di && h < Hi && w < Wi;
        in_off[k] = ok ? (int)((size_t)d * iplane + (size_t)h * Wi + w) : -1;   // channel q = e / kHalo recomputed below
    }
    float in_reg[kInStage], w_reg[kWStage];
    auto fetch = [&](int sidx) {
        const int c0 = sidx * 2 * P;  // first channel of the stage
#pragma unroll
        for (int k = 0; k < kInStage; ++k) {
            const int q = (tid + k * kThreads) / kHalo;
            float v = 0.0f;
            if (in_off[k] >= 0 && c0 + q < Cin) v = xn[(size_t)(c0 + q) * ivol + in_off[k]];
            in_reg[k] = v;
        }
#pragma unroll
        for (int k = 0; k < kWStage; ++k) {
            // offset inside wperm relative to the stage's first channel: (q * 27 + tap27) * Cout + o; recomputed per stage
            // (12 plan registers would not pay: the arithmetic is a handful of shifts)
            const int e = tid + k * kThreads;
            const int o = e % kC0Out, t = (e / kC0Out) % T, q = e / (kC0Out * T);
            const int kw = t % 3, gi = t / 3, th = gi % NH, td = gi / NH;
            const int kd = PD ? (td ? 2 : 0) : 1, kh = PH ? (th ? 2 : 0) : 1;
            const int off = (q * 27 + (kd * 3 + kh) * 3 + kw) * Cout + ob64 * kC0Out + o;
            // wperm is zero padded to an even channel count by the caller; beyond that, zero here
            w_reg[k] = (e < kWFloats && c0 + q < ((Cin + 1) & ~1)) ? wperm[(size_t)c0 * 27 * Cout + off] : 0.0f;
        }
    };

    f32x16 acc[2][2][2];   // [w parity][output-channel block][voxel row]
#pragma unroll
    for (int pw = 0; pw < 2; ++pw)
#pragma unroll
        for (int a = 0; a < 2; ++a)
#pragma unroll
            for (int b = 0; b < 2; ++b)
#pragma unroll
                for (int r = 0; r < 16; ++r) acc[pw][a][b][r] = 0.0f;

    fetch(0);
    for (int sidx = 0; sidx < nstages; ++sidx) {
        __syncthreads();
#pragma unroll
        for (int k = 0; k < kInStage; ++k) {
            const int e = tid + k * kThreads;
            if (e < kInFloats) s_in[e] = in_reg[k];
        }
#pragma unroll
        for (int k = 0; k < kWStage; ++k) {
            const int e = tid + k * kThreads;
            if (e < kWFloats) s_w[e] = w_reg[k];
        }
        __syncthreads();
        if (sidx + 1 < nstages) fetch(sidx + 1);

#pragma unroll
        for (int p = 0; p < P; ++p) {
            const float* bin = s_in + (2 * p + kk) * kHalo + (dz0 * kHH + hy0 + chy) * kHW + cw;
            const float* ain = s_w + (2 * p + kk) * T * kC0Out + col;
            // per dimension: parity 0 sees tap 1 at input offset 0; parity 1 sees tap 0 at offset +1 and tap 2 at offset 0
#pragma unroll
            for (int td = 0; td < ND; ++td)
#pragma unroll
                for (int th = 0; th < NH; ++th) {
                    const int od = PD ? (td ? 0 : 1) : 0, oh = PH ? (th ? 0 : 1) : 0;
                    const float* ag = ain + (td * NH + th) * 3 * kC0Out;
                    const float a00 = ag[0], a01 = ag[32];                             // kw = 0
                    const float a10 = ag[kC0Out], a11 = ag[kC0Out + 32];               // kw = 1
                    const float a20 = ag[2 * kC0Out], a21 = ag[2 * kC0Out + 32];       // kw = 2
                    const float* bg = bin + (od * kHH + oh) * kHW;
                    const float b00 = bg[0], b01 = bg[1];                              // row 0: input i, i + 1
                    const float b10 = bg[kRowsPerCol * kHW], b11 = bg[kRowsPerCol * kHW + 1];
                    // even outputs: kw = 1 on input i
                    acc[0][0][0] = __builtin_amdgcn_mfma_f32_32x32x2f32(a10, b00, acc[0][0][0], 0, 0, 0);
                    acc[0][0][1] = __builtin_amdgcn_mfma_f32_32x32x2f32(a10, b10, acc[0][0][1], 0, 0, 0);
                    acc[0][1][0] = __builtin_amdgcn_mfma_f32_32x32x2f32(a11, b00, acc[0][1][0], 0, 0, 0);
                    acc[0][1][1] = __builtin_amdgcn_mfma_f32_32x32x2f32(a11, b10, acc[0][1][1], 0, 0, 0);
                    // odd outputs: kw = 0 on input i + 1, then kw = 2 on input i
                    acc[1][0][0] = __builtin_amdgcn_mfma_f32_32x32x2f32(a00, b01, acc[1][0][0], 0, 0, 0);
                    acc[1][0][1] = __builtin_amdgcn_mfma_f32_32x32x2f32(a00, b11, acc[1][0][1], 0, 0, 0);
                    acc[1][1][0] = __builtin_amdgcn_mfma_f32_32x32x2f32(a01, b01, acc[1][1][0], 0, 0, 0);
                    acc[1][1][1] = __builtin_amdgcn_mfma_f32_32x32x2f32(a01, b11, acc[1][1][1], 0, 0, 0);
                    acc[1][0][0] = __builtin_amdgcn_mfma_f32_32x32x2f32(a20, b00, acc[1][0][0], 0, 0, 0);
                    acc[1][0][1] = __builtin_amdgcn_mfma_f32_32x32x2f32(a20, b10, acc[1][0][1], 0, 0, 0);
                    acc[1][1][0] = __builtin_amdgcn_mfma_f32_32x32x2f32(a21, b00, acc[1][1][0], 0, 0, 0);
                    acc[1][1][1] = __builtin_amdgcn_mfma_f32_32x32x2f32(a21, b10, acc[1][1][1], 0, 0, 0);
                }
        }
    }

    const int wi = w0 + cw;
#pragma unroll
    for (int rb = 0; rb < 2; ++rb) {
        const int di = d0 + dz0, hi = h0 + hy0 + rb * kRowsPerCol + chy;
        if (di >= Di || hi >= Hi || wi >= Wi) continue;
        const size_t pos = (size_t)(2 * di + PD) * oplane + (size_t)(2 * hi + PH) * Wo + 2 * wi;   // even: 8-byte aligned
#pragma unroll
        for (int ob = 0; ob < 2; ++ob) {
            float sc[16], sh[16];   // requested ahead of the stores (see conv3d_k3_mfma_kernel)
            float2 rv[16];
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const int o = ob64 * kC0Out + ob * 32 + (r & 3) + 8 * (r >> 2) + 4 * kk;
                sc[r] = scale ? scale[o] : 1.0f;
                sh[r] = scale ? shift[o] : 0.0f;
                rv[r] = residual ? *reinterpret_cast<const float2*>(residual + ((size_t)n * Cout + o) * ovol + pos) : make_float2(0.f, 0.f);
            }
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const int o = ob64 * kC0Out + ob * 32 + (r & 3) + 8 * (r >> 2) + 4 * kk;
                float2 v = make_float2(acc[0][ob][rb][r], acc[1][ob][rb][r]);
                if (scale) {
                    v.x = fmaf(v.x, sc[r], sh[r]);
                    v.y = fmaf(v.y, sc[r], sh[r]);
                }
                if (relu) {
                    v.x = fmaxf(v.x, 0.0f);
                    v.y = fmaxf(v.y, 0.0f);
                }
                if (residual) {
                    v.x = rv[r].x + v.x;
                    v.y = rv[r].y + v.y;
                }
                *reinterpret_cast<float2*>(out + ((size_t)n * Cout + o) * ovol + pos) = v;
            }
        }
    }
}

}  // namespace mvsdet

extern "C" int mvsdet_convT3d_k3_s2_mfma_f32(const float* x, const float* weight_perm, const float* scale, const float* shift,
                                             const float* residual, float* out, int N, int Cin, int Cout, int D, int H, int W,
                                             int relu, mvsdet_stream_t stream) {
    const char* name = "convT3d_k3_s2_mfma";
    MVS_REQUIRE(x && weight_perm && out, "%s: NULL pointer", name);
    MVS_REQUIRE((scale == nullptr) == (shift == nullptr), "%s: scale and shift come together", name);
    MVS_REQUIRE(N > 0 && Cin > 0 && D > 0 && H > 0 && W > 0, "%s: bad shape N=%d Cin=%d D=%d H=%d W=%d", name, N, Cin, D, H, W);
    MVS_REQUIRE(((uintptr_t)weight_perm & 15u) == 0, "%s: weights must be 16-byte aligned", name);
    MVS_REQUIRE((size_t)D * H * W < ((size_t)1 << 27), "%s: one input channel volume exceeds 2^27 elements", name);
    MVS_REQUIRE(Cout > 0 && Cout % kC0Out == 0, "%s: Cout=%d must be a multiple of 64", name, Cout);
    // tile = 2 planes x 4*(32/twc) rows x twc columns of the INPUT grid: the column-group width that pads (H, W) least
    int twc = 32;
    long long best = -1;
    for (int cand : {32, 16, 8}) {
        const int th_ = 4 * (32 / cand);
        const long long padded = (long long)((W + cand - 1) / cand * cand) * ((H + th_ - 1) / th_ * th_);
        if (best < 0 || padded < best) { best = padded; twc = cand; }
    }
    const int th = 4 * (32 / twc);
    const int tiles_w = (W + twc - 1) / twc, tiles_h = (H + th - 1) / th, tiles_d = (D + kC0D - 1) / kC0D;
    MVS_REQUIRE((long long)N * (Cout / kC0Out) <= 65535 && tiles_d <= 65535, "%s: N*Cout/64 or D too large", name);
    dim3 grid((unsigned)(tiles_w * tiles_h), (unsigned)tiles_d, (unsigned)(N * (Cout / kC0Out)));
    hipStream_t st = (hipStream_t)stream;
    MVS_REQUIRE(((uintptr_t)out & 7u) == 0 && (residual == nullptr || ((uintptr_t)residual & 7u) == 0),
                "%s: out and residual must be 8-byte aligned", name);
#define MVS_CT_CASE(TW_, PD_, PH_)                                                                                          \
    hipLaunchKernelGGL((convT3d_k3_s2_mfma_kernel<TW_, PD_, PH_>), grid, dim3(kThreads), 0, st, x, weight_perm, scale, shift,  \
                       residual, out, Cin, Cout, D, H, W, tiles_w, tiles_h, relu)
#define MVS_CT_ALL(TW_) MVS_CT_CASE(TW_, 0, 0); MVS_CT_CASE(TW_, 0, 1); MVS_CT_CASE(TW_, 1, 0); MVS_CT_CASE(TW_, 1, 1)
    if (twc == 8) { MVS_CT_ALL(8); } else if (twc == 16) { MVS_CT_ALL(16); } else { MVS_CT_ALL(32); }
#undef MVS_CT_ALL
#undef MVS_CT_CASE
    MVS_LAUNCH_CHECK(name);
    return MVSDET_OK;
}
