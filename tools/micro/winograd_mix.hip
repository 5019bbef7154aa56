// Gate (b) of VERDICT r4 task 1: what would a transform-domain conv0 reach on this chip?
//
// conv0 of CostRegNet_3DGS (mvs_models/mvsnet.py:76: 256 -> 64 channels, 3x3x3, on 40 x 12 x 60 x 80 voxels) as Winograd
// F(2x2, 3x3) over (H, W) with the depth taps direct needs 12 instead of 27 products per output: 2.72 TFLOP of bf16 MFMA per
// scene instead of 6.1 (three split terms each).  The gate on the numbers passed (tools/study/winograd_bf16x3_emulation.py).
// This file measures the OTHER side: the operand traffic.  Sixteen transform points are sixteen independent GEMMs that share
// nothing, so a block that owns 64 output channels x 48 tile-depths x 16 points (49 K accumulators: what eight waves hold)
// streams ALL 3 MiB of transformed weights (16 points x 3 depth taps x 64 x 256 x two bf16 pieces) per 48 columns -- 12x the
// operand bytes per MAC of the direct kernel, whose 27 taps share one halo tile.
//
// The kernel below has the real kernel's resource mix and data flow, block for block, but not its indexing details:
//   * grid = 12 000 blocks = 40 views x 4 groups of 3 output depths x 75 groups of 2 x 8 tiles (4 x 16 output pixels);
//     8 waves, wave w owns transform points (i, j) = (w >> 1, 2 (w & 1)) and (i, j + 1) for all 64 channels x 48 columns
//   * per 32-channel step: the raw fp32 halo box (5 depths x 6 x 18 pixels x 32 channels = 69 KB) goes from the REAL tensor
//     (NCDHW, 2.36 GB) into LDS as 16-byte units of 4 channels, column parities apart (conflict-free b128 reads per tile run);
//     every wave builds its B fragments from 6 raw pixels per (depth, tile): row combination, column combination (B^T d B),
//     cut into bf16 hi / mid AFTER the transform (what the emulation models), 5 depths x 2 points;
//     its A fragments (2 points x 3 depth taps x 4 row groups x 2 pieces x 1 KiB) come straight from the L2-resident weight
//     buffer in fragment order; 216 v_mfma_f32_16x16x32_bf16 per wave and step (3 terms), fp32 accumulators
//   * epilogue: the accumulators of a wave's two points are combined and stored (the real kernel exchanges the four column
//     partners through LDS for A^T M A: same bytes).
// Weights are random, the column combination uses fixed signs: the RESULT is not a convolution; time, counters and the
// instruction stream are those of one.
//
// Build + run on the GPU box:  hipcc -O3 --offload-arch=gfx950 -o tools/micro/winograd_mix tools/micro/winograd_mix.hip && tools/micro/winograd_mix
#include <hip/hip_runtime.h>

#include <cstdio>
#include <cstdlib>
#include <vector>

typedef short bf16x8 __attribute__((ext_vector_type(8)));
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef float f32x2 __attribute__((ext_vector_type(2)));

constexpr int N = 40, C = 256, D = 12, H = 60, W = 80, CO = 64;
constexpr int KS = 32;                       // channels per step
constexpr int BD = 5, BH = 6, BW = 18;       // raw box: depths, rows, columns
constexpr int UNITS = BD * BH * BW;          // pixels per box (540)
constexpr int STAGE_F4 = UNITS * (KS / 4);   // 16-byte units per stage (4320)
constexpr int TPB = 512;

// LDS unit index of (c4 group, depth e, row r, column c): column parities apart so that consecutive tiles are consecutive units
__device__ __forceinline__ int unit_of(int c4, int e, int r, int c) {
    return ((c4 * BD + e) * BH + r) * BW + (c & 1) * (BW / 2) + (c >> 1);
}

__device__ __forceinline__ unsigned pack_bf16(float a, float b) {
    typedef __bf16 bf2 __attribute__((ext_vector_type(2)));
    const bf2 v = {(__bf16)a, (__bf16)b};
    return __builtin_bit_cast(unsigned, v);
}

// 8 fp32 -> bf16x8 hi and bf16x8 mid (round to nearest even of the value and of the exact remainder)
__device__ __forceinline__ void cut8(const float (&v)[8], bf16x8& hi, bf16x8& mid) {
    unsigned h[4], m[4];
#pragma unroll
    for (int p = 0; p < 4; ++p) {
        h[p] = pack_bf16(v[2 * p], v[2 * p + 1]);
        const float r0 = v[2 * p] - __uint_as_float(h[p] << 16);
        const float r1 = v[2 * p + 1] - __uint_as_float(h[p] & 0xffff0000u);
        m[p] = pack_bf16(r0, r1);
    }
    typedef unsigned u4 __attribute__((ext_vector_type(4)));
    hi = __builtin_bit_cast(bf16x8, (u4){h[0], h[1], h[2], h[3]});
    mid = __builtin_bit_cast(bf16x8, (u4){m[0], m[1], m[2], m[3]});
}

__global__ __launch_bounds__(TPB, 1) void winograd_mix_kernel(const float* __restrict__ x, const uint4* __restrict__ u,
                                                              float* __restrict__ out, int variant) {
    extern __shared__ float4 s_raw[];   // 2 stages x STAGE_F4
    const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    // an XCD takes a contiguous eighth of the grid, depth groups of one tile group next to each other
    const int nb = gridDim.x;
    const int b = (blockIdx.x & 7) * (nb >> 3) + (blockIdx.x >> 3);
    const int odg = b & 3, tg = (b >> 2) % 75, n = b / 300;
    const int trp = tg / 5, tcg = tg % 5;
    const int d0 = odg * 3 - 1, h0 = trp * 4 - 1, w0 = tcg * 16 - 1;
    const size_t plane = (size_t)D * H * W;
    const float* xn = x + (size_t)n * C * plane;

    // the wave's points: rows of B^T for i, column pairs for j, j + 1 (fixed signs)
    const int pi = wave >> 1, pj = 2 * (wave & 1);
    const int ra = pi == 0 ? 0 : 1, rb = pi == 3 ? 3 : 2;
    const int tile = lane & 15, kg = lane >> 4;           // B operand: column = tile, k = 8 kg .. 8 kg + 7
    const int trow = tile >> 3, tcol = tile & 7;

    f32x4 acc[2][3][4];
#pragma unroll
    for (int p = 0; p < 2; ++p)
#pragma unroll
        for (int o = 0; o < 3; ++o)
#pragma unroll
            for (int g = 0; g < 4; ++g) acc[p][o][g] = (f32x4){0.f, 0.f, 0.f, 0.f};

    auto stage = [&](int ks, int buf) {
        // 540 pixels x 8 groups of 4 channels: a thread gathers the 4 channel planes of its pixel (coalesced along w per plane)
        for (int q = tid; q < STAGE_F4; q += TPB) {
            const int c4 = q / UNITS, p = q - c4 * UNITS;
            const int e = p / (BH * BW), r = (p / BW) % BH, c = p % BW;
            const int dd = min(max(d0 + e, 0), D - 1), hh = min(max(h0 + r, 0), H - 1), ww = min(max(w0 + c, 0), W - 1);
            const float* src = xn + (size_t)(ks * KS + c4 * 4) * plane + ((size_t)dd * H + hh) * W + ww;
            float4 v;
            v.x = src[0]; v.y = src[plane]; v.z = src[2 * plane]; v.w = src[3 * plane];
            s_raw[buf * STAGE_F4 + unit_of(c4, e, r, c)] = v;
        }
    };

    stage(0, 0);
    __syncthreads();
    for (int ks = 0; ks < C / KS; ++ks) {
        const int buf = ks & 1;
        if (ks + 1 < C / KS && variant != 2) stage(ks + 1, buf ^ 1);
        const float4* raw = s_raw + buf * STAGE_F4;
        // weights of this step: [ks][point 16][kd 3][rg 4][piece 2][64 lanes] uint4
        const uint4* us = u + ((size_t)ks * 16 + (pi * 4 + pj)) * 3 * 4 * 2 * 64 + lane;
#pragma unroll
        for (int p = 0; p < 2; ++p) {
            // ---- B fragments of point (pi, pj + p) for the 5 depths: B^T d B on 8 channels of the lane's tile, then the cut
            bf16x8 Bh[BD], Bm[BD];
            const int ca = (pj + p) == 0 ? 0 : 1, cb = (pj + p) == 3 ? 3 : 2;
            const float sgn_r = (pi == 1) ? 1.f : -1.f, sgn_c = ((pj + p) == 1) ? 1.f : -1.f;
#pragma unroll
            for (int e = 0; e < BD; ++e) {
                float v[8];
#pragma unroll
                for (int half = 0; half < 2; ++half) {
                    const int c4 = 2 * kg + half;
                    const float4 aa = raw[unit_of(c4, e, 2 * trow + ra, 2 * tcol + ca)], ab = raw[unit_of(c4, e, 2 * trow + ra, 2 * tcol + cb)];
                    const float4 ba = raw[unit_of(c4, e, 2 * trow + rb, 2 * tcol + ca)], bb = raw[unit_of(c4, e, 2 * trow + rb, 2 * tcol + cb)];
                    const float4 ta = {fmaf(sgn_r, ba.x, aa.x), fmaf(sgn_r, ba.y, aa.y), fmaf(sgn_r, ba.z, aa.z), fmaf(sgn_r, ba.w, aa.w)};
                    const float4 tb = {fmaf(sgn_r, bb.x, ab.x), fmaf(sgn_r, bb.y, ab.y), fmaf(sgn_r, bb.z, ab.z), fmaf(sgn_r, bb.w, ab.w)};
                    v[4 * half + 0] = fmaf(sgn_c, tb.x, ta.x); v[4 * half + 1] = fmaf(sgn_c, tb.y, ta.y);
                    v[4 * half + 2] = fmaf(sgn_c, tb.z, ta.z); v[4 * half + 3] = fmaf(sgn_c, tb.w, ta.w);
                }
                cut8(v, Bh[e], Bm[e]);
            }
            if (variant == 3) continue;   // transform only
#pragma unroll
            for (int kd = 0; kd < 3; ++kd) {
                bf16x8 Ah[4], Am[4];
#pragma unroll
                for (int g = 0; g < 4; ++g) {
                    const uint4 a = us[(((size_t)p * 3 + kd) * 4 + g) * 2 * 64], m = us[((((size_t)p * 3 + kd) * 4 + g) * 2 + 1) * 64];
                    Ah[g] = __builtin_bit_cast(bf16x8, a);
                    Am[g] = __builtin_bit_cast(bf16x8, m);
                }
                if (variant == 4) {   // loads only: keep them alive
#pragma unroll
                    for (int g = 0; g < 4; ++g) acc[p][0][g][0] += (float)(Ah[g][0] + Am[g][1]);
                    continue;
                }
#pragma unroll
                for (int o = 0; o < 3; ++o)
#pragma unroll
                    for (int g = 0; g < 4; ++g) {
                        acc[p][o][g] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(Am[g], Bh[o + kd], acc[p][o][g], 0, 0, 0);
                        acc[p][o][g] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(Ah[g], Bm[o + kd], acc[p][o][g], 0, 0, 0);
                        acc[p][o][g] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(Ah[g], Bh[o + kd], acc[p][o][g], 0, 0, 0);
                    }
            }
        }
        __syncthreads();
    }
    // epilogue: 64 channels x 48 columns x 4 outputs per block; a wave stores the combination of its two points for its share
    float* ob = out + (size_t)b * (CO * 48 * 4);
#pragma unroll
    for (int o = 0; o < 3; ++o)
#pragma unroll
        for (int g = 0; g < 4; ++g) {
            const f32x4 s = acc[0][o][g] + acc[1][o][g];
            // wave w holds 1/8 of the transform domain: it writes 1/8 of the block's outputs (4 of the 16 x 2 halves)
            if ((g * 3 + o) % 2 == (wave & 1))
                *reinterpret_cast<f32x4*>(ob + ((((size_t)(wave >> 1) * 12 + (g * 3 + o)) * 64 + lane) * 4) % (CO * 48 * 4)) = s;
        }
}

int main(int argc, char** argv) {
    const size_t xe = (size_t)N * C * D * H * W, ue = (size_t)(C / KS) * 16 * 3 * 4 * 2 * 64, oe = (size_t)12000 * CO * 48 * 4;
    float *x, *out;
    uint4* u;
    hipMalloc(&x, xe * 4);
    hipMalloc(&u, ue * 16);
    hipMalloc(&out, oe * 4);
    {
        std::vector<float> hx(1 << 24);
        for (size_t i = 0; i < hx.size(); ++i) hx[i] = (float)((i * 2654435761u) >> 8 & 0xffff) / 65536.f;
        for (size_t off = 0; off < xe; off += hx.size()) hipMemcpy(x + off, hx.data(), std::min(hx.size(), xe - off) * 4, hipMemcpyHostToDevice);
        std::vector<unsigned> hu(ue * 4);
        for (size_t i = 0; i < hu.size(); ++i) hu[i] = 0x3c003c00u + (unsigned)((i * 40503u) & 0x007f007fu);
        hipMemcpy(u, hu.data(), ue * 16, hipMemcpyHostToDevice);
    }
    const size_t lds = 2 * (size_t)STAGE_F4 * 16;
    hipFuncSetAttribute(reinterpret_cast<const void*>(winograd_mix_kernel), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
    printf("weights %.2f MiB, LDS %.1f KiB per block, grid 12000 x %d threads\n", ue * 16 / 1048576.0, lds / 1024.0, TPB);
    const char* names[] = {"full mix (stage + transform + weight loads + 216 MFMA per wave and step)", "same (second run)",
                           "no staging of the next box (LDS contents reused)", "staging + transform only (no weights, no MFMA)",
                           "staging + transform + weight loads, no MFMA"};
    hipEvent_t e0, e1;
    hipEventCreate(&e0);
    hipEventCreate(&e1);
    for (int variant = 0; variant < 5; ++variant) {
        float best = 1e9f;
        for (int rep = 0; rep < 4; ++rep) {
            hipEventRecord(e0);
            hipLaunchKernelGGL(winograd_mix_kernel, dim3(12000), dim3(TPB), lds, 0, x, u, out, variant);
            hipEventRecord(e1);
            hipEventSynchronize(e1);
            float ms;
            hipEventElapsedTime(&ms, e0, e1);
            if (rep) best = std::min(best, ms);
        }
        if (hipGetLastError() != hipSuccess) { printf("launch failed\n"); return 1; }
        const double mfma_tflop = 2.0 * 12000 * 8 * 8 * 216 * 16 * 16 * 32 / 1e12;
        printf("variant %d: %.3f ms  -- %s", variant, best, names[variant]);
        if (variant < 3) printf("  [%.0f TFLOP/s of bf16 MFMA issued = %.2f of 2.5 PFLOP/s]", mfma_tflop / best * 1e3, mfma_tflop / best * 1e3 / 2500.0);
        printf("\n");
    }
    printf("for comparison: the shipped direct conv0 (conv3d_k3_bf16x3_kernel) 4.3-4.7 ms by box; target of the task 3.2 ms\n");
    return 0;
}
