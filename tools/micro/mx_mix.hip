// Gate (b) of VERDICT r5 next #2: what do the matrix pipes deliver on the MIXED-FORMAT product mix that would replace conv0's three
// bf16 MFMAs per fp32-equivalent product (tools/study/mixed_format_gate.py: the numerical gate passes at 1.0-1.3e-5 on the logits)?
//
//     bf16x3 (shipped):  x*w ~= xh*wh + xh*wm + xm*wh        3 x v_mfma_f32_16x16x32_bf16 per K = 32          12 per K = 128
//     fp16 + MX:         x*w ~= xh*wh (fp16) + Q(xh)*Q(wr) + Q(xr)*Q(wh)
//                        4 x v_mfma_f32_16x16x32_f16 + 2 x v_mfma_scale_f32_16x16x128_f8f6f4 per K = 128
//                        e2m3 operands: the scaled instruction takes the cycles of ONE 16x16x32 (MI355X_MICROARCH "MFMA" table) -> 6 units
//                        e4m3 operands: twice that                                                                             -> 8 units
//
// The paper ratio is 0.50 (FP6) / 0.67 (FP8) of the shipped mix.  What this file measures is the ratio THE CHIP holds: every CU
// issues the mix back to back from 12 waves (3 per SIMD, conv0's occupancy: one 160-KB block per CU) on random operands, eight
// independent 16 x 16 accumulator tiles per wave as in conv3d_k3_bf16x3_kernel's 16x16x32 form (2 row tiles x 4 column tiles),
// operands resident in registers.  No LDS, no global traffic inside the loop: this is the FLOOR of the matrix-pipe time; the
// clock the chip holds on each mix (DVFS) is part of the answer.
//
// Build + run on the GPU box:  hipcc -O3 --offload-arch=gfx950 -o tools/micro/mx_mix tools/micro/mx_mix.hip && tools/micro/mx_mix
#include <hip/hip_runtime.h>

#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <random>
#include <vector>

typedef short bf16x8 __attribute__((ext_vector_type(8)));
typedef _Float16 f16x8 __attribute__((ext_vector_type(8)));
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef int i32x8 __attribute__((ext_vector_type(8)));

constexpr int TPB = 768;      // 12 waves
constexpr int TILES = 8;      // accumulator tiles per wave

// conv0 at the reference-true shape: 40 x 12 x 60 x 80 voxels x 64 channels = 576 000 tiles of 16 x 16, K = 27 x 256 = 54 steps of 128
constexpr double CONV0_TILE_KSTEPS = 576000.0 * 54.0;

enum Mix { BF16X3 = 0, F16_FP6 = 1, F16_FP8 = 2, F16_ONLY = 3, FP6_ONLY = 4, BF16X1 = 5, F16_A6B8 = 6, F16_A8B6 = 7 };

template <int MIX>
__global__ __launch_bounds__(TPB, 1) void mx_mix_kernel(const uint4* __restrict__ ops, float* __restrict__ out, int steps) {
    const int tid = threadIdx.x;
    // operands: 16 x uint4 per lane, different per lane and per block (random bits made valid on the host)
    const uint4* mine = ops + ((size_t)(blockIdx.x & 63) * TPB + tid) * 16;
    uint4 raw[16];
#pragma unroll
    for (int i = 0; i < 16; ++i) raw[i] = mine[i];
    f32x4 acc[TILES];
#pragma unroll
    for (int t = 0; t < TILES; ++t) acc[t] = (f32x4){0.f, 0.f, 0.f, 0.f};
    // fragments of one K = 128 step: 4 k-steps of the 16-bit pieces for 2 row tiles (A) and 4 column tiles (B) would be 24 x 4
    // VGPRs; the real kernel re-reads them from LDS per k-step.  Here: 4 A and 4 B 16-bit fragments and 2 + 2 MX fragments,
    // combined differently per (tile, k-step) so that consecutive instructions see different operand bits.
    const int sa = 120 + (tid & 7), sb = 121 + ((tid >> 3) & 7);   // e8m0 scales near 2^-6
    // the MX fragments of a step (Q(xh), Q(xr) for the rows; Q(wr), Q(wh) for the columns), built ONCE: 8 VGPRs each in the builtin's
    // signature (an e2m3 fragment uses 6 of them)
    i32x8 mxa[2], mxb[2];
#pragma unroll
    for (int i = 0; i < 2; ++i) {
        const uint4 p = raw[4 + 2 * i], q = raw[5 + 2 * i], u = raw[12 + 2 * i], v = raw[13 + 2 * i];
        mxa[i] = (i32x8){(int)p.x, (int)p.y, (int)p.z, (int)p.w, (int)q.x, (int)q.y, (int)q.z, (int)q.w};
        mxb[i] = (i32x8){(int)u.x, (int)u.y, (int)u.z, (int)u.w, (int)v.x, (int)v.y, (int)v.z, (int)v.w};
    }
    for (int s = 0; s < steps; ++s) {
#pragma unroll
        for (int t = 0; t < TILES; ++t) {
            if constexpr (MIX == BF16X3 || MIX == BF16X1) {
#pragma unroll
                for (int k = 0; k < 4; ++k) {
                    const bf16x8 ah = __builtin_bit_cast(bf16x8, raw[(k + t) & 3]), am = __builtin_bit_cast(bf16x8, raw[4 + ((k + t) & 3)]);
                    const bf16x8 bh = __builtin_bit_cast(bf16x8, raw[8 + ((k + 2 * t) & 3)]), bm = __builtin_bit_cast(bf16x8, raw[12 + ((k + 2 * t) & 3)]);
                    if constexpr (MIX == BF16X3) {
                        acc[t] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(am, bh, acc[t], 0, 0, 0);
                        acc[t] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(ah, bm, acc[t], 0, 0, 0);
                    }
                    acc[t] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(ah, bh, acc[t], 0, 0, 0);
                }
            } else {
                if constexpr (MIX != FP6_ONLY) {
#pragma unroll
                    for (int k = 0; k < 4; ++k) {
                        const f16x8 a = __builtin_bit_cast(f16x8, raw[(k + t) & 3]), b = __builtin_bit_cast(f16x8, raw[8 + ((k + 2 * t) & 3)]);
                        acc[t] = __builtin_amdgcn_mfma_f32_16x16x32_f16(a, b, acc[t], 0, 0, 0);
                    }
                }
                if constexpr (MIX == F16_FP6 || MIX == F16_FP8 || MIX == FP6_ONLY || MIX == F16_A6B8 || MIX == F16_A8B6) {
                    constexpr int FA = (MIX == F16_FP8 || MIX == F16_A8B6) ? 0 : 2;   // 0 = e4m3, 2 = e2m3
                    constexpr int FB = (MIX == F16_FP8 || MIX == F16_A6B8) ? 0 : 2;
                    acc[t] = __builtin_amdgcn_mfma_scale_f32_16x16x128_f8f6f4(mxa[t & 1], mxb[(t >> 1) & 1], acc[t], FA, FB, 0, sa, 0, sb);
                    acc[t] = __builtin_amdgcn_mfma_scale_f32_16x16x128_f8f6f4(mxa[(t + 1) & 1], mxb[(t >> 2) & 1], acc[t], FA, FB, 0, sb, 0, sa);
                }
            }
        }
    }
    f32x4 sum = acc[0];
#pragma unroll
    for (int t = 1; t < TILES; ++t) sum += acc[t];
    out[(size_t)blockIdx.x * TPB + tid] = sum.x + sum.y + sum.z + sum.w;
}

static unsigned short f2bf(float f) {
    unsigned u;
    memcpy(&u, &f, 4);
    return (unsigned short)((u + 0x7fff + ((u >> 16) & 1)) >> 16);
}
static unsigned short f2h(float f) {
    _Float16 h = (_Float16)f;
    unsigned short u;
    memcpy(&u, &h, 2);
    return u;
}

int main(int argc, char** argv) {
    const int steps = argc > 1 ? atoi(argv[1]) : 4000;
    int dev = 0, cus = 0;
    hipGetDevice(&dev);
    hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, dev);
    const int blocks = cus;
    const size_t n_u4 = (size_t)64 * TPB * 16;
    // two operand images: 16-bit pieces as bf16 (mix 0 / 5) or as fp16 + MX bytes (the others); random values of small magnitude
    std::mt19937 rng(7);
    std::uniform_real_distribution<float> uni(-1.f, 1.f);
    std::vector<unsigned short> img_bf(n_u4 * 8), img_h(n_u4 * 8);
    for (size_t i = 0; i < img_bf.size(); ++i) {
        const float v = uni(rng) * 0.0625f;
        img_bf[i] = f2bf(v);
        const size_t slot = (i / 8) % 16;   // uint4 index within the lane's 16
        if (slot < 4 || (slot >= 8 && slot < 12))
            img_h[i] = f2h(v);                                   // fp16 fragments
        else
            img_h[i] = (unsigned short)(rng() & 0x7e7e);          // MX bytes: any e2m3 / e4m3 pattern but the e4m3 NaN (0x7f / 0xff)
    }
    uint4 *d_bf, *d_h;
    float* d_out;
    hipMalloc(&d_bf, n_u4 * 16);
    hipMalloc(&d_h, n_u4 * 16);
    hipMalloc(&d_out, (size_t)blocks * TPB * 4);
    hipMemcpy(d_bf, img_bf.data(), n_u4 * 16, hipMemcpyHostToDevice);
    hipMemcpy(d_h, img_h.data(), n_u4 * 16, hipMemcpyHostToDevice);
    const char* names[8] = {"bf16x3 (12 x 16x16x32 bf16 per K=128)", "fp16 + 2 x MX e2m3 (4 x f16 + 2 x scaled 16x16x128)",
                            "fp16 + 2 x MX e4m3", "fp16 alone (4 x f16)", "2 x MX e2m3 alone", "bf16 x1 (4 x bf16)",
                            "fp16 + 2 x MX (A e2m3, B e4m3)", "fp16 + 2 x MX (A e4m3, B e2m3)"};
    const double units[8] = {12, 6, 8, 4, 2, 4, 8, 8};
    hipEvent_t e0, e1;
    hipEventCreate(&e0);
    hipEventCreate(&e1);
    double best[8] = {1e30, 1e30, 1e30, 1e30, 1e30, 1e30, 1e30, 1e30};
    printf("%d CUs, %d blocks x %d threads, %d K=128 steps x %d tiles per wave\n", cus, blocks, TPB, steps, TILES);
    for (int round = 0; round < 4; ++round) {
        for (int mix = 0; mix < 8; ++mix) {
            const uint4* src = (mix == 0 || mix == 5) ? d_bf : d_h;
            for (int rep = 0; rep < 2; ++rep) {   // the second launch is the measured one (clock settled on this mix)
                hipEventRecord(e0, 0);
                switch (mix) {
                    case 0: hipLaunchKernelGGL(mx_mix_kernel<BF16X3>, dim3(blocks), dim3(TPB), 0, 0, src, d_out, steps); break;
                    case 1: hipLaunchKernelGGL(mx_mix_kernel<F16_FP6>, dim3(blocks), dim3(TPB), 0, 0, src, d_out, steps); break;
                    case 2: hipLaunchKernelGGL(mx_mix_kernel<F16_FP8>, dim3(blocks), dim3(TPB), 0, 0, src, d_out, steps); break;
                    case 3: hipLaunchKernelGGL(mx_mix_kernel<F16_ONLY>, dim3(blocks), dim3(TPB), 0, 0, src, d_out, steps); break;
                    case 4: hipLaunchKernelGGL(mx_mix_kernel<FP6_ONLY>, dim3(blocks), dim3(TPB), 0, 0, src, d_out, steps); break;
                    case 5: hipLaunchKernelGGL(mx_mix_kernel<BF16X1>, dim3(blocks), dim3(TPB), 0, 0, src, d_out, steps); break;
                    case 6: hipLaunchKernelGGL(mx_mix_kernel<F16_A6B8>, dim3(blocks), dim3(TPB), 0, 0, src, d_out, steps); break;
                    case 7: hipLaunchKernelGGL(mx_mix_kernel<F16_A8B6>, dim3(blocks), dim3(TPB), 0, 0, src, d_out, steps); break;
                }
                hipEventRecord(e1, 0);
                hipEventSynchronize(e1);
            }
            float ms = 0;
            hipEventElapsedTime(&ms, e0, e1);
            if (ms < best[mix]) best[mix] = ms;
            // tile-ksteps issued: blocks x 12 waves x TILES x steps
            const double tk = (double)blocks * 12 * TILES * steps;
            const double cyc_per_unit = ms * 1e-3 / (tk / (blocks * 4.0)) / units[mix];   // seconds per 16x16x32-equivalent on one SIMD
            printf("round %d  %-52s %8.3f ms   conv0-equivalent %6.3f ms   %.2f ns per 16x16x32-unit per SIMD (%.2f GHz if 16 clocks)\n", round,
                   names[mix], ms, ms * CONV0_TILE_KSTEPS / tk, cyc_per_unit * 1e9, 16.0 / (cyc_per_unit * 1e9));
        }
    }
    printf("\nbest of 4, conv0-equivalent matrix-pipe time (all CUs, nothing else in the loop):\n");
    const double tk = (double)blocks * 12 * TILES * steps;
    for (int mix = 0; mix < 8; ++mix)
        printf("  %-52s %6.3f ms   ratio to bf16x3 %.3f   (paper %.3f)\n", names[mix], best[mix] * CONV0_TILE_KSTEPS / tk, best[mix] / best[0],
               units[mix] / 12.0);
    hipError_t err = hipGetLastError();
    if (err != hipSuccess) printf("HIP error: %s\n", hipGetErrorString(err));
    return err != hipSuccess;
}
