// What exactly does v_mfma_scale_f32_16x16x128_f8f6f4 compute on e2m3 (FP6) operands?  "Check the map with exact integer data before
// relying on it" (cdna_hip_programming.md, fragment layout).  Assumed, and tested here against a host sum with exact data:
//   A operand: lane l holds row (l & 15), k-block (l >> 4): k = 32 * (l >> 4) + e, e = 0 .. 31; element e = bits [6e, 6e + 5] of the
//              192-bit little-endian string in the fragment's first six registers (sign bit 5, exponent bits 4:3 bias 1, mantissa 2:0)
//   B operand: lane l holds column (l & 15), the same k-block map
//   scale:     byte 0 (opsel 0) of the lane's scale register is an e8m0 exponent: the lane's 32 elements are multiplied by 2^(byte - 127)
//   C / D:     column = lane & 15, row = 4 * (lane >> 4) + register
// Build + run on the GPU box:  hipcc -O3 --offload-arch=gfx950 -o tools/micro/mx_probe tools/micro/mx_probe.hip && tools/micro/mx_probe
#include <hip/hip_runtime.h>

#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <random>
#include <vector>

typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef int i32x8 __attribute__((ext_vector_type(8)));

__global__ void probe(const unsigned* __restrict__ a, const unsigned* __restrict__ b, const int* __restrict__ sa, const int* __restrict__ sb,
                      float* __restrict__ c) {
    const int l = threadIdx.x;
    i32x8 fa, fb;
#pragma unroll
    for (int i = 0; i < 8; ++i) {
        fa[i] = i < 6 ? (int)a[l * 6 + i] : 0;
        fb[i] = i < 6 ? (int)b[l * 6 + i] : 0;
    }
    f32x4 acc = {0.f, 0.f, 0.f, 0.f};
    acc = __builtin_amdgcn_mfma_scale_f32_16x16x128_f8f6f4(fa, fb, acc, 2, 2, 0, sa[l], 0, sb[l]);
#pragma unroll
    for (int r = 0; r < 4; ++r) c[(4 * (l >> 4) + r) * 16 + (l & 15)] = acc[r];
}

static float e2m3_value(unsigned code) {
    const int s = (code >> 5) & 1, e = (code >> 3) & 3, m = code & 7;
    const float v = e == 0 ? m / 8.0f : (1.0f + m / 8.0f) * (float)(1 << (e - 1));
    return s ? -v : v;
}

int main() {
    std::mt19937 rng(3);
    std::vector<unsigned> ca(16 * 128), cb(128 * 16);     // codes: A[row][k], B[k][col]
    for (auto& v : ca) v = rng() & 63;
    for (auto& v : cb) v = rng() & 63;
    std::vector<int> sa(64), sb(64);
    for (int l = 0; l < 64; ++l) { sa[l] = 120 + (int)(rng() % 12) + (0x5a << 8); sb[l] = 125 + (int)(rng() % 6) + (0x33 << 16); }   // junk in the other bytes
    std::vector<unsigned> fa(64 * 6, 0), fb(64 * 6, 0);
    for (int l = 0; l < 64; ++l)
        for (int e = 0; e < 32; ++e) {
            const int k = 32 * (l >> 4) + e, bit = 6 * e;
            const unsigned long long va = ca[(l & 15) * 128 + k], vb = cb[k * 16 + (l & 15)];
            for (int which = 0; which < 2; ++which) {
                unsigned* f = (which ? fb.data() : fa.data()) + l * 6;
                const unsigned long long v = which ? vb : va;
                f[bit / 32] |= (unsigned)(v << (bit % 32));
                if (bit % 32 > 26) f[bit / 32 + 1] |= (unsigned)(v >> (32 - bit % 32));
            }
        }
    unsigned *da, *db;
    int *dsa, *dsb;
    float* dc;
    hipMalloc(&da, fa.size() * 4); hipMalloc(&db, fb.size() * 4); hipMalloc(&dsa, 256); hipMalloc(&dsb, 256); hipMalloc(&dc, 1024);
    hipMemcpy(da, fa.data(), fa.size() * 4, hipMemcpyHostToDevice);
    hipMemcpy(db, fb.data(), fb.size() * 4, hipMemcpyHostToDevice);
    hipMemcpy(dsa, sa.data(), 256, hipMemcpyHostToDevice);
    hipMemcpy(dsb, sb.data(), 256, hipMemcpyHostToDevice);
    hipLaunchKernelGGL(probe, dim3(1), dim3(64), 0, 0, da, db, dsa, dsb, dc);
    std::vector<float> c(256);
    hipMemcpy(c.data(), dc, 1024, hipMemcpyDeviceToHost);
    double worst = 0, scale_ref = 0;
    for (int i = 0; i < 16; ++i)
        for (int j = 0; j < 16; ++j) {
            double ref = 0;
            for (int kb = 0; kb < 4; ++kb) {
                // the scale of A's block (row i, kb) sits in lane 16*kb + i, of B's block (col j, kb) in lane 16*kb + j
                const double s = std::ldexp(1.0, (sa[16 * kb + i] & 255) - 127) * std::ldexp(1.0, (sb[16 * kb + j] & 255) - 127);
                double part = 0;
                for (int e = 0; e < 32; ++e) part += (double)e2m3_value(ca[i * 128 + 32 * kb + e]) * (double)e2m3_value(cb[(32 * kb + e) * 16 + j]);
                ref += part * s;
            }
            worst = std::fmax(worst, std::fabs(ref - (double)c[i * 16 + j]));
            scale_ref = std::fmax(scale_ref, std::fabs(ref));
        }
    printf("v_mfma_scale_f32_16x16x128_f8f6f4, e2m3 x e2m3 with per-lane e8m0 scales: max |device - host| = %.3e of |max| %.3e -> %s\n", worst,
           scale_ref, worst <= 1e-6 * scale_ref ? "the assumed operand map, element order and scale rule HOLD" : "MISMATCH");
    printf("C[0][0..3] device %g %g %g %g\n", c[0], c[1], c[2], c[3]);
    return worst <= 1e-6 * scale_ref ? 0 : 1;
}
