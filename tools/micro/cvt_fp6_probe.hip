// What does v_cvt_scalef32_2xpk16_fp6_f32 produce?  32 floats (two groups of 16) and one scale in, 32 e2m3 codes in six registers out.
// Probed with values that are exact in e2m3 and all different in their code, and with values between grid points (rounding mode).
// Build + run on the GPU box:  hipcc -O3 --offload-arch=gfx950 -o tools/micro/cvt_fp6_probe tools/micro/cvt_fp6_probe.hip && tools/micro/cvt_fp6_probe
#include <hip/hip_runtime.h>
#include <cmath>
#include <cstdio>
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef unsigned u32x6 __attribute__((ext_vector_type(6)));

__global__ void probe(const float* __restrict__ in, float scale, unsigned* __restrict__ out) {
    f32x16 a, b;
#pragma unroll
    for (int i = 0; i < 16; ++i) { a[i] = in[i]; b[i] = in[16 + i]; }
    const u32x6 r = __builtin_amdgcn_cvt_scalef32_2xpk16_fp6_f32(a, b, scale);
#pragma unroll
    for (int i = 0; i < 6; ++i) out[i] = r[i];
}

static float e2m3_value(unsigned code) {
    const int s = (code >> 5) & 1, e = (code >> 3) & 3, m = code & 7;
    const float v = e == 0 ? m / 8.0f : (1.0f + m / 8.0f) * (float)(1 << (e - 1));
    return s ? -v : v;
}

static void run(const char* what, const float* h, float scale) {
    float* d; unsigned* o;
    hipMalloc(&d, 128); hipMalloc(&o, 24);
    hipMemcpy(d, h, 128, hipMemcpyHostToDevice);
    hipLaunchKernelGGL(probe, dim3(1), dim3(1), 0, 0, d, scale, o);
    unsigned r[6];
    hipMemcpy(r, o, 24, hipMemcpyDeviceToHost);
    printf("%s (scale %g):\n  in : ", what, scale);
    for (int i = 0; i < 32; ++i) printf("%g ", h[i]);
    printf("\n  out: ");
    for (int e = 0; e < 32; ++e) {
        const int bit = 6 * e;
        unsigned long long w = r[bit / 32] | ((unsigned long long)(bit / 32 + 1 < 6 ? r[bit / 32 + 1] : 0) << 32);
        printf("%g ", e2m3_value((unsigned)((w >> (bit % 32)) & 63)));
    }
    printf("\n");
    hipFree(d); hipFree(o);
}

int main() {
    float h[32];
    for (int i = 0; i < 32; ++i) h[i] = e2m3_value((unsigned)i);                      // codes 0..31: 0, .125, ... 7.5
    run("group a = codes 0..15, group b = codes 16..31", h, 1.0f);
    for (int i = 0; i < 32; ++i) h[i] = e2m3_value((unsigned)i) * 4.0f;
    run("the same values x 4, scale 4 (is the scale a divisor?)", h, 4.0f);
    const float mid[32] = {0.0625f, 0.1875f, 0.3125f, 1.0625f, 1.1875f, 2.125f, 2.375f, 4.25f, 4.75f, 7.6f, 7.75f, 8.0f, 100.0f, -0.0625f, -0.1875f, -7.75f,
                           0.06f, 0.07f, 1.06f, 1.07f, 2.12f, 2.13f, 4.24f, 4.26f, 0.0f, -0.0f, 3.9f, 3.95f, 1.9f, 1.95f, 0.99f, 0.93f};
    run("between grid points (ties: 0.0625 -> 0 or .125?, 0.1875 -> .125 or .25?; saturation)", mid, 1.0f);
    return 0;
}
