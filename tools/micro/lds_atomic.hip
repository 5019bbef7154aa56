// Microbenchmark: rate of LDS float atomics (ds_add_f32) against plain LDS read-modify-write, per address pattern.
// Build: hipcc -O3 --offload-arch=gfx950 tools/micro/lds_atomic.hip -o tools/micro/lds_atomic
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>

typedef __attribute__((address_space(3))) float lds_float;

template <int MODE>
__global__ __launch_bounds__(256) void lds_kernel(float* out, int iters, int stride, int spread) {
    extern __shared__ float s[];
    const int tid = threadIdx.x;
    for (int i = tid; i < 8192; i += 256) s[i] = 0.f;
    __syncthreads();
    // address pattern: lane l of a wave touches word (l * stride) % spread (+ a per-wave offset)
    const int lane = tid & 63, wave = tid >> 6;
    int a = ((lane * stride) % spread) + wave * 2048;
    float v = (float)tid;
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int u = 0; u < 16; ++u) {
            const int idx = (a + u * 64) & 8191;
            if (MODE == 0) __builtin_amdgcn_ds_faddf((lds_float*)(s + idx), v, 0, 0, false);
            if (MODE == 1) s[idx] += v;                              // read-modify-write, no atomicity
            if (MODE == 2) atomicAdd((int*)(s + idx), tid);          // integer atomic
            if (MODE == 3) atomicAdd((unsigned long long*)(s + (idx & ~1)), (unsigned long long)tid);   // 64-bit integer atomic
            if (MODE == 4) atomicAdd((double*)(s + (idx & ~1)), (double)v);                             // double atomic
            if (MODE == 5) {                                         // 64-bit fixed point from a float: convert + 64-bit integer atomic
                const float x = v * 1.5f;
                const float hi = truncf(x * 0x1p-20f);
                const float lo = x - hi * 0x1p20f;
                const long long q = ((long long)(int)hi << 32) + (long long)(lo * 0x1p12f);
                atomicAdd((unsigned long long*)(s + (idx & ~1)), (unsigned long long)q);
                v += 1.f;
            }
        }
    }
    __syncthreads();
    float acc = 0.f;
    for (int i = tid; i < 8192; i += 256) acc += s[i];
    if (acc == 12345.678f) out[0] = acc;
}

template <int MODE>
static void run(const char* name, int stride, int spread) {
    float* out;
    hipMalloc(&out, 4);
    const int blocks = 256 * 4, iters = 2000;
    hipEvent_t e0, e1;
    hipEventCreate(&e0); hipEventCreate(&e1);
    lds_kernel<MODE><<<blocks, 256, 32768>>>(out, 10, stride, spread);
    hipEventRecord(e0);
    lds_kernel<MODE><<<blocks, 256, 32768>>>(out, iters, stride, spread);
    hipEventRecord(e1);
    hipEventSynchronize(e1);
    float ms;
    hipEventElapsedTime(&ms, e0, e1);
    const double ops = (double)blocks * 256 * iters * 16;
    // 256 CUs at 2.4 GHz
    printf("%-10s stride %3d spread %5d: %8.3f ms  %7.1f G lane-ops/s  %6.2f lane-ops/clk/CU\n", name, stride, spread, ms,
           ops / ms / 1e6, ops / (ms * 1e-3) / (256 * 2.4e9));
    hipFree(out);
}

int main() {
    const int pats[][2] = {{1, 64}, {2, 128}, {4, 256}, {8, 512}, {32, 2048}, {0, 1}};
    for (auto& p : pats) {
        run<0>("ds_add_f32", p[0], p[1]);
        run<1>("rmw", p[0], p[1]);
        run<2>("ds_add_u32", p[0], p[1]);
        run<3>("ds_add_u64", p[0], p[1]);
        run<4>("ds_add_f64", p[0], p[1]);
        run<5>("fixed64", p[0], p[1]);
    }
    return 0;
}
