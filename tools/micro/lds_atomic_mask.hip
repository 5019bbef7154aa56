// Microbenchmark: what a ds_add_f64 / ds_add_u64 / ds_add_u32 wave-instruction costs when part of its lanes are masked off, on
// the address pattern of the backward sweep's gradient image (8 texels of 128 bytes per instruction, lane g of a texel the
// double 2g + i).  Question behind it: does merging the taps adjacent pixels share (fewer ACTIVE lanes per instruction, or
// fewer instructions) shorten an LDS-atomic-bound loop?
// Build: hipcc -O3 --offload-arch=gfx950 tools/micro/lds_atomic_mask.hip -o tools/micro/lds_atomic_mask
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>

// MODE 0: f64, 1: u64, 2: u32 (the 128-byte texel as 32 words, lane g words 4g..4g+3 -> i of 4), 3: f32
template <int MODE>
__global__ __launch_bounds__(256, 2) void k(float* out, int iters, unsigned long long mask, int texel_step) {
    extern __shared__ double s[];   // 64 KiB: 512 texels of 16 doubles
    const int tid = threadIdx.x;
    for (int i = tid; i < 8192; i += 256) s[i] = 0.0;
    __syncthreads();
    const int lane = tid & 63, wave = tid >> 6, g = lane & 7, ps = lane >> 3;
    const bool on = (mask >> lane) & 1ull;
    int texel = (wave * 37 + ps * texel_step) & 511;
    float v = (float)tid;
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int u = 0; u < 16; ++u) {
            const int t = (texel + u * 5) & 511;
            if (on) {
                if (MODE == 0) __hip_atomic_fetch_add(s + t * 16 + 2 * g + (u & 1), (double)v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
                if (MODE == 1) __hip_atomic_fetch_add((unsigned long long*)s + t * 16 + 2 * g + (u & 1), (unsigned long long)tid, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
                if (MODE == 2) __hip_atomic_fetch_add((unsigned*)s + t * 32 + 4 * g + (u & 3), (unsigned)tid, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
                if (MODE == 3) __hip_atomic_fetch_add((float*)s + t * 32 + 4 * g + (u & 3), v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
            }
        }
        texel = (texel + 11) & 511;
    }
    __syncthreads();
    double acc = 0.0;
    for (int i = tid; i < 8192; i += 256) acc += s[i];
    if (acc == 12345.678) out[0] = (float)acc;
}

template <int MODE>
static void run(const char* name, unsigned long long mask, int texel_step) {
    float* out;
    hipMalloc(&out, 4);
    const int blocks = 256 * 4, iters = 1000;
    hipEvent_t e0, e1;
    hipEventCreate(&e0); hipEventCreate(&e1);
    k<MODE><<<blocks, 256, 65536>>>(out, 10, mask, texel_step);
    hipEventRecord(e0);
    k<MODE><<<blocks, 256, 65536>>>(out, iters, mask, texel_step);
    hipEventRecord(e1);
    hipEventSynchronize(e1);
    float ms;
    hipEventElapsedTime(&ms, e0, e1);
    const double winst = (double)blocks * 4 * iters * 16;
    const double lanes = winst * __builtin_popcountll(mask);
    printf("%-10s mask %016llx step %2d: %8.3f ms  %6.2f wave-instr/clk/CU x1e-2  %6.2f active lane-ops/clk/CU\n", name, mask, texel_step, ms,
           100.0 * winst / (ms * 1e-3) / (256 * 2.4e9), lanes / (ms * 1e-3) / (256 * 2.4e9));
    hipFree(out);
}

int main() {
    const unsigned long long masks[] = {~0ull, 0x5555555555555555ull, 0x00ff00ff00ff00ffull, 0x00000000ffffffffull,
                                        0x000000ff000000ffull, 0x1111111111111111ull};
    for (int step : {1, 4}) {
        for (auto m : masks) {
            run<0>("ds_add_f64", m, step);
            run<1>("ds_add_u64", m, step);
            run<2>("ds_add_u32", m, step);
            run<3>("ds_add_f32", m, step);
        }
    }
    return 0;
}
