// Library-level entry points: version, error string, HBM-ceiling copy kernel.
#include "common.h"

#include <cstdlib>
#include <cstring>

namespace mvsdet {

static thread_local char g_err[512] = "";

void set_error(const char* fmt, ...) {
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(g_err, sizeof(g_err), fmt, ap);
    va_end(ap);
}

static int env_int(const char* name, int dflt) {
    const char* e = getenv(name);
    return e ? atoi(e) : dflt;
}

Options& options() {
    static Options o = {env_int("MVSDET_SWEEP_TW", 0), env_int("MVSDET_SWEEP_BOXCAP", 512), env_int("MVSDET_SWEEP_XCD", 1),
                        env_int("MVSDET_SWEEP_DSPLIT", 0), env_int("MVSDET_SWEEP_GROUPS", -1)};
    return o;
}

static int* option_slot(const char* name) {
    Options& o = options();
    if (!name) return nullptr;
    if (!strcmp(name, "sweep_tw")) return &o.sweep_tw;
    if (!strcmp(name, "sweep_boxcap")) return &o.sweep_boxcap;
    if (!strcmp(name, "sweep_xcd")) return &o.sweep_xcd;
    if (!strcmp(name, "sweep_dsplit")) return &o.sweep_dsplit;
    if (!strcmp(name, "sweep_groups")) return &o.sweep_groups;
    return nullptr;
}

// float4 grid-stride copy: the achievable-HBM yardstick of bench.py (MI355X_MICROARCH: 6.29 TB/s measured).
__global__ __launch_bounds__(kThreads) void copy_kernel(const float4* __restrict__ src, float4* __restrict__ dst, size_t n4) {
    // 4 independent 16-byte loads in flight per lane before the first store
    const size_t stride = (size_t)gridDim.x * kThreads;
    size_t i = (size_t)blockIdx.x * kThreads + threadIdx.x;
    for (; i + 3 * stride < n4; i += 4 * stride) {
        const float4 a = src[i], b = src[i + stride], c = src[i + 2 * stride], d = src[i + 3 * stride];
        dst[i] = a;
        dst[i + stride] = b;
        dst[i + 2 * stride] = c;
        dst[i + 3 * stride] = d;
    }
    for (; i < n4; i += stride) dst[i] = src[i];
}

__global__ void copy_tail_kernel(const float* __restrict__ src, float* __restrict__ dst, size_t begin, size_t n) {
    const size_t i = begin + threadIdx.x;
    if (i < n) dst[i] = src[i];
}

}  // namespace mvsdet

using namespace mvsdet;

extern "C" int mvsdet_version(void) { return 3002; }

extern "C" int mvsdet_set_option(const char* name, int value) {
    int* slot = option_slot(name);
    MVS_REQUIRE(slot, "set_option: unknown option '%s'", name ? name : "(null)");
    *slot = value;
    return MVSDET_OK;
}

extern "C" int mvsdet_validate_neighbors(const int64_t* nbr, int M, int K, int n_src) {
    MVS_REQUIRE(M >= 0 && K >= 0 && n_src > 0, "validate_neighbors: bad shape M=%d K=%d n_src=%d", M, K, n_src);
    MVS_REQUIRE(K == 0 || M == 0 || nbr, "validate_neighbors: NULL pointer");
    for (int m = 0; m < M; ++m)
        for (int j = 0; j < K; ++j) {
            const int64_t v = nbr[(size_t)m * K + j];
            MVS_REQUIRE(v >= 0 && v < n_src, "neighbour id %lld of view %d (slot %d) outside [0, %d)", (long long)v, m, j, n_src);
        }
    return MVSDET_OK;
}

extern "C" int mvsdet_get_option(const char* name, int* value) {
    int* slot = option_slot(name);
    MVS_REQUIRE(slot && value, "get_option: unknown option '%s'", name ? name : "(null)");
    *value = *slot;
    return MVSDET_OK;
}

extern "C" const char* mvsdet_last_error(void) { return g_err; }

extern "C" int mvsdet_copy_f32(const float* src, float* dst, size_t n, mvsdet_stream_t stream) {
    MVS_REQUIRE(src && dst, "copy: NULL pointer");
    MVS_REQUIRE(((uintptr_t)src % 16 == 0) && ((uintptr_t)dst % 16 == 0), "copy: pointers must be 16-byte aligned");
    const size_t n4 = n / 4;
    if (n4) {
        const size_t want = (n4 + kThreads - 1) / kThreads;
        const unsigned grid = (unsigned)(want < 2048 ? want : 2048);  // 256 CUs x 8 blocks, grid-stride the rest
        hipLaunchKernelGGL(copy_kernel, dim3(grid), dim3(kThreads), 0, (hipStream_t)stream, (const float4*)src, (float4*)dst, n4);
    }
    if (n % 4) hipLaunchKernelGGL(copy_tail_kernel, dim3(1), dim3(4), 0, (hipStream_t)stream, src, dst, n4 * 4, n);
    MVS_LAUNCH_CHECK("copy");
    return MVSDET_OK;
}
