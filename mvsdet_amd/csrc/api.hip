// Library-level entry points: version, error string, HBM-ceiling copy kernel.
#include "common.h"

#include <algorithm>
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
                        env_int("MVSDET_SWEEP_DSPLIT", 0), env_int("MVSDET_SWEEP_GROUPS", -1),
                        env_int("MVSDET_CONV_SUBPAIRS", 0), env_int("MVSDET_CONV_NSPLIT", 0), env_int("MVSDET_CONV_CGN", 0), env_int("MVSDET_CONV_S2_CG", 0), env_int("MVSDET_CONV_S2_OB", 0), env_int("MVSDET_CONVT_CG", 0),
                        env_int("MVSDET_PROBE_F16_PAIR", 0), env_int("MVSDET_CONV_XCD", 1), env_int("MVSDET_CONV_SPLIT_BLOCKS", 768), env_int("MVSDET_CONV_SPLIT_MIN_GROUPS", 2), env_int("MVSDET_BWD_GROUPS", 0), env_int("MVSDET_CONV_MX_TH", 0), env_int("MVSDET_CONV_MFMA16", 1), env_int("MVSDET_CONVT_PERSIST", 0)};
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
    if (!strcmp(name, "conv_subpairs")) return &o.conv_subpairs;
    if (!strcmp(name, "conv_nsplit")) return &o.conv_nsplit;
    if (!strcmp(name, "conv_mfma16")) return &o.conv_mfma16;
    if (!strcmp(name, "convT_cg")) return &o.convT_cg;
    if (!strcmp(name, "conv_cgn")) return &o.conv_cgn;
    if (!strcmp(name, "conv_s2_cg")) return &o.conv_s2_cg;
    if (!strcmp(name, "conv_s2_ob")) return &o.conv_s2_ob;
    if (!strcmp(name, "probe_f16_pair")) return &o.probe_f16_pair;
    if (!strcmp(name, "conv_xcd")) return &o.conv_xcd;
    if (!strcmp(name, "bwd_groups")) return &o.bwd_groups;
    if (!strcmp(name, "conv_mx_th")) return &o.conv_mx_th;
    if (!strcmp(name, "convT_persist")) return &o.convT_persist;
    if (!strcmp(name, "conv_split_min_groups")) return &o.conv_split_min_groups;
    if (!strcmp(name, "conv_split_blocks")) return &o.conv_split_blocks;
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

// The plane sweep's WRITE pattern alone (no taps, no LDS, no arithmetic): block = (view, TW x 128/TW pixel tile, 32-channel
// slab) loops over the planes; per plane each wave issues four non-temporal stores of 8 channel rows x 16 B per lane -- the
// block -> address map of plane_sweep_variance_kernel's FAST form on an (N,C,D,H,Wo) volume.  What this pattern reaches is the
// ceiling of the sweep's store stream on the box at hand (a device-to-device copy is not: DESIGN.md 4.1).
template <int TW, typename OutT, bool PAIR = false>
__global__ __launch_bounds__(kThreads) void store_pattern_kernel(OutT* __restrict__ var, int C, int D, int H, int W, int Wo,
                                                                  int tiles_x, int tiles, int d_per_block) {
    constexpr int TH = 128 / TW;
    typedef float v4f __attribute__((ext_vector_type(4)));
    typedef unsigned v2u __attribute__((ext_vector_type(2)));
    typedef unsigned v4u __attribute__((ext_vector_type(4)));
    const int S = C / 32;
    const size_t HWo = (size_t)H * Wo;
    const int id = blockIdx.x;
    const int slab = id % S, bt = id / S;
    const int tile = bt % tiles, n = bt / tiles;
    const int tx0 = (tile % tiles_x) * TW, ty0 = (tile / tiles_x) * TH;
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int g = lane & 7, ps = lane >> 3;
    // PAIR (fp16): the lanes of two adjacent pixel quads own the OCTET between them -- the even one its channel rows 0, 1, the odd
    // one rows 2, 3 -- and store 16 bytes each (an experiment: the sweep's flush is not paired)
    const int odd = PAIR ? (ps & 1) : 0;
    const int p0 = 32 * wave + 4 * (ps - odd);
    const int px0 = tx0 + p0 % TW, py = ty0 + p0 / TW;
    if (py >= H || px0 + (PAIR ? 8 : 4) > W) return;
    const size_t st_off = (size_t)py * Wo + px0;
    const v4f vv = {1.0f * id, 2.0f, 3.0f, (float)lane};
    const v2u hv = {(unsigned)id, (unsigned)lane};   // fp16 storage: the lane's 4 pixels are 8 bytes
    const v4u pv = {(unsigned)id, (unsigned)lane, 3u, 4u};
    const int d0 = blockIdx.y * d_per_block, d1 = min(D, d0 + d_per_block);
    for (int d = d0; d < d1; ++d) {
        if constexpr (PAIR) {
#pragma unroll
            for (int j = 0; j < 2; ++j) {
                const int c = slab * 32 + 8 * (2 * odd + j) + g;
                OutT* dst = var + (((size_t)n * C + c) * D + d) * HWo + st_off;
                __builtin_nontemporal_store(pv, reinterpret_cast<v4u*>(dst));
            }
        } else {
#pragma unroll
            for (int i = 0; i < 4; ++i) {
                const int c = slab * 32 + 8 * i + g;
                OutT* dst = var + (((size_t)n * C + c) * D + d) * HWo + st_off;
                if constexpr (sizeof(OutT) == 4) __builtin_nontemporal_store(vv, reinterpret_cast<v4f*>(dst));
                else __builtin_nontemporal_store(hv, reinterpret_cast<v2u*>(dst));
            }
        }
    }
}

}  // namespace mvsdet

using namespace mvsdet;

extern "C" int mvsdet_version(void) { return 6001; }

template <typename OutT>
static int store_pattern_probe(OutT* var, int N, int C, int D, int H, int W, int out_w_pitch, int tile_w, int planes_per_block,
                               mvsdet_stream_t stream) {
    MVS_REQUIRE(var, "store_pattern_probe: NULL pointer");
    MVS_REQUIRE(N > 0 && C > 0 && C % 32 == 0 && D > 0 && H > 0 && W > 0 && W % 4 == 0, "store_pattern_probe: C %% 32 == 0 and W %% 4 == 0 wanted (the sweep's FAST form)");
    const int Wo = out_w_pitch > 0 ? out_w_pitch : W;
    MVS_REQUIRE(Wo >= W && Wo % 4 == 0 && ((uintptr_t)var % 16 == 0), "store_pattern_probe: pitch / alignment");
    MVS_REQUIRE(tile_w == 16 || tile_w == 32 || tile_w == 64, "store_pattern_probe: tile width 16, 32 or 64");
    const int th = 128 / tile_w;
    const int tiles_x = (W + tile_w - 1) / tile_w, tiles = tiles_x * ((H + th - 1) / th);
    const long long blocks = (long long)N * tiles * (C / 32);
    MVS_REQUIRE(blocks <= INT32_MAX, "store_pattern_probe: grid too large");
    const int dpb = planes_per_block > 0 ? std::min(planes_per_block, D) : D;
    dim3 grid((unsigned)blocks, (unsigned)((D + dpb - 1) / dpb));
    if (sizeof(OutT) == 2 && options().probe_f16_pair && W % 8 == 0 && Wo % 8 == 0 && tile_w >= 16) {
        // option "probe_f16_pair": what a paired flush of the fp16 kernel would reach
        if (tile_w == 16) hipLaunchKernelGGL((store_pattern_kernel<16, OutT, true>), grid, dim3(kThreads), 0, (hipStream_t)stream, var, C, D, H, W, Wo, tiles_x, tiles, dpb);
        else if (tile_w == 64) hipLaunchKernelGGL((store_pattern_kernel<64, OutT, true>), grid, dim3(kThreads), 0, (hipStream_t)stream, var, C, D, H, W, Wo, tiles_x, tiles, dpb);
        else hipLaunchKernelGGL((store_pattern_kernel<32, OutT, true>), grid, dim3(kThreads), 0, (hipStream_t)stream, var, C, D, H, W, Wo, tiles_x, tiles, dpb);
        MVS_LAUNCH_CHECK("store_pattern_probe");
        return MVSDET_OK;
    }
    if (tile_w == 16) hipLaunchKernelGGL((store_pattern_kernel<16, OutT>), grid, dim3(kThreads), 0, (hipStream_t)stream, var, C, D, H, W, Wo, tiles_x, tiles, dpb);
    else if (tile_w == 64) hipLaunchKernelGGL((store_pattern_kernel<64, OutT>), grid, dim3(kThreads), 0, (hipStream_t)stream, var, C, D, H, W, Wo, tiles_x, tiles, dpb);
    else hipLaunchKernelGGL((store_pattern_kernel<32, OutT>), grid, dim3(kThreads), 0, (hipStream_t)stream, var, C, D, H, W, Wo, tiles_x, tiles, dpb);
    MVS_LAUNCH_CHECK("store_pattern_probe");
    return MVSDET_OK;
}

extern "C" int mvsdet_store_pattern_probe_f32(float* var, int N, int C, int D, int H, int W, int out_w_pitch, int tile_w,
                                              int planes_per_block, mvsdet_stream_t stream) {
    return store_pattern_probe(var, N, C, D, H, W, out_w_pitch, tile_w, planes_per_block, stream);
}

extern "C" int mvsdet_store_pattern_probe_f16(void* var, int N, int C, int D, int H, int W, int out_w_pitch, int tile_w,
                                              int planes_per_block, mvsdet_stream_t stream) {
    return store_pattern_probe(static_cast<unsigned short*>(var), N, C, D, H, W, out_w_pitch, tile_w, planes_per_block, stream);
}

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
