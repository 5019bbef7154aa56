// Stage 1 of the MVSDet hot path on gfx950: feature packing, homography warp (a3) and the fused
// plane-sweep variance cost volume (a3+a4).  Reference: mvs_models/module.py:105-146 and
// mvsdet.py:439-467 of Pixie8888/MVSDet (projects/NeRF-Det/nerfdet/).
//
// Roofline: HBM.  Per cost volume the kernel must write C*D*H*W*4 B and read (K+1)*C*H*W*4 B; there is
// no dense contraction, so MFMA does not apply.  The design problem is the gather: 4*K
// bilinear taps per output element, i.e. 8x more bytes through the load path than are written.
//
// v1 of this kernel (git history; profiles/r01_v1_*) gathered the taps straight from a channel-last copy
// of the maps with every block covering all 256 channels.  rocprofv3 showed it saturating the fabric
// (2*FETCH_SIZE + WRITE_SIZE = 183 GB for 52.7 GB algorithmic, 6.3 TB/s) at a 43 % L2 hit rate: the 64
// blocks resident on an XCD touched ~8 MB of source rows at once, twice its L2.  This version
//   (1) splits the channels into 32-channel slabs and deals slab s to XCD s (8 slabs = 8 XCDs at C=256),
//       so an XCD only ever reads ONE 128-byte slab of each source texel and its live working set
//       (2 source images x 128 B/texel x the rows in flight) fits its 4 MiB L2;
//   (2) uses 2-D pixel tiles (32x4 / 16x8) and stages the bounding box of a tile's sampling footprint in
//       LDS with fully coalesced row reads, so each source texel is fetched once per (tile, plane) instead
//       of once per tap (about 1.3-2 texels per output pixel and neighbour instead of 4), and the taps
//       become ds_read_b128;
//   (3) keeps the LDS transpose so the (N,C,D,H,W) output is written as whole 128-byte rows, with
//       non-temporal stores so the write-once stream does not evict the source images.
// A footprint that does not fit the LDS box (extreme roll / scale, or planes behind the camera) falls back
// to direct global gathers for that (tile, plane, neighbour) -- same arithmetic, same result.
#include "common.h"

#include <algorithm>
#include "pack.h"
#include "sweep_kernel.h"

namespace mvsdet {

// ---------------------------------------------------------------------------------------------
// a3: homo_warping with the reference's own layout (NCHW source, one source view per batch row).
// Compatibility operator: one thread per output pixel, loop over channels; stores are coalesced
// along W, taps are per-lane dword gathers.  The fused kernel below is the fast path.
// ---------------------------------------------------------------------------------------------
__global__ __launch_bounds__(kThreads) void homo_warp_kernel(const float* __restrict__ src,
                                                              const float* __restrict__ proj,
                                                              const float* __restrict__ depth,
                                                              float* __restrict__ out, int C, int D, int H, int W) {
    const int HW = H * W;
    const int pix = blockIdx.x * kThreads + threadIdx.x;
    const int d = blockIdx.y, b = blockIdx.z;
    if (pix >= HW) return;
    const int y = pix / W, x = pix - y * W;
    int4 off;
    float4 w;
    compute_taps(proj + (size_t)b * 16, (float)x, (float)y, depth[(size_t)b * D + d], H, W, 1, off, w);
    const float* plane = src + (size_t)b * C * HW;
    float* o = out + ((size_t)b * C * D + d) * HW + pix;
    for (int c = 0; c < C; ++c) {
        float s = plane[off.x] * w.x;
        s = fmaf(plane[off.y], w.y, s);
        s = fmaf(plane[off.z], w.z, s);
        s = fmaf(plane[off.w], w.w, s);
        *o = s;
        plane += HW;
        o += (size_t)D * HW;
    }
}

}  // namespace mvsdet

using namespace mvsdet;

extern "C" size_t mvsdet_packed_bytes(int N, int C, int H, int W) {
    if (N <= 0 || C <= 0 || H <= 0 || W <= 0) return 0;
    return (size_t)N * num_slabs(C) * H * W * kSlab * sizeof(float);
}

template <typename InT>
static int pack_entry(const char* name, const InT* feat, const int64_t* fs, float* packed, int N, int C, int H, int W,
                      mvsdet_stream_t stream) {
    MVS_REQUIRE(feat && fs && packed, "%s: NULL pointer", name);
    MVS_REQUIRE(N > 0 && C > 0 && H > 0 && W > 0, "%s: bad shape N=%d C=%d H=%d W=%d", name, N, C, H, W);
    MVS_REQUIRE((size_t)H * W * kSlab < (size_t)INT32_MAX, "%s: one slab image exceeds 2^31 elements", name);
    MVS_REQUIRE(N <= 65535 && num_slabs(C) <= 65535, "%s: N or C too large", name);
    const int S = num_slabs(C);
    if constexpr (sizeof(InT) == 4) {
        // dense maps (the usual case: the 2-D backbone's output; a cropped view takes the general kernel): float4 loads
        if (fs[3] == 1 && fs[2] == W && (H * W) % 4 == 0 && fs[1] % 4 == 0 && fs[0] % 4 == 0 && (uintptr_t)feat % 16 == 0) {
            dim3 dgrid((H * W + 127) / 128, S, N);
            hipLaunchKernelGGL(pack_features_dense_kernel, dgrid, dim3(kThreads), 0, (hipStream_t)stream, (const float*)feat, fs[0],
                               fs[1], packed, C, S, H * W);
            MVS_LAUNCH_CHECK(name);
            return MVSDET_OK;
        }
    }
    dim3 grid((H * W + 63) / 64, S, N);
    hipLaunchKernelGGL(pack_features_kernel<InT>, grid, dim3(kThreads), 0, (hipStream_t)stream, feat, fs[0], fs[1], fs[2],
                       fs[3], packed, C, S, H, W);
    MVS_LAUNCH_CHECK(name);
    return MVSDET_OK;
}

extern "C" int mvsdet_pack_features_f32(const float* feat, const int64_t* fs, float* packed, int N, int C, int H, int W,
                                        mvsdet_stream_t stream) {
    return pack_entry<float>("pack_features", feat, fs, packed, N, C, H, W, stream);
}

extern "C" int mvsdet_pack_features_f16(const void* feat_f16, const int64_t* fs, float* packed, int N, int C, int H, int W,
                                        mvsdet_stream_t stream) {
    return pack_entry<__half>("pack_features_f16", static_cast<const __half*>(feat_f16), fs, packed, N, C, H, W, stream);
}

extern "C" int mvsdet_homo_warp_f32(const float* src, const float* proj, const float* depth, float* out, int B, int C,
                                    int D, int H, int W, mvsdet_stream_t stream) {
    MVS_REQUIRE(src && proj && depth && out, "homo_warp: NULL pointer");
    MVS_REQUIRE(B > 0 && C > 0 && D > 0 && H > 1 && W > 1, "homo_warp: bad shape B=%d C=%d D=%d H=%d W=%d", B, C, D, H, W);
    MVS_REQUIRE(B <= 65535 && D <= 65535, "homo_warp: B or D > 65535");
    MVS_REQUIRE((size_t)C * H * W < (size_t)INT32_MAX, "homo_warp: one view exceeds 2^31 elements");
    dim3 grid((H * W + kThreads - 1) / kThreads, D, B);
    hipLaunchKernelGGL(homo_warp_kernel, grid, dim3(kThreads), 0, (hipStream_t)stream, src, proj, depth, out, C, D, H, W);
    MVS_LAUNCH_CHECK("homo_warp");
    return MVSDET_OK;
}

namespace {
// 32x4 tiles write 128-byte runs per channel row, 16x8 tiles 64-byte ones (5.5 against 3.6 TB/s of pattern
// ceiling).  A width that is a multiple of 16 but not of 32 (the 80-wide maps of the shipped configs) leaves a third column
// of 32x4 tiles half empty: with many planes per block the wider runs win all the same (50 views x 96 planes, 60x80:
// 6.6 against 7.1 ms), with the reference's 12 planes the smaller footprints and the fuller grid of the 16x8 tiles do.
// A PITCHED output (row pitch a multiple of 32 elements, chosen by the caller: every row then starts on a 128-byte line) takes
// the 32x4 tiles whatever the width: tools/micro/store_pattern_ref.hip, 60x80 maps x 12 planes, stores only: 16x8 tiles on
// the contiguous volume 0.66 ms (3.5 TB/s: 64-byte runs; 32x4 tiles there straddle two lines on every odd row: 0.55),
// 32x4 tiles on rows pitched to 96: 0.46 ms.
int pick_tile_width(int W, int D, int out_pitch = 0) {
    const int tw = options().sweep_tw;  // tuning knob (mvsdet_set_option "sweep_tw"); 0 = by the map shape
    if (tw == 16 || tw == 32) return tw;
    if (out_pitch > 0 && out_pitch != W && out_pitch % 32 == 0) return 32;
    if (W % 32 == 0 || W % 16 != 0) return 32;
    return D >= 48 ? 32 : 16;
}

int num_tiles(int H, int W, int tw) {
    const int th = kTilePix / tw;
    return ((W + tw - 1) / tw) * ((H + th - 1) / th);
}

// texels of one LDS footprint box.  K resident boxes of 128-byte texels share the block's LDS: 80 KiB = two blocks per CU
// for the 32x4 tiles (their footprints need ~300 texels to form long runs); the 16x8 tiles of maps whose width is no
// multiple of 32 have smaller footprints and run three blocks per CU on 52 KiB (+5 % at the reference-true shape).
// the largest capacity a table of this (K, tile shape) can have been built with, whatever "sweep_boxcap" said then
int max_box_cap(int K, int tw) {
    if (K <= 0) return 0;
    const int budget = (tw == 16 && K <= 2) ? 52 * 1024 : 80 * 1024;
    const int fit = (budget / 128) / K - kBoxPad;
    return fit > 512 ? 512 : fit;
}

int effective_box_cap(int K, int tw) {
    if (K <= 0) return 0;
    const int cap = options().sweep_boxcap, fit = max_box_cap(K, tw);
    return cap < 0 ? 0 : (cap > fit ? fit : cap);
}

template <typename F>
int allow_dynamic_lds(F* kernel, size_t bytes) {
    // above 48 KiB of dynamic LDS the launch needs the function attribute; set once per kernel (host-side state,
    // nothing is enqueued: safe under stream capture)
    if (bytes <= 48 * 1024) return MVSDET_OK;
    if (hipFuncSetAttribute(reinterpret_cast<const void*>(kernel), hipFuncAttributeMaxDynamicSharedMemorySize, 80 * 1024) != hipSuccess) {
        set_error("plane_sweep_variance: hipFuncSetAttribute(MaxDynamicSharedMemorySize) failed");
        return MVSDET_ERR_HIP;
    }
    return MVSDET_OK;
}

size_t align16(size_t v) { return (v + 15) & ~(size_t)15; }

}  // namespace

namespace mvsdet {
// Sweep geometry in the scratch buffer (built by plane_sweep_coords_kernel, consumed by the slab kernels):
//   header int4 | boxes [N*tiles*D*K] int4 | flags [N*tiles*D] u32 | proj copy [N*K*16] f32 | depth copy [N*D] f32 |
//   plane groups [N*tiles][kSweepGroups+1] u16
// `tiles` is that of the tile width in force; the size query assumes the larger of the two tile shapes.
SweepGeometry sweep_geometry(void* scratch, int N, int K, int D, int tiles) {
    SweepGeometry g;
    char* p = static_cast<char*>(scratch);
    g.header = reinterpret_cast<int4*>(p);   // first, so that its place does not depend on the tile shape
    p += sizeof(int4);
    g.boxes = reinterpret_cast<int4*>(p);
    p += (size_t)N * tiles * D * K * sizeof(int4);
    g.flags = reinterpret_cast<unsigned*>(p);
    p += align16((size_t)N * tiles * D * sizeof(unsigned));
    g.proj = reinterpret_cast<float*>(p);
    p += align16((size_t)N * K * 16 * sizeof(float));
    g.depth = reinterpret_cast<float*>(p);
    p += align16((size_t)N * D * sizeof(float));
    g.groups = reinterpret_cast<unsigned short*>(p);
    return g;
}
}  // namespace mvsdet

namespace {
template <int KV, int TW, bool FAST, typename OutT>
int launch_slab(dim3 grid, hipStream_t stream, size_t lds, const float* packed, const float* ref_packed, const int64_t* nbr,
                const SweepGeometry& geo, const unsigned short* groups, OutT* var, int n_src, int C, int S, int D, int H, int W,
                int Wo, int tiles_x, int tiles, int d_per_block, int box_cap, int n_bt, int xcd_parts) {
    auto* k = plane_sweep_variance_kernel<KV, TW, FAST, OutT>;
    if (int rc = allow_dynamic_lds(k, lds)) return rc;
    hipLaunchKernelGGL(k, grid, dim3(kThreads), lds, stream, packed, ref_packed, nbr, geo.header, geo.proj, geo.depth, geo.boxes, geo.flags,
                       groups, var, n_src, C, S, D, H, W, Wo, tiles_x, tiles, d_per_block, box_cap, n_bt, xcd_parts);
    return MVSDET_OK;
}

template <int TW>
int launch_sweep(const float* packed, const int64_t* nbr, const float* proj, const float* depth, void* var_any,
                 void* scratch, int N, int K, int C, int D, int H, int W, hipStream_t stream, int phases, int n_src,
                 int ref_first, bool half_out, int Wo) {
    float* var = static_cast<float*>(var_any);
    __half* var16 = static_cast<__half*>(var_any);  // half_out: same kernel, variance rounded to fp16 at the store
    // phases: bit 0 = build the sweep geometry (coords kernel), bit 1 = run the slab kernel.  N reference views starting at view ref_first of the n_src packed source views (a view shard;
    // N == n_src and ref_first == 0 for a whole scene); nbr / proj / depth / var / scratch are indexed by the LOCAL view
    constexpr int TH = kTilePix / TW;
    const int S = num_slabs(C);
    const int tiles_x = (W + TW - 1) / TW, tiles_y = (H + TH - 1) / TH;
    const int tiles = tiles_x * tiles_y;
    const long long nblocks = (long long)N * tiles * S;
    if (nblocks > INT32_MAX) {
        set_error("plane_sweep_variance: grid too large");
        return MVSDET_ERR_INVALID_ARG;
    }
    const bool fast = (C % kSlab == 0) && (W % 4 == 0) && (Wo % 4 == 0);   // whole float4 stores of whole slabs (see the kernel)
    // A call that only CONSUMES a geometry (phases == 2) cannot know the capacity it was built with -- the staged /
    // refill flags and the union boxes carry it implicitly -- so its LDS slots are sized for the largest capacity any
    // geometry of this (K, tile shape) can have: "sweep_boxcap" may change between the two calls without harm.
    const int box_cap = (phases == 2) ? max_box_cap(K, TW) : effective_box_cap(K, TW);
    // every block sweeps all its planes (reference features stay in registers, a resident footprint box serves a run
    // of planes) unless the grid would be too small to fill 256 CUs x 2 blocks
    int dsplit = 1;
    while (nblocks * dsplit < 1024 && D / (dsplit * 2) >= 2) dsplit *= 2;
    // Few planes over the whole depth range (the shipped configs' 12 over 0.2-5 m): a third of a block's planes refill a
    // footprint box -- a block-wide stall -- and a CU's three blocks stall together; about four planes per block give the
    // CU more, shorter blocks in different phases (60x80 x 12 planes: 0.70 -> 0.66 ms, the store-pattern ceiling of the
    // 16x8 tiles: tools/micro/store_pattern_ref.hip; more planes per block or 32x4 tiles gain nothing: tools/sweep_pitch_timing.py)
    if (D <= 16 && TW == 16 && dsplit == 1) dsplit = (D + 3) / 4;
    if (options().sweep_dsplit > 0) dsplit = std::min(options().sweep_dsplit, D);
    const int d_per_block = (D + dsplit - 1) / dsplit;
    // plane groups (sweep_kernel.h): worth their extra block start-ups where refills are a large share of a block's planes
    // -- few planes spanning the whole depth range (the shipped configs' 12)
    int gmax = options().sweep_groups;
    if (gmax < 0) gmax = 1;   // measured (tools/sweep_pitch_timing.py): cutting at the refills loses to equal shares of planes
    gmax = std::min(gmax, kSweepGroups);
    const bool grouped = K > 0 && gmax >= 2 && dsplit == 1;
    const SweepGeometry geo = sweep_geometry(scratch, N, K, D, tiles);
    dim3 cgrid((unsigned)(N * tiles));
    const int n_bt = N * tiles;
    // option "sweep_xcd": 0 = slab = id % S; 1 = an XCD owns a slab (compacted for fewer than 8 slabs)
    int xcd_parts = 1;
    long long grid_x = nblocks;
    if (S < 8 && 8 % S == 0 && options().sweep_xcd != 0) {
        xcd_parts = 8 / S;
        grid_x = 8LL * ((n_bt + xcd_parts - 1) / xcd_parts);
    }
    // the geometry carries its own cuts (at most kSweepGroups groups, whatever "sweep_groups" said when it was built):
    // the grid allows for all of them, blocks of absent groups leave at once
    dim3 grid((unsigned)grid_x, grouped ? kSweepGroups : (D + d_per_block - 1) / d_per_block);
    const unsigned short* groups = grouped ? geo.groups : nullptr;
    const float* ref_packed = packed ? packed + (size_t)ref_first * S * H * W * kSlab : nullptr;
    const size_t lds = sweep_lds_bytes(K, box_cap);
    int rc = MVSDET_OK;
#define MVS_SLAB(KV, FV)                                                                                               \
    (half_out ? launch_slab<KV, TW, FV, __half>(grid, stream, lds, packed, ref_packed, nbr, geo, groups, var16, n_src, C, S, \
                                                D, H, W, Wo, tiles_x, tiles, d_per_block, box_cap, n_bt, xcd_parts)   \
              : launch_slab<KV, TW, FV, float>(grid, stream, lds, packed, ref_packed, nbr, geo, groups, var, n_src, C, S, D, \
                                               H, W, Wo, tiles_x, tiles, d_per_block, box_cap, n_bt, xcd_parts))
#define MVS_SWEEP_CASE(KV)                                                                                            \
    case KV:                                                                                                          \
        if (phases & 1)                                                                                               \
            hipLaunchKernelGGL((plane_sweep_coords_kernel<KV, TW>), cgrid, dim3(kThreads),                            \
                               (size_t)D * (KV * sizeof(int4) + sizeof(unsigned) + sizeof(float)), stream, proj, depth,               \
                               geo.header, geo.boxes, geo.flags, geo.proj, geo.depth, geo.groups, gmax, D, H, W, tiles_x, tiles, box_cap); \
        if (phases & 2) rc = fast ? MVS_SLAB(KV, true) : MVS_SLAB(KV, false);                                         \
        break;
    switch (K) {
        case 0:
            if (phases & 2) rc = fast ? MVS_SLAB(0, true) : MVS_SLAB(0, false);
            break;
        MVS_SWEEP_CASE(1)
        MVS_SWEEP_CASE(2)
        MVS_SWEEP_CASE(3)
        MVS_SWEEP_CASE(4)
    }
#undef MVS_SWEEP_CASE
#undef MVS_SLAB
    if (rc) return rc;
    MVS_LAUNCH_CHECK("plane_sweep_variance");
    return MVSDET_OK;
}
}  // namespace

namespace mvsdet {
int sweep_tile_width(int W, int D) { return pick_tile_width(W, D, 0); }  // shared with planesweep_bwd.hip
int sweep_box_cap(int K, int tw) { return effective_box_cap(K, tw); }
int sweep_max_box_cap(int K, int tw) { return max_box_cap(K, tw); }
}

// The tile shape and LDS box capacity the sweep uses for this problem (the layout of the geometry in the scratch buffer
// follows from them): for tools and statistics.
extern "C" int mvsdet_plane_sweep_tile_shape(int K, int D, int H, int W, int* tile_w, int* tile_h, int* box_texels) {
    MVS_REQUIRE(K >= 0 && D > 0 && H > 0 && W > 0, "plane_sweep_tile_shape: bad shape");
    const int tw = pick_tile_width(W, D);
    if (tile_w) *tile_w = tw;
    if (tile_h) *tile_h = kTilePix / tw;
    if (box_texels) *box_texels = effective_box_cap(K, tw);
    return MVSDET_OK;
}

extern "C" size_t mvsdet_plane_sweep_scratch_bytes(int N, int K, int D, int H, int W) {
    if (N <= 0 || K <= 0 || D <= 0 || H <= 0 || W <= 0) return 0;
    // footprint boxes (16 B per view, tile, plane, neighbour) + flags (4 B per view, tile, plane) + copies of the camera
    // data; sized for the larger of the two tile shapes so the "sweep_tw" option cannot outgrow it
    const size_t tiles = (size_t)std::max(num_tiles(H, W, 16), num_tiles(H, W, 32));
    return sizeof(int4) + (size_t)N * tiles * D * K * sizeof(int4) + align16((size_t)N * tiles * D * sizeof(unsigned)) +
           align16((size_t)N * K * 16 * sizeof(float)) + align16((size_t)N * D * sizeof(float)) +
           align16((size_t)N * tiles * (kSweepGroups + 1) * sizeof(unsigned short));
}

static int sweep_entry(const char* name, const float* packed, const int64_t* nbr, const float* proj, const float* depth,
                       void* var, void* scratch, size_t scratch_bytes, int N, int K, int C, int D, int H, int W,
                       mvsdet_stream_t stream, int phases, int n_src = -1, int ref_first = 0, bool half_out = false,
                       int out_pitch = 0) {
    if (n_src < 0) n_src = N;
    const int Wo = out_pitch > 0 ? out_pitch : W;
    MVS_REQUIRE(Wo >= W, "%s: output row pitch %d < W=%d", name, Wo, W);
    MVS_REQUIRE(ref_first >= 0 && N <= n_src && ref_first <= n_src - N,
                "%s: reference views [%d, %d) outside the %d packed views", name, ref_first, ref_first + N, n_src);
    MVS_REQUIRE(!(phases & 2) || (packed && var), "%s: NULL pointer", name);
    MVS_REQUIRE(K == 0 || scratch, "%s: NULL scratch with K=%d", name, K);
    MVS_REQUIRE(K == 0 || !(phases & 1) || (proj && depth), "%s: NULL proj / depth", name);
    MVS_REQUIRE(K == 0 || !(phases & 2) || nbr, "%s: NULL neighbour ids", name);
    MVS_REQUIRE(N > 0 && C > 0 && D > 0 && H > 1 && W > 1, "%s: bad shape N=%d C=%d D=%d H=%d W=%d", name, N, C, D, H, W);
    MVS_REQUIRE(K >= 0 && K <= MVSDET_MAX_NEIGHBORS, "%s: K=%d outside [0,%d]", name, K, MVSDET_MAX_NEIGHBORS);
    MVS_REQUIRE(D <= MVSDET_MAX_DEPTH && H < 65535 && W < 65535, "%s: D > %d, or H or W > 65534", name, MVSDET_MAX_DEPTH);
    MVS_REQUIRE((size_t)8 * D * H * Wo * sizeof(float) < ((size_t)1 << 32), "%s: 8 channel rows of the cost volume (8*D*H*W floats) exceed 4 GiB", name);
    MVS_REQUIRE((size_t)H * W * kSlab < (size_t)INT32_MAX, "%s: one slab image exceeds 2^31 elements", name);
    if (scratch_bytes < mvsdet_plane_sweep_scratch_bytes(N, K, D, H, W)) {
        set_error("%s: scratch %zu B < %zu B", name, scratch_bytes, mvsdet_plane_sweep_scratch_bytes(N, K, D, H, W));
        return MVSDET_ERR_WORKSPACE;
    }
    MVS_REQUIRE(K == 0 || ((uintptr_t)scratch % 16 == 0), "%s: scratch must be 16-byte aligned", name);
    const int tw = pick_tile_width(W, D, Wo);
    hipStream_t st = (hipStream_t)stream;
    if (tw == 16) return launch_sweep<16>(packed, nbr, proj, depth, var, scratch, N, K, C, D, H, W, st, phases, n_src, ref_first, half_out, Wo);
    return launch_sweep<32>(packed, nbr, proj, depth, var, scratch, N, K, C, D, H, W, st, phases, n_src, ref_first, half_out, Wo);
}

extern "C" int mvsdet_plane_sweep_variance_packed_f32(const float* packed, const int64_t* nbr, const float* proj,
                                                      const float* depth, float* var, void* scratch,
                                                      size_t scratch_bytes, int N, int K, int C, int D, int H, int W,
                                                      mvsdet_stream_t stream) {
    return sweep_entry("plane_sweep_variance", packed, nbr, proj, depth, var, scratch, scratch_bytes, N, K, C, D, H, W,
                       stream, 3);
}

extern "C" int mvsdet_plane_sweep_variance_shard_f32(const float* packed, const int64_t* nbr, const float* proj,
                                                     const float* depth, float* var, void* scratch,
                                                     size_t scratch_bytes, int N_src, int ref_first, int M, int K, int C,
                                                     int D, int H, int W, mvsdet_stream_t stream) {
    return sweep_entry("plane_sweep_variance_shard", packed, nbr, proj, depth, var, scratch, scratch_bytes, M, K, C, D, H,
                       W, stream, 3, N_src, ref_first);
}

extern "C" int mvsdet_plane_sweep_variance_shard_f16(const float* packed, const int64_t* nbr, const float* proj,
                                                     const float* depth, void* var_f16, void* scratch,
                                                     size_t scratch_bytes, int N_src, int ref_first, int M, int K, int C,
                                                     int D, int H, int W, mvsdet_stream_t stream) {
    return sweep_entry("plane_sweep_variance_shard_f16", packed, nbr, proj, depth, var_f16, scratch, scratch_bytes, M, K, C,
                       D, H, W, stream, 3, N_src, ref_first, true);
}

extern "C" int mvsdet_plane_sweep_table_f32(const float* proj, const float* depth, void* scratch, size_t scratch_bytes,
                                            int N, int K, int D, int H, int W, mvsdet_stream_t stream) {
    return sweep_entry("plane_sweep_table", nullptr, nullptr, proj, depth, nullptr, scratch, scratch_bytes, N, K, 1, D, H, W,
                       stream, 1);
}

extern "C" int mvsdet_plane_sweep_variance_tabled_f32(const float* packed, const int64_t* nbr, const void* table,
                                                      size_t table_bytes, float* var, int N, int K, int C, int D, int H,
                                                      int W, mvsdet_stream_t stream) {
    return sweep_entry("plane_sweep_variance_tabled", packed, nbr, nullptr, nullptr, var, const_cast<void*>(table),
                       table_bytes, N, K, C, D, H, W, stream, 2);
}

// The two halves again for a PITCHED output: var is (N,C,D,H,out_w_pitch) in memory, of which columns [0, W) are written
// (the caller hands out the view).  A pitch that is a multiple of 32 puts every row on a 128-byte line and selects the
// 32x4 tiles (pick_tile_width); the geometry must be built with the same pitch (the tile shape decides its layout).
extern "C" int mvsdet_plane_sweep_table_pitched_f32(const float* proj, const float* depth, void* scratch, size_t scratch_bytes,
                                                    int N, int K, int D, int H, int W, int out_w_pitch, mvsdet_stream_t stream) {
    return sweep_entry("plane_sweep_table_pitched", nullptr, nullptr, proj, depth, nullptr, scratch, scratch_bytes, N, K, 1, D, H,
                       W, stream, 1, -1, 0, false, out_w_pitch);
}

extern "C" int mvsdet_plane_sweep_variance_tabled_pitched_f32(const float* packed, const int64_t* nbr, const void* table,
                                                              size_t table_bytes, float* var, int N, int K, int C, int D, int H,
                                                              int W, int out_w_pitch, mvsdet_stream_t stream) {
    return sweep_entry("plane_sweep_variance_tabled_pitched", packed, nbr, nullptr, nullptr, var, const_cast<void*>(table),
                       table_bytes, N, K, C, D, H, W, stream, 2, -1, 0, false, out_w_pitch);
}

extern "C" int mvsdet_plane_sweep_variance_f32(const float* feat, const int64_t* nbr, const float* proj,
                                               const float* depth, float* var, void* workspace, size_t workspace_bytes,
                                               int N, int K, int C, int D, int H, int W, mvsdet_stream_t stream) {
    MVS_REQUIRE(feat && workspace, "plane_sweep_variance: NULL pointer");
    MVS_REQUIRE(N > 0 && C > 0 && D > 0 && H > 1 && W > 1, "plane_sweep_variance: bad shape");
    const size_t pb = (mvsdet_packed_bytes(N, C, H, W) + 255) / 256 * 256;
    const size_t sb = mvsdet_plane_sweep_scratch_bytes(N, K, D, H, W);
    if (workspace_bytes < pb + sb) {
        set_error("plane_sweep_variance: workspace %zu B < %zu B", workspace_bytes, pb + sb);
        return MVSDET_ERR_WORKSPACE;
    }
    const int64_t fs[4] = {(int64_t)C * H * W, (int64_t)H * W, W, 1};
    int rc = mvsdet_pack_features_f32(feat, fs, (float*)workspace, N, C, H, W, stream);
    if (rc) return rc;
    return mvsdet_plane_sweep_variance_packed_f32((const float*)workspace, nbr, proj, depth, var, (char*)workspace + pb, sb,
                                                  N, K, C, D, H, W, stream);
}

extern "C" size_t mvsdet_plane_sweep_workspace_bytes(int N, int K, int C, int D, int H, int W) {
    return (mvsdet_packed_bytes(N, C, H, W) + 255) / 256 * 256 + mvsdet_plane_sweep_scratch_bytes(N, K, D, H, W);
}
