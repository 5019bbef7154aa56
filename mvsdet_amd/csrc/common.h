// Shared helpers of the mvsdet_hip library (gfx950 only).
#pragma once
#include <hip/hip_runtime.h>

#include <cstdarg>
#include <cstdint>
#include <cstdio>

#include "../../include/mvsdet_hip.h"

namespace mvsdet {

void set_error(const char* fmt, ...);

// Tuning options of the library: initialised ONCE from the environment (MVSDET_SWEEP_TW, _BOXCAP, _XCD) the
// first time they are needed, afterwards changed only through mvsdet_set_option() -- no getenv on the launch path.
struct Options {
    int sweep_tw;      // 0 = by the map width, 16 / 32 force the tile shape (16x8 / 32x4)
    int sweep_boxcap;  // texels of one LDS footprint box (0 forces the global-gather path)
    int sweep_xcd;     // XCD-aware block map for fewer than 8 slabs
    int sweep_dsplit;  // 0 = by the grid size; n > 0: every block sweeps ceil(D / n) consecutive planes
    int sweep_groups;  // -1 = by the plane count; 0 / 1 = off; g in [2, kSweepGroups]: a tile's planes are dealt to up to g blocks,
                       // cut where its footprint boxes are refilled (the near planes)
    int conv_subpairs; // bf16x3 stride-1 convolution on 3x16x8 tiles: tap pairs per weight sub-stage; 0 = by the grid, 2 / 5 force
    int conv_nsplit;   // bf16x3 convolutions: splits of the input channels; 0 = by the grid, n > 0 force (needs the workspace)
    int conv_cgn;      // stride-1 bf16x3 convolution (16x16x32 form) on 3x16x8 tiles: 0 / 2 = 12 waves of 32 voxels, 4 = 6 waves of 64
    int conv_s2_cg;    // stride-2 bf16x3 convolution fed by PSCL on 3x16x8 tiles: 0 / 1 = 12 waves of one column group, 2 = 6 of two
    int conv_s2_ob;    // stride-2 bf16x3 convolution fed by PSCL: 1 = 64 output channels per block, else 128 where Cout allows
    int convT_cg;      // transposed bf16x3 convolution on 3x16x8 tiles: 0 / 3 = all eight output parity classes in one block of 32 output
                       // channels (12 waves); 1 = one block per (PD, PH), 12 waves of one column group; 2 = 6 waves of two
    int probe_f16_pair; // mvsdet_store_pattern_probe_f16 only: 1 = lanes of adjacent pixel quads own the octet between them and store
                       // 16 bytes of two channel rows instead of 8 of four (what a paired flush of the fp16 sweep WOULD reach: +12 %,
                       // not built -- the fp16 sweep is bound by its vector work); 0 (default) = the kernel's own pattern
    int conv_xcd;      // stride-1 bf16x3 convolution: 1 (default) = an XCD takes a contiguous eighth of the grid (neighbouring tiles share
                       // their halo voxels in one L2); 0 = blocks round-robin over the XCDs
    int conv_split_blocks; // bf16x3 convolutions on small grids: the input channels are split until about this many blocks run (768 =
                       // three rounds of 256 CUs)
    int conv_split_min_groups; // ... and a split keeps at least this many groups of 8 input channels (default 2: neck + head 2.39 -> 2.35 ms against 4)
    int bwd_groups;    // backward sweep: wave groups of a block that share its gradient images and split its planes; 0 = by the plane
                       // count (2 from 32 planes), 1 / 2 force
    int conv_mx_th;    // fp16 + MX convolution (costreg_mx.h): 0 (default) = wave-specialised kernel (8 multiplying + 4 staging waves, 4 x 8 x 16 tiles);
                       // 8 / 12 = every wave does everything on 4 x 8 x 16 (8 waves) / 4 x 12 x 16 (12 waves) tiles
    int conv_mfma16;   // 1 (default): the bf16x3 stride-1 convolution on v_mfma_f32_16x16x32_bf16 (conv0 of the cost network 5.37 -> 4.91 ms:
                       // the chip holds a higher clock on this shape); 0: 32x32x16.  Weights must be split under the same setting.
    int convT_persist; // transposed bf16x3 convolution, all-classes form with a skip tensor: 0 (default) = one block per (tile, 32 channels);
                       // 1 = the persistent kernel (convt_persist.h: one block per CU, an item's stores behind the next item's
                       // multiplications; the same bits, not faster: profiles/r06_convt_persist.txt); n >= 8: persistent on n blocks
};
Options& options();

// where plane_sweep_coords_kernel leaves the sweep geometry inside the scratch buffer (planesweep.hip)
struct SweepGeometry {
    int4* header;   // {kGeoMagic, tile width | (box capacity << 8), W, (D << 8) | K} written by plane_sweep_coords_kernel: the layout below hangs on the
                    // tile shape, so a consumer whose own choice differs (a pitched table fed to the contiguous call, another
                    // "sweep_tw") must not touch it -- the slab kernel checks the header and fills its output with NaN instead
    int4* boxes;
    unsigned* flags;
    float* proj;
    float* depth;
    unsigned short* groups;   // [N*tiles][kSweepGroups + 1] plane-group boundaries of every tile (sweep_kernel.h)
};
constexpr int kGeoMagic = 0x4d565347;   // "MVSG"
constexpr int kSweepGroups = 6;   // at most this many plane groups (blocks) per (tile, slab)
SweepGeometry sweep_geometry(void* scratch, int N, int K, int D, int tiles);

constexpr int kWave = 64;       // CDNA wavefront
constexpr int kThreads = 256;   // 4 waves, one per SIMD
constexpr int kXcds = 8;        // MI355X: 8 XCDs, blocks are dealt round-robin over them

#define MVS_REQUIRE(cond, ...)              \
    do {                                    \
        if (!(cond)) {                      \
            ::mvsdet::set_error(__VA_ARGS__); \
            return MVSDET_ERR_INVALID_ARG;  \
        }                                   \
    } while (0)

#define MVS_LAUNCH_CHECK(name)                                                        \
    do {                                                                              \
        hipError_t e_ = hipGetLastError();                                            \
        if (e_ != hipSuccess) {                                                       \
            ::mvsdet::set_error("%s: launch failed: %s", name, hipGetErrorString(e_)); \
            return MVSDET_ERR_HIP;                                                    \
        }                                                                             \
    } while (0)

// Blocks b and b+8 share an XCD (and its 4 MiB L2).  Map the hardware block id to a logical id so
// that every XCD walks ONE contiguous range of logical ids: neighbouring tiles of a view then find
// the source rows they share in their own L2.  Bijective for any nblocks (speed only, never
// correctness: MI355X_MICROARCH "Workgroup dispatch").
__device__ __forceinline__ int xcd_contiguous_id(int b, int nblocks) {
    const int q = nblocks / kXcds, r = nblocks % kXcds;
    const int xcd = b % kXcds, idx = b / kXcds;
    return (xcd < r) ? xcd * (q + 1) + idx : r * (q + 1) + (xcd - r) * q + idx;
}

// ---------------------------------------------------------------------------------------------
// Sampling taps of one (reference pixel, depth plane, source view): mvs_models/module.py:116-143 +
// ATen grid_sample(bilinear, zeros, align_corners=False).  Same rounding points as
// oracle/planesweep_oracle.c:orc_compute_taps (the library is compiled with -ffp-contract=off, so
// a fused multiply-add happens exactly where fmaf() is written).
//   P      proj = src_proj @ inverse(ref_proj), row-major 4x4
//   scale  element stride of one pixel in the sampled map (1 for NCHW planes, 4*G for packed maps)
//   off    element offsets of taps nw, ne, sw, se (0 for a tap outside the map)
//   w      bilinear weights (an outside tap carries weight*0, so Inf/NaN positions give NaN as ATen-CPU)
// ---------------------------------------------------------------------------------------------
__device__ __forceinline__ void compute_taps(const float* __restrict__ P, float x, float y, float d, int H, int W,
                                             int scale, int4& off, float4& w) {
    const float rx = fmaf(P[1], y, P[0] * x) + P[2];
    const float ry = fmaf(P[5], y, P[4] * x) + P[6];
    const float rz = fmaf(P[9], y, P[8] * x) + P[10];
    const float X = rx * d + P[3];
    const float Y = ry * d + P[7];
    const float Z = rz * d + P[11];
    const float px = X / Z;
    const float py = Y / Z;
    const float gx = px / ((float)(W - 1) * 0.5f) - 1.0f;
    const float gy = py / ((float)(H - 1) * 0.5f) - 1.0f;
    const float ix = fmaf(gx + 1.0f, (float)W * 0.5f, -0.5f);
    const float iy = fmaf(gy + 1.0f, (float)H * 0.5f, -0.5f);
    const float x0 = floorf(ix), y0 = floorf(iy);
    const float wx = ix - x0, wy = iy - y0;
    const float ex = 1.0f - wx, sy = 1.0f - wy;
    const bool x0in = (x0 >= 0.0f) && (x0 <= (float)(W - 1));
    const bool x1in = (x0 >= -1.0f) && (x0 <= (float)(W - 2));
    const bool y0in = (y0 >= 0.0f) && (y0 <= (float)(H - 1));
    const bool y1in = (y0 >= -1.0f) && (y0 <= (float)(H - 2));
    const int xi = (x0in || x1in) ? (int)x0 : 0;
    const int yi = (y0in || y1in) ? (int)y0 : 0;
    const float wnw = sy * ex, wne = sy * wx, wsw = wy * ex, wse = wy * wx;
    const int base = yi * W + xi;
    off.x = (x0in && y0in) ? base * scale : 0;
    off.y = (x1in && y0in) ? (base + 1) * scale : 0;
    off.z = (x0in && y1in) ? (base + W) * scale : 0;
    off.w = (x1in && y1in) ? (base + W + 1) * scale : 0;
    w.x = (x0in && y0in) ? wnw : wnw * 0.0f;
    w.y = (x1in && y0in) ? wne : wne * 0.0f;
    w.z = (x0in && y1in) ? wsw : wsw * 0.0f;
    w.w = (x1in && y1in) ? wse : wse * 0.0f;
}

__device__ __forceinline__ int clampi(int v, int lo, int hi) { return min(max(v, lo), hi); }

// Voxel -> pixel projection of backproject_Weigh (mvsdet.py:1383-1391): q = P @ [p;1] summed in k
// order with fused multiply-adds (oracle: orc_project), x = round_half_even(q0/q2), valid test.
__device__ __forceinline__ bool project_voxel(const float* __restrict__ P, float px, float py, float pz, int h, int w,
                                              float& xr, float& yr, float& z) {
    const float q0 = fmaf(P[2], pz, fmaf(P[1], py, P[0] * px)) + P[3];
    const float q1 = fmaf(P[6], pz, fmaf(P[5], py, P[4] * px)) + P[7];
    const float q2 = fmaf(P[10], pz, fmaf(P[9], py, P[8] * px)) + P[11];
    xr = rintf(q0 / q2);
    yr = rintf(q1 / q2);
    z = q2;
    return (xr >= 0.0f) && (yr >= 0.0f) && (xr < (float)w) && (yr < (float)h) && (q2 > 0.0f);
}

// Depth window + weight of one (view, voxel) pair: mvsdet.py:1393-1428 (oracle: orc_depth_window).
// depth/dens point at view i; candidate j of pixel (yi,xi) is at [j*s1 + yi*s2 + xi*s3].
// Returns valid' and the weight max_j(m_j ? prob_norm_j : 0); *arg = index of the first maximal
// matching candidate (-1 if none has positive weight) for the backward pass.
__device__ __forceinline__ bool depth_window(const float* __restrict__ depth, const float* __restrict__ dens,
                                             int64_t s1, int64_t s2, int64_t s3, int J, int yi, int xi, float z,
                                             float vz, float& weight, float& psum_out, int& arg) {
    const int64_t base = (int64_t)yi * s2 + (int64_t)xi * s3;
    float psum = 0.0f;
    for (int j = 0; j < J; ++j) psum = psum + dens[base + j * s1];
    float wmax = 0.0f;
    bool any = false;
    int a = -1;
    for (int j = 0; j < J; ++j) {
        const float dj = depth[base + j * s1];
        const bool m = (z > dj - vz) && (z < dj + vz);
        const float pn = dens[base + j * s1] / psum;
        const float cand = m ? pn : 0.0f;
        if (cand > wmax) {
            wmax = cand;
            a = j;
        }
        if (cand != cand) wmax = cand;
        any = any || m;
    }
    weight = wmax;
    psum_out = psum;
    arg = a;
    return any;
}

}  // namespace mvsdet
