// a3+a4 fused: plane-sweep variance (included by planesweep.hip and planesweep_bwd.hip).  Two kernels:
//
// (1) plane_sweep_coords_kernel -- the sweep GEOMETRY, everything that depends on (view, tile, neighbour, plane) but not
//     on the channel: per (tile, neighbour, plane) the bounding box of the bilinear taps of the tile's 128 pixels
//     (sampling positions of mvs_models/module.py:116-143), then RUNS: consecutive planes whose boxes have a UNION of at
//     most `box_cap` texels all carry that union box.  Beyond ~1.2 m the footprints of neighbouring planes move by less
//     than a texel per plane, so a 312-texel box serves ~13 planes on the ScanNet-like geometry and ~50 on the ARKit-like
//     one (tools/box_runs.py).  Output: boxes + one flags word per (tile, plane) (live / staged / refill per neighbour).
//     There is no per-pixel table: the slab kernels recompute the positions with the same code (sample_at).
//
// (2) plane_sweep_variance_kernel -- the channel work, one 32-channel slab per block:
//       block   = (reference view n, TWxTH pixel tile (128 pixels), slab s, depth chunk), 4 waves
//       LDS     = ONE resident footprint box per neighbour (K slots of box_cap 128-byte texels).  A slot is refilled
//                 (LDS-DMA, two block barriers) only when the run's union box changes: ~8 % of the live (tile, plane,
//                 neighbour) triples.  All other planes run without any block-level synchronisation: position -> decode
//                 -> taps from LDS -> variance -> stores, wave-private from end to end.
//       lanes   = (pixel slot ps = lane>>3, channel group g = lane&7); the lane owns the 4 CONSECUTIVE pixels
//                 4*ps .. 4*ps+3 of its wave's 32 (one per step) x 4 channels (8*i+g), so a variance value is stored
//                 straight from registers as 16 bytes per lane = 8 channel rows x 128 contiguous bytes per
//                 wave-instruction -- no LDS transpose.
//       decode  = in pass p lane (ps, g) computes the sampling position of pixel-step g&3 for neighbour 2p + (g>>2) --
//                 one position (4 IEEE divisions), one bilinear decode per lane and plane; the 8 lanes of the pixel slot
//                 fetch the tap offsets and weights with DPP (a bank-masked row shift for the neighbour's quad, once per
//                 pass, then one quad_perm broadcast per value and step) -- no LDS tables, no barrier.
//       FAST    = C % 32 == 0 and W % 4 == 0: four unconditional 16-byte stores under one lane predicate.
//
// Design history and what bounds the kernel: DESIGN.md 4.1.
//
// Arithmetic (device rounding, oracle mode 1): warped = fma chain over the 4 taps; S = f + w1 + ..;
// Q = fma(w,w,Q); var = fma(-m, m, Q*r) with m = S*r, r = 1/(K+1).
#pragma once
#include <type_traits>

#include "common.h"
#include "pack.h"

namespace mvsdet {

constexpr int kTilePix = 128;       // pixels per tile

// Wave-wide integer min / max of a footprint (xlo, xhi, ylo, yhi), all on DPP: butterfly inside each row of 16 lanes (every
// lane of a row ends up with the row's result), then row_bcast:15 folds row 0 into row 1 and row 2 into row 3, row_bcast:31
// folds rows 0-1 into row 3; lane 63 holds the wave's result, read into SGPRs.  The DPP operand sits on the min / max
// itself (v_min_i32_dpp; hipcc emits a v_mov_b32_dpp + the operation + an s_nop for the DPP read-after-write hazard), and
// the four independent reductions are interleaved step by step, which keeps dependent instructions three apart: 24 VALU
// instructions instead of 48 + 24 s_nop -- a sixth of the geometry kernel's plane loop.
__device__ __forceinline__ void wave_reduce_box(int& xlo, int& xhi, int& ylo, int& yhi) {
#define MVS_DPP_STEP(MOD)                                                                     \
    asm volatile("v_min_i32_dpp %0, %0, %0 " MOD "\n\t"                                       \
                 "v_max_i32_dpp %1, %1, %1 " MOD "\n\t"                                       \
                 "v_min_i32_dpp %2, %2, %2 " MOD "\n\t"                                       \
                 "v_max_i32_dpp %3, %3, %3 " MOD                                              \
                 : "+v"(xlo), "+v"(xhi), "+v"(ylo), "+v"(yhi));
    asm volatile("s_nop 1" ::: "memory");   // the values may have been written by the instruction just before
    MVS_DPP_STEP("quad_perm:[1,0,3,2] row_mask:0xf bank_mask:0xf")
    MVS_DPP_STEP("quad_perm:[2,3,0,1] row_mask:0xf bank_mask:0xf")
    MVS_DPP_STEP("row_half_mirror row_mask:0xf bank_mask:0xf")
    MVS_DPP_STEP("row_mirror row_mask:0xf bank_mask:0xf")
    MVS_DPP_STEP("row_bcast:15 row_mask:0xa bank_mask:0xf")   // -> rows 1 and 3
    MVS_DPP_STEP("row_bcast:31 row_mask:0xc bank_mask:0xf")   // -> rows 2 and 3
#undef MVS_DPP_STEP
    xlo = __builtin_amdgcn_readlane(xlo, 63);
    xhi = __builtin_amdgcn_readlane(xhi, 63);
    ylo = __builtin_amdgcn_readlane(ylo, 63);
    yhi = __builtin_amdgcn_readlane(yhi, 63);
}

// The un-normalised sample position (ix, iy) of module.py:116-143: everything bilinear
// sampling derives from it -- tap origin, fractional weights, which taps are inside -- is decode_sample(), shared
// by the coords kernel (footprint boxes), the slab kernel and the backward kernel, so all three agree bit for bit.
struct SampleTaps {
    int x0, y0;      // clamped tap origin (NaN / Inf positions become finite indices)
    float wx, wy;    // fractional weights
    bool x0in, x1in, y0in, y1in;
};

__device__ __forceinline__ SampleTaps decode_sample(float ix, float iy, int H, int W) {
    SampleTaps t;
    const float x0 = floorf(ix), y0 = floorf(iy);
    t.wx = ix - x0;
    t.wy = iy - y0;
    t.x0in = (x0 >= 0.0f) && (x0 <= (float)(W - 1));
    t.x1in = (x0 >= -1.0f) && (x0 <= (float)(W - 2));
    t.y0in = (y0 >= 0.0f) && (y0 <= (float)(H - 1));
    t.y1in = (y0 >= -1.0f) && (y0 <= (float)(H - 2));
    t.x0 = (int)fminf(fmaxf(x0, -1.0f), (float)(W - 1));
    t.y0 = (int)fminf(fmaxf(y0, -1.0f), (float)(H - 1));
    return t;
}

// the 4 bilinear weights; an outside tap carries weight*0, so Inf/NaN positions give NaN as ATen-CPU does
__device__ __forceinline__ float4 tap_weights(const SampleTaps& t) {
    const float ex = 1.0f - t.wx, sy = 1.0f - t.wy;
    const float wnw = sy * ex, wne = sy * t.wx, wsw = t.wy * ex, wse = t.wy * t.wx;
    float4 w;
    w.x = (t.x0in && t.y0in) ? wnw : wnw * 0.0f;
    w.y = (t.x1in && t.y0in) ? wne : wne * 0.0f;
    w.z = (t.x0in && t.y1in) ? wsw : wsw * 0.0f;
    w.w = (t.x1in && t.y1in) ? wse : wse * 0.0f;
    return w;
}

constexpr unsigned kFlagLive = 1u, kFlagStaged = 2u, kFlagRefill = 4u;  // per-neighbour nibble of a plane's flags word
constexpr int kBoxPad = 8;                // texels of slack per LDS slot (the bank swizzle may use slot index ntex)
constexpr int kBoxSkip = INT32_MIN;       // boxes[].w of an empty footprint whose positions are all finite
constexpr int kBoxEmpty = INT32_MIN + 1;  // empty footprint with a non-finite position: taps run and give NaN
constexpr float kNoSample = -2.0f;  // entry of a tile pixel outside the image: no tap inside, all weights +0

// module.py:116-143 in two steps, shared by the geometry kernel (footprint boxes, table for the backward pass) and the
// slab kernel (positions recomputed per plane), so both get the same bits: the plane-independent ray of a pixel,
// rot @ [x, y, 1], and the un-normalised sample position on the plane of depth d.
struct SampleRay { float rx, ry, rz; };
__device__ __forceinline__ SampleRay sample_ray(const float* __restrict__ P, float x, float y) {
    SampleRay r;
    r.rx = fmaf(P[1], y, P[0] * x) + P[2];
    r.ry = fmaf(P[5], y, P[4] * x) + P[6];
    r.rz = fmaf(P[9], y, P[8] * x) + P[10];
    return r;
}
__device__ __forceinline__ float2 sample_at(const SampleRay& r, float t0, float t1, float t2, float d, int H, int W) {
    const float X = r.rx * d + t0;
    const float Y = r.ry * d + t1;
    const float Z = r.rz * d + t2;
    const float px = X / Z;
    const float py = Y / Z;
    const float gx = px / ((float)(W - 1) * 0.5f) - 1.0f;
    const float gy = py / ((float)(H - 1) * 0.5f) - 1.0f;
    return make_float2(fmaf(gx + 1.0f, (float)W * 0.5f, -0.5f), fmaf(gy + 1.0f, (float)H * 0.5f, -0.5f));
}

// ---------------------------------------------------------------------------------------------
// (1) sweep geometry: block = (view, tile); thread = (neighbour, pixel); loops over all planes.
//     boxes [((n*tiles + tile)*D + d)*K + j] int4, flags [(n*tiles + tile)*D + d]; copies of proj / depth for the slab
//     kernels (forward and backward recompute the sample positions from them).
// ---------------------------------------------------------------------------------------------
__device__ __forceinline__ bool box_nonempty(const int4& b) { return b.y >= b.x && b.w >= b.z; }
__device__ __forceinline__ int box_area(const int4& b) { return (b.y - b.x + 1) * (b.w - b.z + 1); }

template <int K, int TW>
__global__ __launch_bounds__(kThreads) void plane_sweep_coords_kernel(const float* __restrict__ proj,
                                                                       const float* __restrict__ depth,
                                                                       int4* __restrict__ header, int4* __restrict__ boxes,
                                                                       unsigned* __restrict__ flags, float* __restrict__ proj_copy,
                                                                       float* __restrict__ depth_copy, unsigned short* __restrict__ groups,
                                                                       int gmax, int D, int H, int W, int tiles_x, int tiles, int box_cap) {
    constexpr int TH = kTilePix / TW;
    extern __shared__ int4 s_geo[];    // [K][D] the tile's boxes of all planes, then [D] flags words, then [D] plane depths
    int4* s_pb = s_geo;
    unsigned* s_fl = reinterpret_cast<unsigned*>(s_geo + (size_t)K * D);
    float* s_dv = reinterpret_cast<float*>(s_fl + D);
    const int bt = blockIdx.x;  // n*tiles + tile
    const int tile = bt % tiles, n = bt / tiles;
    const int tx0 = (tile % tiles_x) * TW, ty0 = (tile / tiles_x) * TH;
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    for (int d = tid; d < D; d += kThreads) {
        s_fl[d] = 0u;  // nibbles are OR-ed in below
        s_dv[d] = depth[(size_t)n * D + d];   // one global round trip for all planes instead of one per plane
    }
    __syncthreads();
    if (bt == 0 && tid == 0) *header = make_int4(kGeoMagic, TW | (box_cap << 8), W, (D << 8) | K);   // what this layout was built for
    if (tile == 0) {  // the slab kernel reads the camera data from the scratch buffer (the tabled entry point has no other)
        for (int i = tid; i < K * 16; i += kThreads) proj_copy[(size_t)n * K * 16 + i] = proj[(size_t)n * K * 16 + i];
        for (int d = tid; d < D; d += kThreads) depth_copy[(size_t)n * D + d] = depth[(size_t)n * D + d];
    }
    // ---- fast path: one THREAD per (neighbour, plane).  For a fixed plane the sample position is a linear-fractional function
    // of the pixel; where its denominator Z keeps one sign over the tile (checked at the four corner pixels: Z is affine in the
    // pixel), each coordinate takes its extremes over the tile at a corner.  So the footprint of all 128 pixels lies inside
    // the box spanned by the corners' positions -- 4 positions per (tile, plane, neighbour) instead of 128.  The box may
    // exceed the exact one (taps outside the image are clipped by the rectangle, not one by one) and carries a margin of
    // 1e-3 px against the rounding noise of positions computed at interior pixels (the chain is good to ~2e-4 px:
    // DESIGN "tolerance budget"); a larger box only costs LDS, results never depend on it.  Anything else -- Z changing
    // sign or vanishing, a non-finite position -- is left to the exact wave-wide scan below.
    constexpr int kSlowMark = INT32_MIN + 2;   // boxes[].w of a (plane, neighbour) that needs the exact scan
    {
        const int xa = tx0, xb = min(tx0 + TW, W) - 1, ya = ty0, yb = min(ty0 + TH, H) - 1;
        for (int task = tid; task < K * D; task += kThreads) {
            const int j = task / D, d = task - j * D;
            const float* P = proj + ((size_t)n * K + j) * 16;
            const float t0 = P[3], t1 = P[7], t2 = P[11];
            const float dval = s_dv[d];
            float ixmin = INFINITY, ixmax = -INFINITY, iymin = INFINITY, iymax = -INFINITY;
            bool ok = xb >= xa && yb >= ya;
            int zpos = 0, zneg = 0;
#pragma unroll
            for (int c = 0; c < 4; ++c) {
                const SampleRay r = sample_ray(P, (float)((c & 1) ? xb : xa), (float)((c & 2) ? yb : ya));
                const float Z = r.rz * dval + t2;                       // the denominator inside sample_at, same arithmetic
                const float2 e = sample_at(r, t0, t1, t2, dval, H, W);
                ok = ok && isfinite(Z) && isfinite(e.x) && isfinite(e.y);
                zpos += Z > 0.0f;
                zneg += Z < 0.0f;
                ixmin = fminf(ixmin, e.x); ixmax = fmaxf(ixmax, e.x);
                iymin = fminf(iymin, e.y); iymax = fmaxf(iymax, e.y);
            }
            ok = ok && (zpos == 4 || zneg == 4);
            int4 bx = make_int4(0, 0, 0, kSlowMark);
            if (ok) {
                // taps of a position ix are floor(ix), floor(ix) + 1; clipped to the image (and clamped before the conversion)
                const float fx0 = fmaxf(floorf(ixmin - 1e-3f), 0.0f), fx1 = fminf(floorf(ixmax + 1e-3f) + 1.0f, (float)(W - 1));
                const float fy0 = fmaxf(floorf(iymin - 1e-3f), 0.0f), fy1 = fminf(floorf(iymax + 1e-3f) + 1.0f, (float)(H - 1));
                if (fx1 < fx0 || fy1 < fy0) bx = make_int4(INT32_MAX, INT32_MIN, INT32_MAX, kBoxSkip);   // all taps outside, all finite
                else bx = make_int4((int)fx0, (int)fx1, (int)fy0, (int)fy1);
            }
            s_pb[(size_t)j * D + d] = bx;
        }
    }
    __syncthreads();
    // ---- exact scan of the marked (plane, neighbour) pairs.  A footprint is the work of ONE wave: its lanes take the tile's
    // pixels lane and lane + 64 (two independent positions in flight per lane), one wave reduction gives the box, lane 0
    // files it -- no block barrier and no cross-wave merge per plane.  Wave w takes the planes w, w + 4, ... of every neighbour.
    bool inside[2];
    SampleRay ray[K > 0 ? K : 1][2];   // the pixel rays (plane-independent)
#pragma unroll
    for (int h = 0; h < 2; ++h) {
        const int p = lane + 64 * h;
        const int x = tx0 + (p % TW), y = ty0 + (p / TW);
        inside[h] = (x < W) && (y < H);
#pragma unroll
        for (int j = 0; j < K; ++j) ray[j][h] = sample_ray(proj + ((size_t)n * K + j) * 16, (float)x, (float)y);
    }
#pragma unroll
    for (int j = 0; j < K; ++j) {
        const float* P = proj + ((size_t)n * K + j) * 16;
        const float t0 = P[3], t1 = P[7], t2 = P[11];
        for (int d = wave; d < D; d += kThreads / 64) {
            if (s_pb[(size_t)j * D + d].w != kSlowMark) continue;   // wave-uniform (LDS broadcast)
            const float dval = s_dv[d];
            int xlo = INT32_MAX, xhi = INT32_MIN, ylo = INT32_MAX, yhi = INT32_MIN;
            int fin = 1;
#pragma unroll
            for (int h = 0; h < 2; ++h) {
                if (inside[h]) {   // a pixel outside the image has no taps and no footprint
                    const float2 e = sample_at(ray[j][h], t0, t1, t2, dval, H, W);
                    fin &= (isfinite(e.x) && isfinite(e.y)) ? 1 : 0;
                    const SampleTaps t = decode_sample(e.x, e.y, H, W);
                    if ((t.x0in || t.x1in) && (t.y0in || t.y1in)) {  // bounding box of the taps that are inside
                        xlo = min(xlo, t.x0in ? t.x0 : t.x0 + 1);
                        xhi = max(xhi, t.x1in ? t.x0 + 1 : t.x0);
                        ylo = min(ylo, t.y0in ? t.y0 : t.y0 + 1);
                        yhi = max(yhi, t.y1in ? t.y0 + 1 : t.y0);
                    }
                }
            }
            wave_reduce_box(xlo, xhi, ylo, yhi);
            const bool all_finite = __ballot(fin == 0) == 0ull;
            if (lane == 0) {
                int4 bx = make_int4(xlo, xhi, ylo, yhi);
                // no tap of the whole tile is inside the source image: the warped values are exactly 0 and the sweep
                // skips this neighbour (kBoxSkip) -- unless a position is NaN / Inf, whose taps must still produce NaN
                if (bx.y < bx.x || bx.w < bx.z) bx = make_int4(INT32_MAX, INT32_MIN, INT32_MAX, all_finite ? kBoxSkip : kBoxEmpty);
                s_pb[(size_t)j * D + d] = bx;
            }
        }
    }
    // Runs: consecutive live planes of one neighbour whose union box holds at most box_cap texels all get that union
    // box (greedy; out-of-view planes in between do not break a run -- the resident box stays valid across them -- a
    // footprint larger than the cap does).  Wave j walks its neighbour's boxes, then publishes per plane its
    // nibble of the flags word: kFlagLive (taps run), kFlagStaged (from the LDS box), kFlagRefill (the box differs from
    // the one the previous staged plane of this neighbour left resident).  (Cutting the runs of all neighbours together,
    // so that their refills share one stall, was tried: fewer stalls, more box traffic, no net gain.)
    __syncthreads();  // the zeroed flags words, the last plane's boxes
    // Wave j walks neighbour j.  The walk is sequential (greedy), so the boxes of 64 planes at a time sit in the lanes'
    // registers and the walk reads them with v_readlane (a few cycles) instead of one LDS round trip per plane; closing a
    // run rewrites its planes' boxes 64 at a time.  (With one thread per neighbour reading LDS plane by plane this pass was
    // 60 % of the kernel at 128 planes.)
    if (wave < K) {
        int4* bj = s_pb + (size_t)wave * D;
        const unsigned sh = 4 * wave;
        auto lane_box = [&](const int4& v, int i) {
            return make_int4(__builtin_amdgcn_readlane(v.x, i), __builtin_amdgcn_readlane(v.y, i),
                             __builtin_amdgcn_readlane(v.z, i), __builtin_amdgcn_readlane(v.w, i));
        };
        // planes first .. last-1 take the union box u; their flags: staged, and the first one refills the slot unless the
        // previous run left the very same box resident (all planes of a run carry one box, so only its first can differ)
        auto close_run = [&](int first, int last, const int4& u, bool differs) {
            for (int e0 = first; e0 < last; e0 += 64) {
                const int e = e0 + lane;
                if (e < last) {
                    const int4 o = bj[e];
                    if (box_nonempty(o) && box_area(o) <= box_cap) {
                        bj[e] = u;
                        atomicOr(s_fl + e, (kFlagStaged | ((e == first && differs) ? kFlagRefill : 0u)) << sh);
                    }
                }
            }
        };
        int run_first = -1;
        int4 u = make_int4(0, 0, 0, 0), res = make_int4(0, -1, 0, -1);   // res: what the sweep holds resident in this slot
        for (int base = 0; base < D; base += 64) {
            const int4 mine = base + lane < D ? bj[base + lane] : make_int4(0, -1, 0, -1);
            const bool live = box_nonempty(mine);
            // live: the taps run; an empty footprint with a non-finite position too (from global memory, giving NaN)
            if (base + lane < D && (live || mine.w != kBoxSkip)) atomicOr(s_fl + base + lane, kFlagLive << sh);
            unsigned long long todo = box_cap > 0 ? __ballot(live) : 0ull;   // the walk only visits non-empty footprints
            while (todo) {
                const int i = __builtin_ctzll(todo);
                todo &= todo - 1;
                const int d = base + i;
                const int4 b = lane_box(mine, i);
                bool close = false;
                if (box_area(b) > box_cap) {
                    close = true;
                } else if (run_first >= 0) {
                    const int4 c = make_int4(min(u.x, b.x), max(u.y, b.y), min(u.z, b.z), max(u.w, b.w));
                    if (box_area(c) <= box_cap) { u = c; continue; }
                    close = true;
                }
                if (close && run_first >= 0) {
                    close_run(run_first, d, u, u.x != res.x || u.y != res.y || u.z != res.z || u.w != res.w);
                    res = u;
                    run_first = -1;
                }
                if (box_area(b) <= box_cap) { run_first = d; u = b; }
            }
        }
        if (run_first >= 0) close_run(run_first, D, u, u.x != res.x || u.y != res.y || u.z != res.z || u.w != res.w);
    }
    __syncthreads();
    for (int i = tid; i < D * K; i += kThreads) boxes[(size_t)bt * D * K + i] = s_pb[(size_t)(i % K) * D + i / K];
    for (int d = tid; d < D; d += kThreads) flags[(size_t)bt * D + d] = s_fl[d];
    // Plane groups: the slab kernel deals a tile's planes to up to gmax blocks, cut in front of a plane that refills a box.
    // The near planes' footprints jump by many texels per plane (12 planes over 0.2-5 m: every one of the first few is a
    // refill -- a block-wide stall of two barriers and a DMA round trip); as the first plane of its own block a refill is
    // part of the block's start-up, which the other blocks of the CU cover.  The last group takes all remaining planes.
    if (tid == 0 && groups) {
        unsigned short* gr = groups + (size_t)bt * (kSweepGroups + 1);
        int g = 0;
        gr[0] = 0;
        constexpr unsigned kAnyRefill = kFlagRefill | (kFlagRefill << 4) | (kFlagRefill << 8) | (kFlagRefill << 12);
        for (int d = 1; d < D; ++d)
            if ((s_fl[d] & kAnyRefill) && g < gmax - 1) gr[++g] = (unsigned short)d;
        for (++g; g <= kSweepGroups; ++g) gr[g] = (unsigned short)D;
    }
}

// DPP helpers.  A wave's lanes are (pixel slot ps = lane >> 3, channel group g = lane & 7): an 8-lane group = two quads.
// quad_bcast<S>:   every lane reads quad-lane S of its own quad.
// from_quad<Q>:    every lane of the 8-lane group reads the value its quad Q (0 = lanes 0-3, 1 = lanes 4-7) holds at the
//                  reader's quad-lane: the other quad's lanes take it over a row shift by 4 (bank-masked), the quad's own
//                  lanes keep theirs.
template <int S>
__device__ __forceinline__ int quad_bcast(int v) {
    return __builtin_amdgcn_update_dpp(0, v, S * 0x55, 0xf, 0xf, true);
}
// quad_bcast<S>(v) + add as ONE instruction (v_add_u32_dpp), for the four tap offsets of a pixel step at once.  A DPP operand
// fresh from the vector pipe needs two wait states, a hazard hipcc's recogniser cannot see inside inline assembly -- and a
// separate `s_nop` statement ahead of the additions does not bind the register allocator: it may still place a copy of an
// operand (a live-range split under the three-blocks-per-CU register budget) between the nop and the DPP read.  So the wait
// states and the four additions of a step are ONE asm block: whatever wrote the operands, copies included, lies before the
// nop.  Outputs are early-clobber: none may share a register with an input that a later line of the block still reads.
#define MVS_QBA(S) \
    "s_nop 1\n\t" \
    "v_add_u32_dpp %0, %4, %8 quad_perm:[" #S "," #S "," #S "," #S "] row_mask:0xf bank_mask:0xf bound_ctrl:1\n\t" \
    "v_add_u32_dpp %1, %5, %8 quad_perm:[" #S "," #S "," #S "," #S "] row_mask:0xf bank_mask:0xf bound_ctrl:1\n\t" \
    "v_add_u32_dpp %2, %6, %8 quad_perm:[" #S "," #S "," #S "," #S "] row_mask:0xf bank_mask:0xf bound_ctrl:1\n\t" \
    "v_add_u32_dpp %3, %7, %8 quad_perm:[" #S "," #S "," #S "," #S "] row_mask:0xf bank_mask:0xf bound_ctrl:1"
template <int S>
__device__ __forceinline__ void quad_bcast_add4(int v0, int v1, int v2, int v3, int add, int& r0, int& r1, int& r2, int& r3) {
#define MVS_QBA_IO : "=&v"(r0), "=&v"(r1), "=&v"(r2), "=&v"(r3) : "v"(v0), "v"(v1), "v"(v2), "v"(v3), "v"(add)
    if constexpr (S == 0) asm volatile(MVS_QBA(0) MVS_QBA_IO);
    else if constexpr (S == 1) asm volatile(MVS_QBA(1) MVS_QBA_IO);
    else if constexpr (S == 2) asm volatile(MVS_QBA(2) MVS_QBA_IO);
    else asm volatile(MVS_QBA(3) MVS_QBA_IO);
#undef MVS_QBA_IO
}
#undef MVS_QBA
// fmaf(-m, m, q) as the one IEEE instruction it is, spelled out so that hipcc does not pair two of them into a v_pk_fma_f32
// (whose results then need register copies to reach their places in the store vectors)
__device__ __forceinline__ float fma_neg_sq(float m, float q) {
    float r;
    asm("v_fma_f32 %0, -%1, %1, %2" : "=v"(r) : "v"(m), "v"(q));
    return r;
}
// variance of channels (2h, 2h+1)'s pixel s from the sum and the sum of squares over the views: the two products packed over
// the channel pair, the final fused multiply-add per value, written straight into its place in the lane's store vectors
typedef float sweep_f2 __attribute__((ext_vector_type(2)));
__device__ __forceinline__ void variance_of(float (&vout)[4][4], int s, int h, sweep_f2 sum, sweep_f2 sq, float rcp) {
    const sweep_f2 r2 = {rcp, rcp};
    const sweep_f2 m = sum * r2, q = sq * r2;
    vout[2 * h][s] = fma_neg_sq(m.x, q.x);
    vout[2 * h + 1][s] = fma_neg_sq(m.y, q.y);
}
// 16 bytes at an LDS ADDRESS (not an index into a __shared__ array: the address arrives ready-made from quad_bcast_add4)
__device__ __forceinline__ float4 lds_f4_at(int addr) {
    typedef float v4 __attribute__((ext_vector_type(4)));
    const v4 t = *(const __attribute__((address_space(3))) v4*)(size_t)(unsigned)addr;
    return make_float4(t.x, t.y, t.z, t.w);
}
template <int Q>
__device__ __forceinline__ int from_quad(int v) {
    return Q == 0 ? __builtin_amdgcn_update_dpp(v, v, 0x114 /* row_shr:4 */, 0xf, 0xA, false)
                  : __builtin_amdgcn_update_dpp(v, v, 0x104 /* row_shl:4 */, 0xf, 0x5, false);
}

typedef float f2 __attribute__((ext_vector_type(2)));  // v_pk_fma_f32 / v_pk_mul_f32 / v_pk_add_f32 operands
__device__ __forceinline__ f2 pk_fma(f2 a, f2 b, f2 c) { return __builtin_elementwise_fma(a, b, c); }
__device__ __forceinline__ f2 splat(float v) { f2 r = {v, v}; return r; }

// LDS bank swizzle of a staged box: box texel t (row-major) is kept in texel slot t ^ bit2(t).  A ds_read_b128 serves 16 lanes = two
// pixel slots = two 128-byte texels per clock: conflict-free when the two lie in different halves of the 64 banks.  Pixel slots are
// four pixels apart, their texels 3-5 apart: flipping bit 0 with bit 2 makes texels 4 apart alternate halves
// (SQ_LDS_BANK_CONFLICT / SQ_LDS_IDX_ACTIVE 0.42 -> 0.33; five other bit choices: 0.33-0.44, profiles/r06_sweep_swizzle_ab.txt).
// An involution; the largest slot index of an n-texel box is n (hence kBoxPad).
// The BACKWARD kernel keeps it (its LDS atomics pay for conflicts).  The FORWARD kernel does not (round 6): its LDS pipe is not what
// bounds it, and the two vector operations per tap address cost more than the conflicts -- 1.0-1.2 % on the 32 x 4-tile shapes
// (10.69 -> 10.59 ms at the headline, 29.9 -> 29.5 at the stress shape; no difference on the 16 x 8 tiles), the same bits.
__device__ __forceinline__ int box_slot(int t) { return t ^ ((t >> 2) & 1); }
__device__ __forceinline__ int fwd_box_slot(int t) { return t; }

__host__ __device__ constexpr size_t sweep_lds_bytes(int K, int box_cap) { return (size_t)K * (box_cap + kBoxPad) * 128; }

// OutT = float, or __half: the variance is computed in fp32 exactly as before and rounded to nearest-even at the
// store (BASELINE configs[4], fp16 storage), which halves the dominant write stream.
// FAST: every channel row of the slab exists (C % 32 == 0) and every lane's 4 pixels are all inside or all outside the
// image with 16-byte aligned rows (W % 4 == 0): the stores are four unconditional vector stores under one lane predicate.
template <int K, int TW, bool FAST, typename OutT = float>
__global__ __launch_bounds__(kThreads, (TW == 16 && K <= 2) ? 3 : 2) void plane_sweep_variance_kernel(   // 16x8 tiles: three 52-KiB blocks per CU (planesweep.hip: max_box_cap)
    
    const float* __restrict__ packed, const float* __restrict__ ref_packed, const int64_t* __restrict__ nbr,
    const int4* __restrict__ header, const float* __restrict__ proj, const float* __restrict__ depth, const int4* __restrict__ boxes,
    const unsigned* __restrict__ flags, const unsigned short* __restrict__ groups, OutT* __restrict__ var, int N, int C, int S,
    int D, int H, int W, int Wo, int tiles_x, int tiles, int d_per_block, int box_cap, int n_bt, int xcd_parts) {
    constexpr int KK = K > 0 ? K : 1;
    constexpr bool kTapsAhead = TW == 32 && K <= 2;   // all four steps' taps of a neighbour requested ahead of their arithmetic (two blocks per CU: registers to spare)
    constexpr int NP = (K + 1) / 2;             // decode passes: a lane decodes ONE (pixel-step, neighbour) pair per pass
    constexpr int NPP = NP > 0 ? NP : 1;
    constexpr int TH = kTilePix / TW;
    extern __shared__ float4 s_box[];  // K slots of (box_cap + kBoxPad) texels (8 float4 each)

    if constexpr (K > 0) {
        // The geometry's layout hangs on the tile shape it was built for: one that was built for another (a pitched table handed
        // to the contiguous call or the other way round, another "sweep_tw" or "sweep_boxcap") holds boxes of other tiles, boxes
        // larger than this launch's LDS slots -- or nothing at all --
        // where this kernel would look.  Nothing of it is touched: the whole output becomes NaN instead (block-uniform exit
        // before any barrier).
        const int4 hd = *header;
        if (hd.x != kGeoMagic || (hd.y & 0xff) != TW || (hd.y >> 8) > box_cap || hd.z != W || hd.w != ((D << 8) | K)) {
            const size_t total = (size_t)(n_bt / tiles) * C * D * H * Wo;
            const size_t step = (size_t)gridDim.x * gridDim.y * kThreads;
            for (size_t i = ((size_t)blockIdx.y * gridDim.x + blockIdx.x) * kThreads + threadIdx.x; i < total; i += step)
                var[i] = (OutT)__builtin_nanf("");
            return;
        }
    }
    const int HW = H * W;
    const int HWo = H * Wo;   // Wo = row pitch of the OUTPUT in elements (W for a contiguous volume; a multiple of 32 puts every
                              // row on a 128-byte boundary: 32x4 tiles then write whole lines even when W is no multiple of 32)
    const int id = blockIdx.x;
    int slab = id % S;
    int bt = id / S;  // n*tiles + tile
    if (xcd_parts > 1) {
        // fewer than 8 slabs (C < 256): the 8/S XCDs that share a slab each take one contiguous range of
        // (view, tile) pairs, so an XCD's L2 sees a compact set of source rows instead of every 8th tile's
        const int xcd = id & 7, k = id >> 3;
        const int per_part = (n_bt + xcd_parts - 1) / xcd_parts;
        slab = xcd % S;
        bt = (xcd / S) * per_part + k;
        if (k >= per_part || bt >= n_bt) return;  // padding blocks of the rounded-up grid (whole block, before any barrier)
    }
    const int tile = bt % tiles, n = bt / tiles;
    const int tx0 = (tile % tiles_x) * TW, ty0 = (tile / tiles_x) * TH;
    // the block's planes: a fixed share of D, or group blockIdx.y of this tile's plane groups (plane_sweep_coords_kernel)
    int d_begin = blockIdx.y * d_per_block;
    int d_end = min(D, d_begin + d_per_block);
    if (groups) {
        d_begin = groups[(size_t)bt * (kSweepGroups + 1) + blockIdx.y];
        d_end = groups[(size_t)bt * (kSweepGroups + 1) + blockIdx.y + 1];
        if (d_begin >= d_end) return;   // fewer groups than the grid allows for (block-uniform, before any barrier)
    }

    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);  // wave-uniform by construction: keep it scalar
    const int sub = wave;                                       // pixel quarter of the tile
    const int g = lane & 7, ps = lane >> 3;
    const size_t slab_stride = (size_t)HW * kSlab;       // floats per (view, slab) image
    // ref_packed = packed + first reference view of this launch (a view shard); N bounds the NEIGHBOUR ids
    const float* ref_img = ref_packed + ((size_t)n * S + slab) * slab_stride;
    const float4* nb_img[KK];
#pragma unroll
    for (int j = 0; j < K; ++j) {
        int64_t v = nbr[(size_t)n * K + j];
        v = v < 0 ? 0 : (v >= N ? N - 1 : v);  // never read outside the packed maps (ids are validated on the host)
        nb_img[j] = reinterpret_cast<const float4*>(packed + ((size_t)v * S + slab) * slab_stride);
    }
    const float rcp = 1.0f / (float)(K + 1);
    const int slot_f4 = (box_cap + kBoxPad) * 8;  // float4 per LDS slot
    const int g16 = g * 16;                       // the lane's 16 bytes of a texel
    const int g16_lds = g16 + (int)(unsigned)(size_t)(__attribute__((address_space(3))) char*)s_box;   // ... as an LDS address

    // the lane's 4 consecutive pixels: tile-local p0 .. p0+3 (4 | TW, so they share a row)
    const int p0 = 32 * sub + 4 * ps;
    const int px0 = tx0 + p0 % TW, py = ty0 + p0 / TW;
    f2 f[4][2];  // loop invariant: their reference features (channels 8*i + g; [s][0] = i 0,1; [s][1] = i 2,3)
#pragma unroll
    for (int s = 0; s < 4; ++s) {
        const int pix = min(py, H - 1) * W + min(px0 + s, W - 1);
        const float4 v = *reinterpret_cast<const float4*>(ref_img + (size_t)pix * kSlab + 4 * g);
        f[s][0] = (f2){v.x, v.y};
        f[s][1] = (f2){v.z, v.w};
    }
    f2 fsq[4][2];  // their squares: the reference view's term of the sum of squares
#pragma unroll
    for (int s = 0; s < 4; ++s) { fsq[s][0] = f[s][0] * f[s][0]; fsq[s][1] = f[s][1] * f[s][1]; }
    // stores: uniform base of (channel 8*i of the slab, plane d) + one 32-bit lane offset (channel g, the pixels)
    // (bytes; the entry point checks that 8 channel rows of the volume stay below 4 GiB)
    const unsigned st_off = ((unsigned)g * (unsigned)D * (unsigned)HWo + (unsigned)(py * Wo + px0)) * (unsigned)sizeof(OutT);
    const int st_n = (py < H) ? max(0, min(4, W - px0)) : 0;               // how many of the 4 pixels are inside the image
    const bool st_vec = (st_n == 4) && ((W & 3) == 0) && ((Wo & 3) == 0);  // 16-byte aligned in every channel row

    // ---- the lane's decode duty.  The sampling position of (pixel, plane, neighbour) does not depend on the channel:
    // in pass p lane (ps, g) computes it for pixel p0 + (g & 3) and neighbour 2p + (g >> 2) -- ONE position, one decode
    // per lane, pass and plane -- and the 8 lanes of the pixel slot fetch the results with DPP (quad broadcast for the
    // step, row shift by 4 for the neighbour's quad).  No sampling table: the positions are recomputed from the pixel's
    // ray (plane-independent, kept in registers) with the very arithmetic of the geometry kernel (sample_at), 4 IEEE
    // divisions per position; this costs ~15 % more VALU work than decoding table entries, and saves the table's
    // 8 B per (pixel, plane, neighbour) -- read by every slab, 12 % of the sweep's traffic, and worse than its share
    // in time because the reads interleave with the write stream (measured 2.3 of 13.6 ms).
    const int sd = g & 3, qd = g >> 2;
    const int dx = px0 + sd;                            // the pixel this lane decodes (same row as its own)
    const bool d_inside = (dx < W) && (py < H);
    SampleRay ray[NPP];
    float tr0[NPP], tr1[NPP], tr2[NPP];                 // translation column of the lane's neighbour
#pragma unroll
    for (int p = 0; p < NP; ++p) {
        const int jd = min(2 * p + qd, K - 1);
        const float* P = proj + ((size_t)n * K + jd) * 16;
        ray[p] = sample_ray(P, (float)dx, (float)py);
        tr0[p] = P[3]; tr1[p] = P[7]; tr2[p] = P[11];
    }
    // the box resident in the slot of the lane's neighbour (per pass), and block-uniform copies per neighbour
    int lx0[NPP], lx1[NPP], ly0[NPP], ly1[NPP];
    int rx0[KK], ry0[KK], rx1[KK], ry1[KK];
    bool have[KK];
#pragma unroll
    for (int p = 0; p < NPP; ++p) { lx0[p] = 0; lx1[p] = 0; ly0[p] = 0; ly1[p] = 0; }
#pragma unroll
    for (int j = 0; j < KK; ++j) { rx0[j] = 0; ry0[j] = 0; rx1[j] = -1; ry1[j] = -1; have[j] = false; }

    // LDS-DMA of one footprint box into its slot: each wave-instruction fills 8 texel slots = 1 KiB without touching
    // VGPRs (LDS address = wave-uniform base + 16*lane, global address per lane: the texel that belongs in the lane's
    // slot).  Pieces are dealt round-robin to the block's waves.
    auto load_box = [&](int j, int bx0, int by0, int nc, int nr) {
        const int ntex = nc * nr;
        const float inv_nc = 1.0f / (float)nc;
        for (int q = wave; q * 8 <= ntex; q += 4) {
            const int t = fwd_box_slot(q * 8 + ps);             // the box texel kept in slot q*8 + ps
            const int row = (int)(((float)t + 0.5f) * inv_nc);  // t / nc for t < 2^11 (never within rounding of an integer)
            const int col = t - row * nc;
            const float4* src = nb_img[j] + ((size_t)(by0 + row) * W + (bx0 + col)) * 8 + g;
            float4* dst = s_box + (size_t)j * slot_f4 + q * 64;  // wave-uniform
            if (t < ntex)
                __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)src,
                                                 (__attribute__((address_space(3))) void*)dst, 16, 0, 0);
        }
    };

    const unsigned* fl_bt = flags + (size_t)bt * D;
    const float* depth_n = depth + (size_t)n * D;
    unsigned fl_next = (K > 0 && d_begin < d_end) ? fl_bt[d_begin] : 0u;
    float dv_next = (K > 0 && d_begin < d_end) ? depth_n[d_begin] : 0.0f;

    // results of the last computed plane, and the stores that send them out: straight from registers, 16 bytes per lane =
    // 8 channel rows x 128 contiguous bytes per wave-instruction; non-temporal (written once, never re-read here: keeps the
    // stream from evicting the source slabs)
    float vout[4][4];
    int d_pending = -1;
    OutT* const var_slab = var + ((size_t)n * C + slab * kSlab) * D * HWo;  // block-uniform: channel row 0 of the slab, plane 0
    const size_t row8 = (size_t)8 * D * HWo;                                // 8 channel rows further
    auto flush = [&](int d) {
        OutT* plane_base = var_slab + (size_t)d * HWo;
        if constexpr (FAST) {
            if (st_n == 4) {
#pragma unroll
                for (int i = 0; i < 4; ++i) {
                    const float* v = vout[i];
                    OutT* dst = reinterpret_cast<OutT*>(reinterpret_cast<char*>(plane_base + i * row8) + st_off);
                    if constexpr (sizeof(OutT) == 4) {
                        typedef float v4f __attribute__((ext_vector_type(4)));
                        const v4f vv = {v[0], v[1], v[2], v[3]};
                        __builtin_nontemporal_store(vv, reinterpret_cast<v4f*>(dst));
                    } else {
                        const __half h[4] = {__float2half_rn(v[0]), __float2half_rn(v[1]), __float2half_rn(v[2]), __float2half_rn(v[3])};
                        typedef unsigned v2u __attribute__((ext_vector_type(2)));
                        const v2u vv = {(unsigned)__half_as_ushort(h[0]) | ((unsigned)__half_as_ushort(h[1]) << 16),
                                        (unsigned)__half_as_ushort(h[2]) | ((unsigned)__half_as_ushort(h[3]) << 16)};
                        __builtin_nontemporal_store(vv, reinterpret_cast<v2u*>(dst));
                    }
                }
            }
        } else {
#pragma unroll
            for (int i = 0; i < 4; ++i) {
                const float* v = vout[i];
                const int c = slab * kSlab + g + 8 * i;
                if (c < C) {
                    OutT* dst = reinterpret_cast<OutT*>(reinterpret_cast<char*>(plane_base + i * row8) + st_off);
                    if constexpr (sizeof(OutT) == 4) {
                        if (st_vec) {
                            typedef float v4f __attribute__((ext_vector_type(4)));
                            const v4f vv = {v[0], v[1], v[2], v[3]};
                            __builtin_nontemporal_store(vv, reinterpret_cast<v4f*>(dst));
                        } else {
#pragma unroll
                            for (int s = 0; s < 4; ++s)
                                if (s < st_n) dst[s] = v[s];
                        }
                    } else {
                        const __half h[4] = {__float2half_rn(v[0]), __float2half_rn(v[1]), __float2half_rn(v[2]), __float2half_rn(v[3])};
                        if (st_vec) {  // 4 pixels x 2 B: the fp32 alignment condition also gives 8-byte alignment
                            typedef unsigned v2u __attribute__((ext_vector_type(2)));
                            const v2u vv = {(unsigned)__half_as_ushort(h[0]) | ((unsigned)__half_as_ushort(h[1]) << 16),
                                            (unsigned)__half_as_ushort(h[2]) | ((unsigned)__half_as_ushort(h[3]) << 16)};
                            __builtin_nontemporal_store(vv, reinterpret_cast<v2u*>(dst));
                        } else {
#pragma unroll
                            for (int s = 0; s < 4; ++s)
                                if (s < st_n) dst[s] = h[s];
                        }
                    }
                }
            }
        }
    };

    for (int d = d_begin; d < d_end; ++d) {
        // ---- this plane's flags and depth (scalars, requested a plane ahead); every wave follows every plane so that all
        //      of them take the same refill decisions, whichever planes they compute
        const unsigned fl = __builtin_amdgcn_readfirstlane(fl_next);
        const float dval = __int_as_float(__builtin_amdgcn_readfirstlane(__float_as_int(dv_next)));
        if (K > 0 && d + 1 < d_end) { fl_next = fl_bt[d + 1]; dv_next = depth_n[d + 1]; }
        bool refill = false;
#pragma unroll
        for (int j = 0; j < K; ++j)
            if ((fl >> (4 * j)) & kFlagStaged)
                if (((fl >> (4 * j)) & kFlagRefill) || !have[j]) refill = true;
        if (refill) {
            __syncthreads();  // every wave is done with the planes that read the old boxes
#pragma unroll
            for (int j = 0; j < K; ++j) {
                if (((fl >> (4 * j)) & kFlagStaged) && (((fl >> (4 * j)) & kFlagRefill) || !have[j])) {
                    const int4 b = boxes[((size_t)bt * D + d) * K + j];
                    rx0[j] = __builtin_amdgcn_readfirstlane(b.x);
                    rx1[j] = __builtin_amdgcn_readfirstlane(b.y);
                    ry0[j] = __builtin_amdgcn_readfirstlane(b.z);
                    ry1[j] = __builtin_amdgcn_readfirstlane(b.w);
                    have[j] = true;
                    if (min(2 * (j / 2) + qd, K - 1) == j) { lx0[j / 2] = rx0[j]; lx1[j / 2] = rx1[j]; ly0[j / 2] = ry0[j]; ly1[j / 2] = ry1[j]; }
                    load_box(j, rx0[j], ry0[j], rx1[j] - rx0[j] + 1, ry1[j] - ry0[j] + 1);
                }
            }
            // The LDS-DMA pieces of this wave count on vmcnt, and for a workgroup barrier hipcc only waits for lgkmcnt:
            // without the explicit wait a wave can pass the barrier while its own pieces are still in flight and the
            // other waves read stale box texels.  (vmcnt is the one in-order counter of loads AND stores: the wait also
            // drains this wave's stores -- the reason the compute loop below contains no other load.)
            __builtin_amdgcn_s_waitcnt(0);
            __syncthreads();
        }
        if (d_pending >= 0) flush(d_pending);

        // sum and sum of squares over the views, starting from the reference view's own term.  Where neighbour 0 is visible (the
        // usual plane) its tap steps write f + v and f*f + v*v directly -- the same additions, without 16 register copies a plane
        f2 S_[4][2], Q_[4][2];
        const bool first_live = K > 0 && (fl & kFlagLive);   // scalar
        if (!first_live) {
#pragma unroll
            for (int s = 0; s < 4; ++s) {
                S_[s][0] = f[s][0]; S_[s][1] = f[s][1];
                Q_[s][0] = fsq[s][0]; Q_[s][1] = fsq[s][1];
            }
        }
        // A pixel step = the 4 taps of the lane's pixel SS of neighbour j: LOAD fetches them (tap address = quad broadcast of the decoding
        // lane's offset + the lane's own 16 bytes), MATH weighs and sums them into the step's sums.  32x4 tiles run two blocks per CU and
        // have ~80 registers to spare: there all four steps' taps of a neighbour are requested before the first step's arithmetic
        // (16 float4 in flight: one exposed LDS round trip per neighbour instead of four).  16x8 tiles (three blocks per CU, 170
        // registers) keep load - use - load - use.
#define MVS_TAP_LOAD(SS, LOADER, OFF)                                                                                 \
        {                                                                                                             \
            int o0, o1, o2, o3;                                                                                       \
            OFF(SS, o0, o1, o2, o3)                                                                                   \
            tq[SS][0] = LOADER(o0); tq[SS][1] = LOADER(o1); tq[SS][2] = LOADER(o2); tq[SS][3] = LOADER(o3);           \
        }
#define MVS_TAP_MATH(SS)                                                                                              \
        {                                                                                                             \
            const f2 w0 = splat(__int_as_float(quad_bcast<SS>(rw0))), w1 = splat(__int_as_float(quad_bcast<SS>(rw1)));  \
            const f2 w2 = splat(__int_as_float(quad_bcast<SS>(rw2))), w3 = splat(__int_as_float(quad_bcast<SS>(rw3)));  \
            const float4 t0 = tq[SS][0], t1 = tq[SS][1], t2 = tq[SS][2], t3 = tq[SS][3];                              \
            f2 va = (f2){t0.x, t0.y} * w0, vb = (f2){t0.z, t0.w} * w0;                                                \
            va = pk_fma((f2){t1.x, t1.y}, w1, va); vb = pk_fma((f2){t1.z, t1.w}, w1, vb);                             \
            va = pk_fma((f2){t2.x, t2.y}, w2, va); vb = pk_fma((f2){t2.z, t2.w}, w2, vb);                             \
            va = pk_fma((f2){t3.x, t3.y}, w3, va); vb = pk_fma((f2){t3.z, t3.w}, w3, vb);                             \
            const f2 sa = (j == 0 ? f[SS][0] : S_[SS][0]) + va, sb = (j == 0 ? f[SS][1] : S_[SS][1]) + vb;             \
            const f2 qa = pk_fma(va, va, j == 0 ? fsq[SS][0] : Q_[SS][0]);                                             \
            const f2 qb = pk_fma(vb, vb, j == 0 ? fsq[SS][1] : Q_[SS][1]);                                             \
            if constexpr (j == K - 1) {   /* the last neighbour: its sums are complete -- straight on to the variance */ \
                variance_of(vout, SS, 0, sa, qa, rcp);                                                                \
                variance_of(vout, SS, 1, sb, qb, rcp);                                                                \
            } else {                                                                                                  \
                S_[SS][0] = sa; S_[SS][1] = sb; Q_[SS][0] = qa; Q_[SS][1] = qb;                                       \
            }                                                                                                         \
        }
#define MVS_TAP_STEPS(LOADER, OFF)                                                                                    \
        {                                                                                                             \
            float4 tq[4][4];                                                                                          \
            if constexpr (kTapsAhead) {                                                                               \
                MVS_TAP_LOAD(0, LOADER, OFF) MVS_TAP_LOAD(1, LOADER, OFF) MVS_TAP_LOAD(2, LOADER, OFF) MVS_TAP_LOAD(3, LOADER, OFF) \
                __builtin_amdgcn_sched_barrier(0);   /* the scheduler sinks the requests back to their uses otherwise */ \
                MVS_TAP_MATH(0) MVS_TAP_MATH(1) MVS_TAP_MATH(2) MVS_TAP_MATH(3)                                       \
            } else {                                                                                                  \
                MVS_TAP_LOAD(0, LOADER, OFF) MVS_TAP_MATH(0) MVS_TAP_LOAD(1, LOADER, OFF) MVS_TAP_MATH(1)             \
                MVS_TAP_LOAD(2, LOADER, OFF) MVS_TAP_MATH(2) MVS_TAP_LOAD(3, LOADER, OFF) MVS_TAP_MATH(3)             \
            }                                                                                                         \
        }
        // Tap offsets travel in BYTES, so that the quad broadcast and the addition of the lane's own 16 bytes of the texel are ONE
        // v_add_u32_dpp (an index needs a v_mov_b32_dpp and a v_lshl_add_u32 -- a VOP3, which takes no DPP operand).  hipcc folds
        // only half of them by itself: the LDS path spells the instruction out (quad_bcast_add4).
#define MVS_LDS_TAP(O) lds_f4_at(O)   /* O holds the LDS address itself */
#define MVS_GLB_TAP(O) (*reinterpret_cast<const float4*>(reinterpret_cast<const char*>(nb_img[j]) + (O)))
#define MVS_LDS_OFF(SS, A, B, C_, D_) quad_bcast_add4<SS>(ro0, ro1, ro2, ro3, g16_lds, A, B, C_, D_);
#define MVS_GLB_OFF(SS, A, B, C_, D_) A = quad_bcast<SS>(ro0) + g16; B = quad_bcast<SS>(ro1) + g16; C_ = quad_bcast<SS>(ro2) + g16; D_ = quad_bcast<SS>(ro3) + g16;
#define MVS_TAPS_OF(QQ)                                                                                               \
        {                                                                                                             \
            if constexpr (2 * p + QQ < K) {                                                                           \
                constexpr int j = 2 * p + QQ;                                                                         \
                const unsigned fj = fl >> (4 * j);                                                                    \
                if (fj & kFlagLive) {                                                                                 \
                    /* quad QQ of every 8-lane group decoded this neighbour: bring its 8 values to both quads once, */ \
                    /* then each step is one quad broadcast per value                                               */ \
                    const int ro0 = from_quad<QQ>(do0), ro1 = from_quad<QQ>(do1), ro2 = from_quad<QQ>(do2), ro3 = from_quad<QQ>(do3); \
                    const int rw0 = from_quad<QQ>(__float_as_int(dw.x)), rw1 = from_quad<QQ>(__float_as_int(dw.y));   \
                    const int rw2 = from_quad<QQ>(__float_as_int(dw.z)), rw3 = from_quad<QQ>(__float_as_int(dw.w));   \
                    if (fj & kFlagStaged) {                                                                           \
                        MVS_TAP_STEPS(MVS_LDS_TAP, MVS_LDS_OFF)                                                       \
                    } else {                                                                                          \
                        MVS_TAP_STEPS(MVS_GLB_TAP, MVS_GLB_OFF)                                                       \
                    }                                                                                                 \
                }                                                                                                     \
            }                                                                                                         \
        }
        // one pass = the neighbours 2p and 2p+1: every lane decodes its (pixel-step, neighbour) pair, then the taps of both
        auto pass = [&](auto pc) {
            constexpr int p = decltype(pc)::value;
            const unsigned fp = fl >> (8 * p);
            if (!(fp & (kFlagLive | (kFlagLive << 4)))) return;  // nothing of these neighbours is visible: warped values all zero
            const unsigned fq = qd ? ((2 * p + 1 < K) ? (fp >> 4) : fp) : fp;   // the nibble of the lane's neighbour
            const bool l_staged = (fq & kFlagStaged) != 0;
            float2 e = make_float2(kNoSample, kNoSample);  // pixel outside the image: no tap inside, all weights +0
            if (d_inside) e = sample_at(ray[p], tr0[p], tr1[p], tr2[p], dval, H, W);
            const SampleTaps tp = decode_sample(e.x, e.y, H, W);
            const float4 dw = tap_weights(tp);
            // tap offsets: inside the resident box of the lane's neighbour (slot index with the bank swizzle), or inside
            // its slab image for a gathered footprint (too large for the box, or empty with non-finite positions)
            const int lox = l_staged ? lx0[p] : 0, hix = l_staged ? lx1[p] : W - 1;
            const int loy = l_staged ? ly0[p] : 0, hiy = l_staged ? ly1[p] : H - 1;
            const int pitch = hix - lox + 1;
            const int xa = clampi(tp.x0, lox, hix) - lox, xb = clampi(tp.x0 + 1, lox, hix) - lox;
            const int ya = (clampi(tp.y0, loy, hiy) - loy) * pitch, yb = (clampi(tp.y0 + 1, loy, hiy) - loy) * pitch;
            const int sbase = min(2 * p + qd, K - 1) * slot_f4 * 16;
            const int do0 = l_staged ? fwd_box_slot(ya + xa) * 128 + sbase : (ya + xa) * 128;   // bytes: a texel of a slab is 128
            const int do1 = l_staged ? fwd_box_slot(ya + xb) * 128 + sbase : (ya + xb) * 128;
            const int do2 = l_staged ? fwd_box_slot(yb + xa) * 128 + sbase : (yb + xa) * 128;
            const int do3 = l_staged ? fwd_box_slot(yb + xb) * 128 + sbase : (yb + xb) * 128;
            MVS_TAPS_OF(0)
            MVS_TAPS_OF(1)
        };
        if constexpr (NP > 0) pass(std::integral_constant<int, 0>{});
        if constexpr (NP > 1) pass(std::integral_constant<int, 1>{});
#undef MVS_TAPS_OF
#undef MVS_TAP_STEPS
#undef MVS_TAP_MATH
#undef MVS_TAP_LOAD
#undef MVS_LDS_TAP
#undef MVS_GLB_TAP
#undef MVS_LDS_OFF
#undef MVS_GLB_OFF
        // ---- variance (channel 8*i + g, the lane's 4 consecutive pixels): kept in registers, stored at the top of the
        //      next iteration (after a possible box refill)
        // where the last neighbour was visible its tap steps have done it already (the usual plane: no merge of register sets)
        if (!(K > 0 && ((fl >> (4 * (KK - 1))) & kFlagLive))) {
#pragma unroll
            for (int s = 0; s < 4; ++s) {
                variance_of(vout, s, 0, S_[s][0], Q_[s][0], rcp);
                variance_of(vout, s, 1, S_[s][1], Q_[s][1], rcp);
            }
        }
        d_pending = d;
    }
    if (d_pending >= 0) flush(d_pending);
}

}  // namespace mvsdet
