// a3+a4 fused: plane-sweep variance (included by planesweep.hip).  Two kernels:
//
// (1) plane_sweep_coords_kernel -- everything that depends on (view, neighbour, plane, pixel) but NOT on the
//     channel: the sampling position of mvs_models/module.py:116-143, reduced to an 8-byte table entry
//     (the un-normalised sample position ix, iy) plus, per (tile, neighbour, plane), the bounding box of the
//     valid taps.  The reference builds its sampling grid once per plane too.
//     Cost: N*K*D*H*W entries (0.8 GB at the 64-plane shape, 1.5 % of the cost volume), written once.
//
// (2) plane_sweep_variance_kernel -- the channel work, one 32-channel slab per block:
//       block   = (reference view n, TWxTH pixel tile (128 pixels), slab s, depth chunk)
//                 logical id = (n*tiles + tile)*S + s: consecutive blocks are the S slabs of one tile and the
//                 round-robin block->XCD dispatch sends slab s of every tile to the same XCD when S == 8, so
//                 an XCD only ever touches ONE 128-byte slab of each source texel and its live working set
//                 fits its 4 MiB L2 (rocprofv3: 2*FETCH_SIZE + WRITE_SIZE == algorithmic bytes).
//       lanes   = (pixel slot ps = lane>>3, channel group g = lane&7); a wave-instruction covers 8 pixels x
//                 128 B; each wave owns 32 pixels = 4 steps; the reference features of those pixels stay in
//                 registers across the depth loop.
//       per depth plane
//         P1  scalar-load the footprint boxes; a neighbour with NO tap of the whole tile inside its image is dropped
//             for this plane (its warped values are exactly 0; a third of all (tile, plane, neighbour) triples at
//             ScanNet-like geometry); start the LDS-DMA of the first live neighbour's box; one thread per (pixel,
//             live neighbour) decodes its table entry into 4 weights + 4 tap offsets inside the box (or inside the
//             slab image when the footprint does not fit in LDS and that neighbour gathers from global memory)
//         per live neighbour j
//           P2  LDS-DMA of the box rows (contiguous nc*128-byte runs of the slab image), 1 KiB per wave-instruction;
//               explicit s_waitcnt 0 before the barrier that publishes the box (hipcc waits for lgkmcnt only)
//           P3  taps = 4 x ds_read_b128 per step; fma chain -> S, Q
//         P4  variance -> LDS tile [32 channels][128 pixels] (aliases the box storage)
//         P5  tile rows -> global as 16-byte non-temporal stores, 128 B contiguous per (channel, tile row); each wave
//             stores the pixels it computed (wave-private transpose, no block barrier between P4 and P5)
//
// Why this shape: v1 (all channels per block, taps gathered from global memory) saturated the fabric at a 43 %
// L2 hit rate (profiles/r01_v1_*); with slabs the kernel became VALU-issue bound (82 % busy), so everything
// that is not per-channel arithmetic was moved out of the per-slab loop (the table), made scalar (row bases,
// boxes) or turned into immediates (LDS addresses).
//
// Arithmetic (device rounding, oracle mode 1): warped = fma chain over the 4 taps; S = f + w1 + ..;
// Q = fma(w,w,Q); var = fma(-m, m, Q*r) with m = S*r, r = 1/(K+1).
#pragma once
#include "common.h"
#include "pack.h"

namespace mvsdet {

constexpr int kTilePix = 128;       // pixels per tile
constexpr int kBoxCap = 256;        // texels (128 B each) of the LDS footprint box: 32 KiB
constexpr int kTileStride = 132;    // floats per channel row of the output tile (132 % 32 == 4: conflict-free writes)

// Wave-wide integer min / max: butterfly inside each row of 16 lanes with DPP (4 VALU), then the four row
// results are combined on the scalar unit.  The result is wave-uniform (an SGPR).
template <bool kMin>
__device__ __forceinline__ int wave_reduce(int v) {
#define MVS_DPP_STEP(ctrl)                                                         \
    {                                                                              \
        const int o = __builtin_amdgcn_update_dpp(v, v, ctrl, 0xf, 0xf, false);    \
        v = kMin ? min(v, o) : max(v, o);                                          \
    }
    MVS_DPP_STEP(0xB1)   // quad_perm [1,0,3,2]
    MVS_DPP_STEP(0x4E)   // quad_perm [2,3,0,1]
    MVS_DPP_STEP(0x141)  // row_half_mirror
    MVS_DPP_STEP(0x140)  // row_mirror
#undef MVS_DPP_STEP
    const int a = __builtin_amdgcn_readlane(v, 0), b = __builtin_amdgcn_readlane(v, 16);
    const int c = __builtin_amdgcn_readlane(v, 32), d = __builtin_amdgcn_readlane(v, 48);
    return kMin ? min(min(a, b), min(c, d)) : max(max(a, b), max(c, d));
}

// Table entry = the un-normalised sample position (ix, iy) of module.py:116-143 (8 bytes); everything bilinear
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

constexpr int kBoxSkip = INT32_MIN;       // boxes[].w of an empty footprint whose positions are all finite
constexpr int kBoxEmpty = INT32_MIN + 1;  // empty footprint with a non-finite position: taps run and give NaN
constexpr float kNoSample = -2.0f;  // entry of a tile pixel outside the image: no tap inside, all weights +0

__device__ __forceinline__ float2 sample_position(const float* __restrict__ P, float x, float y, float d, int H, int W) {
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
    return make_float2(fmaf(gx + 1.0f, (float)W * 0.5f, -0.5f), fmaf(gy + 1.0f, (float)H * 0.5f, -0.5f));
}

// ---------------------------------------------------------------------------------------------
// (1) sampling table: block = (view, tile); thread = (neighbour, pixel); loops over the planes of its chunk.
//     table [((n*tiles + tile)*D + d)*K + j][128] float2;  boxes [((n*tiles + tile)*D + d)*K + j] int4.
// ---------------------------------------------------------------------------------------------
template <int K, int TW>
__global__ __launch_bounds__(kThreads) void plane_sweep_coords_kernel(const float* __restrict__ proj,
                                                                       const float* __restrict__ depth,
                                                                       float2* __restrict__ table, int4* __restrict__ boxes,
                                                                       int D, int H, int W, int tiles_x, int tiles,
                                                                       int d_per_block) {
    constexpr int TH = kTilePix / TW;
    constexpr int ITER = (K * kTilePix + kThreads - 1) / kThreads;
    __shared__ int s_red[2][K][2][5];  // [plane parity][neighbour][wave of the neighbour][xlo,xhi,ylo,yhi,all finite]
    const int bt = blockIdx.x;  // n*tiles + tile
    const int tile = bt % tiles, n = bt / tiles;
    const int tx0 = (tile % tiles_x) * TW, ty0 = (tile / tiles_x) * TH;
    const int tid = threadIdx.x, lane = tid & 63;
    const int p = tid % kTilePix;
    const int x = tx0 + (p % TW), y = ty0 + (p / TW);
    const bool inside = (x < W) && (y < H);
    const int d_begin = blockIdx.y * d_per_block, d_end = min(D, d_begin + d_per_block);
    for (int d = d_begin; d < d_end; ++d) {
        const int par = (d - d_begin) & 1;
        const float dval = depth[(size_t)n * D + d];
        const size_t base = ((size_t)bt * D + d) * K;
#pragma unroll
        for (int it = 0; it < ITER; ++it) {
            // 128 pixels = 2 whole waves per neighbour: j is wave-uniform
            const int j = __builtin_amdgcn_readfirstlane((it * kThreads + tid) / kTilePix);
            if (j < K) {
                int xlo = INT32_MAX, xhi = INT32_MIN, ylo = INT32_MAX, yhi = INT32_MIN;
                float2 e = make_float2(kNoSample, kNoSample);  // pixel outside the image: no taps, no footprint
                int fin = 1;
                if (inside) {
                    e = sample_position(proj + ((size_t)n * K + j) * 16, (float)x, (float)y, dval, H, W);
                    fin = (isfinite(e.x) && isfinite(e.y)) ? 1 : 0;
                    const SampleTaps t = decode_sample(e.x, e.y, H, W);
                    if ((t.x0in || t.x1in) && (t.y0in || t.y1in)) {  // bounding box of the taps that are inside
                        xlo = t.x0in ? t.x0 : t.x0 + 1;
                        xhi = t.x1in ? t.x0 + 1 : t.x0;
                        ylo = t.y0in ? t.y0 : t.y0 + 1;
                        yhi = t.y1in ? t.y0 + 1 : t.y0;
                    }
                }
                table[(base + j) * kTilePix + p] = e;
                xlo = wave_reduce<true>(xlo);
                xhi = wave_reduce<false>(xhi);
                ylo = wave_reduce<true>(ylo);
                yhi = wave_reduce<false>(yhi);
                fin = wave_reduce<true>(fin);
                if (lane == 0) {
                    int* r = s_red[par][j][(tid >> 6) & 1];
                    r[0] = xlo; r[1] = xhi; r[2] = ylo; r[3] = yhi; r[4] = fin;
                }
            }
        }
        __syncthreads();  // one barrier per plane: s_red is double-buffered by plane parity
        if (tid < K) {
            const int* a = s_red[par][tid][0];
            const int* b = s_red[par][tid][1];
            int4 bx = make_int4(min(a[0], b[0]), max(a[1], b[1]), min(a[2], b[2]), max(a[3], b[3]));
            // no tap of the whole tile is inside the source image: the warped values are exactly 0 and the sweep
            // skips this neighbour (kBoxSkip) -- unless a position is NaN / Inf, whose taps must still produce NaN
            if (bx.y < bx.x || bx.w < bx.z) bx = make_int4(INT32_MAX, INT32_MIN, INT32_MAX, min(a[4], b[4]) ? kBoxSkip : kBoxEmpty);
            boxes[base + tid] = bx;
        }
    }
}

// ---------------------------------------------------------------------------------------------
// (2) the slab kernel
// ---------------------------------------------------------------------------------------------
// OutT = float, or __half: the variance is computed in fp32 exactly as before and rounded to nearest-even at the
// store (BASELINE configs[4], fp16 storage), which halves the dominant write stream.
template <int K, int TW, bool NT, bool STAMP, typename OutT = float>
__global__ __launch_bounds__(kThreads, 4) void plane_sweep_variance_kernel(
    const float* __restrict__ packed, const float* __restrict__ ref_packed, const int64_t* __restrict__ nbr,
    const float2* __restrict__ table, const int4* __restrict__ boxes, OutT* __restrict__ var, int N, int C, int S, int D, int H, int W, int tiles_x,
    int tiles, int d_per_block, int box_cap, unsigned long long* __restrict__ stamps, int n_bt, int xcd_parts) {
    constexpr int KK = K > 0 ? K : 1;
    constexpr int TH = kTilePix / TW;
    constexpr int ITER = (KK * kTilePix + kThreads - 1) / kThreads;
    static_assert(kBoxCap * 8 >= 32 * kTileStride / 4, "output tile must fit in the box storage");
    __shared__ float4 s_box[kBoxCap * 8];       // footprint box of one neighbour; later the output tile
    __shared__ int4 s_off[KK][kTilePix];        // float4 index of the 4 taps (inside s_box or the slab image)
    __shared__ float4 s_w[KK][kTilePix];        // tap weights

    const int HW = H * W;
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
    const int d_begin = blockIdx.y * d_per_block;
    const int d_end = min(D, d_begin + d_per_block);

    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);  // wave-uniform by construction: keep it scalar
    const int g = lane & 7, ps = lane >> 3;
    const size_t slab_stride = (size_t)HW * kSlab;       // floats per (view, slab) image
    // ref_packed = packed + first reference view of this launch (a view shard); N bounds the NEIGHBOUR ids
    const float* ref_img = ref_packed + ((size_t)n * S + slab) * slab_stride;
    const float4* nb_img[KK];
#pragma unroll
    for (int j = 0; j < K; ++j) {
        int64_t v = nbr[(size_t)n * K + j];
        v = v < 0 ? 0 : (v >= N ? N - 1 : v);  // never read outside the packed maps
        nb_img[j] = reinterpret_cast<const float4*>(packed + ((size_t)v * S + slab) * slab_stride);
    }
    const float rcp = 1.0f / (float)(K + 1);

    // loop invariants of this lane: reference features of its 4 pixels (one per step) ...
    float4 f[4];
#pragma unroll
    for (int s = 0; s < 4; ++s) {
        const int p = (wave * 4 + s) * 8 + ps;
        const int x = tx0 + (p % TW), y = ty0 + (p / TW);
        const int pix = min(y, H - 1) * W + min(x, W - 1);
        f[s] = *reinterpret_cast<const float4*>(ref_img + (size_t)pix * kSlab + 4 * g);
    }
    // ... and the 4 consecutive output pixels it stores in P5.  A wave stores what it computed itself: its 32
    // pixels (tile columns 32*wave ..) of all 32 channel rows, so the tile transpose is wave-private and needs no
    // block barrier.  Lane -> channel row 8*k + (lane >> 3), float4 slot (lane & 7) of the wave's 32 pixels.
    const int sq = lane & 7, sh = lane >> 3;
    const int st_p = wave * 32 + 4 * sq;  // first of the 4 pixels, tile-local
    const int st_x = tx0 + st_p % TW, st_y = ty0 + st_p / TW;
    const int st_off = st_y * W + st_x;
    const int st_n = (st_y < H) ? max(0, min(4, W - st_x)) : 0;        // how many of the 4 pixels are inside the image
    const bool st_vec = (st_n == 4) && ((W & 3) == 0) && ((HW & 3) == 0);  // 16-byte aligned in every channel row

    auto load_box = [&](int j, int bx0, int by0, int nc, int nr) {
        // LDS-DMA: each wave-instruction moves 1 KiB global -> LDS without touching VGPRs; the LDS address is the
        // wave-uniform base + 16*lane, the global address is per lane.  The DMA counts on vmcnt, which the next
        // __syncthreads() drains.  Every wave copies whole rows: one contiguous nc*128-byte run each.
        const int row_f4 = nc * 8;
        const int cpr = (row_f4 + 63) >> 6;  // 1-KiB chunks per row; units (row, chunk) are dealt round-robin to the waves
        int row = 0, chunk = wave;
        while (chunk >= cpr) { chunk -= cpr; ++row; }
        while (row < nr) {
            const float4* src = nb_img[j] + ((size_t)(by0 + row) * W + bx0) * 8 + chunk * 64;  // wave-uniform
            float4* dst = s_box + row * row_f4 + chunk * 64;
            if (chunk * 64 + lane < row_f4)
                __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)(src + lane),
                                                 (__attribute__((address_space(3))) void*)dst, 16, 0, 0);
            chunk += 4;
            while (chunk >= cpr) { chunk -= cpr; ++row; }
        }
    };

    // Boxes and table entries are requested one plane ahead (during the taps of the previous plane), so the loop
    // never stalls on them; they are issued BEFORE that plane's result stores, so waiting for them does not wait
    // for store acknowledgements either (vmcnt retires in order).
    int4 bn[KK];
    float2 en[ITER];
#pragma unroll
    for (int j = 0; j < KK; ++j) bn[j] = make_int4(INT32_MAX, INT32_MIN, INT32_MAX, INT32_MIN);
#pragma unroll
    for (int it = 0; it < ITER; ++it) en[it] = make_float2(kNoSample, kNoSample);
    auto prefetch = [&](int d) {
#pragma unroll
        for (int j = 0; j < K; ++j) bn[j] = boxes[((size_t)bt * D + d) * K + j];
#pragma unroll
        for (int it = 0; it < ITER; ++it) {
            const int j = __builtin_amdgcn_readfirstlane((it * kThreads + tid) / kTilePix);
            if (j < K) en[it] = table[(((size_t)bt * D + d) * K + j) * kTilePix + (tid % kTilePix)];
        }
    };
    if (K > 0 && d_begin < d_end) prefetch(d_begin);

    // diagnostic instantiation only (STAMP, tools/stamp_sweep.py): cycles per loop segment, summed per wave
    unsigned long long tacc[8] = {0, 0, 0, 0, 0, 0, 0, 0}, tprev = 0;
#define MVS_STAMP(IDX)                                                           \
    if (STAMP) {                                                                 \
        __builtin_amdgcn_sched_barrier(0);                                       \
        const unsigned long long tn_ = __builtin_amdgcn_s_memtime();             \
        __builtin_amdgcn_sched_barrier(0);                                       \
        if ((IDX) >= 0) tacc[(IDX) < 0 ? 0 : (IDX)] += tn_ - tprev;              \
        tprev = tn_;                                                             \
    }
    for (int d = d_begin; d < d_end; ++d) {
        MVS_STAMP(-1)
        // ---- P1: footprint boxes (block-uniform scalars, computed once per tile by the coords kernel)
        int bx0[KK], bx1[KK], by0[KK], by1[KK], nc[KK], nr[KK];
        bool staged[KK], skip[KK];
        // make sure the prefetched values have landed before the DMA below is queued behind them
#pragma unroll
        for (int it = 0; it < ITER; ++it) asm volatile("" ::"v"(en[it].x), "v"(en[it].y));
#pragma unroll
        for (int j = 0; j < K; ++j) {
            const int4 b = bn[j];
            bx0[j] = __builtin_amdgcn_readfirstlane(b.x);
            bx1[j] = __builtin_amdgcn_readfirstlane(b.y);
            by0[j] = __builtin_amdgcn_readfirstlane(b.z);
            by1[j] = __builtin_amdgcn_readfirstlane(b.w);
            nc[j] = bx1[j] - bx0[j] + 1;
            nr[j] = by1[j] - by0[j] + 1;
            staged[j] = (bx1[j] >= bx0[j]) && (by1[j] >= by0[j]) && (nc[j] * nr[j] <= box_cap);
            skip[j] = (bx1[j] < bx0[j]) && (by1[j] == kBoxSkip);  // nothing of this neighbour is visible: w_j == 0
        }
        // the previous plane's tile reads (P5) must be over before the box storage is refilled
        if (d != d_begin) __syncthreads();
        MVS_STAMP(0)  // unpack boxes + wait for the prefetch + barrier (previous tile reads)
        // the box of the FIRST neighbour that is not skipped is in flight while the table is decoded
        bool early[KK];
        {
            bool taken = false;
#pragma unroll
            for (int j = 0; j < K; ++j) {
                early[j] = !taken && !skip[j];
                if (early[j]) {
                    taken = true;
                    if (staged[j]) load_box(j, bx0[j], by0[j], nc[j], nr[j]);
                }
            }
        }
        // ---- table entry (sample position) -> weights + tap offsets (float4 units, lane slot g not yet added)
#pragma unroll
        for (int it = 0; it < ITER; ++it) {
            const int j = __builtin_amdgcn_readfirstlane((it * kThreads + tid) / kTilePix);
            bool live = false;
#pragma unroll
            for (int jj = 0; jj < K; ++jj)
                if (jj == j && !skip[jj]) live = true;
            if (live) {
                const int p = tid % kTilePix;
                const SampleTaps tp = decode_sample(en[it].x, en[it].y, H, W);
                const int x0 = tp.x0, y0 = tp.y0;
                s_w[j][p] = tap_weights(tp);
                int lox = 0, hix = W - 1, loy = 0, hiy = H - 1, pitch = W;
#pragma unroll
                for (int jj = 0; jj < K; ++jj)
                    if (jj == j && staged[jj]) { lox = bx0[jj]; hix = bx1[jj]; loy = by0[jj]; hiy = by1[jj]; pitch = nc[jj]; }
                const int xa = clampi(x0, lox, hix) - lox, xb = clampi(x0 + 1, lox, hix) - lox;
                const int ya = (clampi(y0, loy, hiy) - loy) * pitch, yb = (clampi(y0 + 1, loy, hiy) - loy) * pitch;
                s_off[j][p] = make_int4((ya + xa) * 8, (ya + xb) * 8, (yb + xa) * 8, (yb + xb) * 8);
            }
        }

        MVS_STAMP(1)  // DMA issue of neighbour 0 + decode
        float S_[4][4], Q_[4][4];
#pragma unroll
        for (int s = 0; s < 4; ++s) {
            S_[s][0] = f[s].x; S_[s][1] = f[s].y; S_[s][2] = f[s].z; S_[s][3] = f[s].w;
            Q_[s][0] = f[s].x * f[s].x; Q_[s][1] = f[s].y * f[s].y; Q_[s][2] = f[s].z * f[s].z; Q_[s][3] = f[s].w * f[s].w;
        }
        bool tables_visible = false, prefetched = false;
#pragma unroll
        for (int j = 0; j < K; ++j) {
            if (skip[j]) continue;  // S and Q keep their values: the warped features are all zero
            if (!early[j] && staged[j]) load_box(j, bx0[j], by0[j], nc[j], nr[j]);
            // The LDS-DMA pieces of this wave count on vmcnt, and for a workgroup barrier hipcc only waits for
            // lgkmcnt: without the explicit wait a wave can pass the barrier while its own pieces are still in flight
            // and the other waves read stale box texels.  (Found when a variant let some waves skip the decode, whose
            // scratch reload had been supplying a vmcnt(0) by accident.)  s_waitcnt 0 = vmcnt, expcnt and lgkmcnt all
            // zero, so the decode's LDS table writes are covered by the same instruction.  Costs nothing measurable.
            if (staged[j]) __builtin_amdgcn_s_waitcnt(0);
            if (staged[j] || !tables_visible) __syncthreads();  // box (and, the first time, the tables) visible
            tables_visible = true;
            MVS_STAMP(2 + 2 * (j > 0 ? 1 : 0))  // (DMA issue of neighbour j>0) + wait for the box + barrier
            if (!prefetched && d + 1 < d_end) prefetch(d + 1);  // lands while the taps below are computed
            prefetched = true;
            // ---- P3: taps -> warped value -> running sums
            if (staged[j]) {
#pragma unroll
                for (int s = 0; s < 4; ++s) {
                    const int p = (wave * 4 + s) * 8 + ps;
                    const int4 o = s_off[j][p];
                    const float4 w = s_w[j][p];
                    const float4 t0 = s_box[o.x + g], t1 = s_box[o.y + g], t2 = s_box[o.z + g], t3 = s_box[o.w + g];
                    const float a0[4] = {t0.x, t0.y, t0.z, t0.w}, a1[4] = {t1.x, t1.y, t1.z, t1.w};
                    const float a2[4] = {t2.x, t2.y, t2.z, t2.w}, a3[4] = {t3.x, t3.y, t3.z, t3.w};
#pragma unroll
                    for (int i = 0; i < 4; ++i) {
                        float v = a0[i] * w.x;
                        v = fmaf(a1[i], w.y, v);
                        v = fmaf(a2[i], w.z, v);
                        v = fmaf(a3[i], w.w, v);
                        S_[s][i] = S_[s][i] + v;
                        Q_[s][i] = fmaf(v, v, Q_[s][i]);
                    }
                }
            } else {  // fallback: footprint too large (or empty): taps straight from the slab image
#pragma unroll
                for (int s = 0; s < 4; ++s) {
                    const int p = (wave * 4 + s) * 8 + ps;
                    const int4 o = s_off[j][p];
                    const float4 w = s_w[j][p];
                    const float4* b = nb_img[j] + g;
                    const float4 t0 = b[o.x], t1 = b[o.y], t2 = b[o.z], t3 = b[o.w];
                    const float a0[4] = {t0.x, t0.y, t0.z, t0.w}, a1[4] = {t1.x, t1.y, t1.z, t1.w};
                    const float a2[4] = {t2.x, t2.y, t2.z, t2.w}, a3[4] = {t3.x, t3.y, t3.z, t3.w};
#pragma unroll
                    for (int i = 0; i < 4; ++i) {
                        float v = a0[i] * w.x;
                        v = fmaf(a1[i], w.y, v);
                        v = fmaf(a2[i], w.z, v);
                        v = fmaf(a3[i], w.w, v);
                        S_[s][i] = S_[s][i] + v;
                        Q_[s][i] = fmaf(v, v, Q_[s][i]);
                    }
                }
            }
            if (staged[j]) __syncthreads();  // box fully read before it is overwritten (next neighbour / output tile)
            MVS_STAMP(3 + 2 * (j > 0 ? 1 : 0))  // taps + barrier
        }
        if (!prefetched && d + 1 < d_end) prefetch(d + 1);  // every neighbour was skipped
        // ---- P4: variance -> output tile [channel row 8*i+g][pixel]
        float* s_tile = reinterpret_cast<float*>(s_box);
        {
            float* t = s_tile + g * kTileStride + wave * 32 + ps;
#pragma unroll
            for (int s = 0; s < 4; ++s) {
#pragma unroll
                for (int i = 0; i < 4; ++i) {
                    const float m = S_[s][i] * rcp;
                    t[8 * i * kTileStride + s * 8] = fmaf(-m, m, Q_[s][i] * rcp);
                }
            }
        }
        // same-wave hand-over through LDS: the LDS queue of a wave is in order, only the compiler must not move
        // the reads above the writes
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
        __builtin_amdgcn_wave_barrier();
        __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
        MVS_STAMP(6)  // variance -> LDS tile
        // ---- P5: every wave stores its own 32 pixels: 8 channel rows per instruction, 16 bytes per lane
        {
            const float* t = s_tile + sh * kTileStride + st_p;
            int c0 = slab * kSlab + sh;
#pragma unroll
            for (int k = 0; k < 4; ++k) {
                const int c = c0 + 8 * k;
                const float4 v = *reinterpret_cast<const float4*>(t + 8 * k * kTileStride);
                if (c < C) {
                    OutT* dst = var + (((size_t)n * C + c) * D + d) * HW + st_off;
                    // written once, never re-read here: non-temporal keeps the stream from evicting the source slabs
                    if constexpr (sizeof(OutT) == 4) {
                        if (st_vec) {
                            typedef float v4f __attribute__((ext_vector_type(4)));
                            const v4f vv = {v.x, v.y, v.z, v.w};
                            if (NT) __builtin_nontemporal_store(vv, reinterpret_cast<v4f*>(dst));
                            else *reinterpret_cast<v4f*>(dst) = vv;
                        } else {
                            const float a[4] = {v.x, v.y, v.z, v.w};
#pragma unroll
                            for (int i = 0; i < 4; ++i)
                                if (i < st_n) dst[i] = a[i];
                        }
                    } else {
                        const __half h[4] = {__float2half_rn(v.x), __float2half_rn(v.y), __float2half_rn(v.z),
                                             __float2half_rn(v.w)};
                        if (st_vec) {  // 4 pixels x 2 B: the fp32 alignment condition also gives 8-byte alignment
                            typedef unsigned v2u __attribute__((ext_vector_type(2)));
                            const v2u vv = {(unsigned)__half_as_ushort(h[0]) | ((unsigned)__half_as_ushort(h[1]) << 16),
                                            (unsigned)__half_as_ushort(h[2]) | ((unsigned)__half_as_ushort(h[3]) << 16)};
                            if (NT) __builtin_nontemporal_store(vv, reinterpret_cast<v2u*>(dst));
                            else *reinterpret_cast<v2u*>(dst) = vv;
                        } else {
#pragma unroll
                            for (int i = 0; i < 4; ++i)
                                if (i < st_n) dst[i] = h[i];
                        }
                    }
                }
            }
        }
        MVS_STAMP(7)  // tile -> global stores
    }
    if (STAMP && stamps && lane == 0 && blockIdx.x < 65536) {
#pragma unroll
        for (int kk = 0; kk < 8; ++kk) stamps[((size_t)blockIdx.x * 4 + wave) * 8 + kk] = tacc[kk];
    }
#undef MVS_STAMP
}

}  // namespace mvsdet
