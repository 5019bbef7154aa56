// a3+a4 fused: plane-sweep variance, slab kernel (included by planesweep.hip).
//
//   block   = (reference view n, TWxTH pixel tile (128 pixels), 32-channel slab s, depth chunk)
//             logical id = (n*tiles + tile)*S + s, so consecutive blocks are the S slabs of one tile and the
//             round-robin block->XCD dispatch sends slab s of every tile to the same XCD when S == 8.
//   lanes   = (pixel slot ps = lane>>3, channel group g = lane&7); a wave-instruction covers 8 pixels x
//             128 B; each wave owns 32 pixels = 4 steps; the reference features of those pixels stay in
//             registers across the depth loop.
//   per depth plane
//     P1   one thread per (pixel, neighbour): sampling position -> {x0,y0}, weights in LDS; the valid tap
//          ranges are min/max-reduced with DPP + readlane into the footprint box of each neighbour
//     P1b  the same thread turns {x0,y0} into the four tap offsets inside the box (or inside the slab image
//          when the footprint does not fit and the neighbour falls back to global gathers)
//     per neighbour j
//       P2  every wave copies whole box rows (contiguous nc*128 B runs of the slab image) into LDS
//       P3  taps = 4 x ds_read_b128 per step; fma chain -> S, Q
//     P4   variance -> LDS tile [32 channels][128 pixels] (aliases the box storage)
//     P5   tile rows -> global, TW*4-byte contiguous runs, non-temporal, scalar row base + per-lane offset
//
// The kernel is VALU-issue bound, not bandwidth bound (profiles/r01_v2_*: 82 % VALU busy at 2*FETCH+WRITE =
// algorithmic bytes), so the code below is arranged to keep address arithmetic out of the inner loops:
// per-thread offsets are loop invariants, row bases are wave-uniform scalars, LDS addresses are one VGPR
// plus immediates.
//
// Arithmetic (device rounding, oracle mode 1): warped = fma chain over the 4 taps; S = f + w1 + ..;
// Q = fma(w,w,Q); var = fma(-m, m, Q*r) with m = S*r, r = 1/(K+1).
#pragma once
#include "common.h"
#include "pack.h"

namespace mvsdet {

constexpr int kTilePix = 128;       // pixels per tile
constexpr int kBoxCap = 224;        // texels (128 B each) of the LDS footprint box: 28 KiB
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

template <int K, int TW, bool NT, bool DMA>
__global__ __launch_bounds__(kThreads, 4) void plane_sweep_variance_kernel(
    const float* __restrict__ packed, const int64_t* __restrict__ nbr, const float* __restrict__ proj,
    const float* __restrict__ depth, float* __restrict__ var, int N, int C, int S, int D, int H, int W, int tiles_x,
    int tiles, int d_per_block) {
    constexpr int KK = K > 0 ? K : 1;
    constexpr int TH = kTilePix / TW;
    static_assert(kBoxCap * 8 >= 32 * kTileStride / 4, "output tile must fit in the box storage");
    __shared__ float4 s_box[kBoxCap * 8];       // footprint box of one neighbour; later the output tile
    __shared__ int2 s_xy[KK][kTilePix];         // tap origin per (neighbour, pixel)
    __shared__ int4 s_off[KK][kTilePix];        // float4 index of the 4 taps (inside s_box or the slab image)
    __shared__ float4 s_w[KK][kTilePix];        // tap weights
    __shared__ int s_bounds[2][KK][4];          // xlo, xhi, ylo, yhi of the valid taps; double-buffered by plane

    const int HW = H * W;
    const int id = blockIdx.x;
    const int slab = id % S;
    const int t_ = id / S;
    const int tile = t_ % tiles, n = t_ / tiles;
    const int tx0 = (tile % tiles_x) * TW, ty0 = (tile / tiles_x) * TH;
    const int d_begin = blockIdx.y * d_per_block;
    const int d_end = min(D, d_begin + d_per_block);

    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);  // wave-uniform by construction: keep it scalar
    const int g = lane & 7, ps = lane >> 3;
    const size_t slab_stride = (size_t)HW * kSlab;       // floats per (view, slab) image
    const float* ref_img = packed + ((size_t)n * S + slab) * slab_stride;
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
    // ... and the two output pixels it stores in P5 (tile pixel index lane and lane + 64)
    int st_off[2];
    bool st_ok[2];
#pragma unroll
    for (int h = 0; h < 2; ++h) {
        const int pp = lane + 64 * h;
        const int x = tx0 + (pp % TW), y = ty0 + (pp / TW);
        st_ok[h] = (x < W) && (y < H);
        st_off[h] = y * W + x;
    }
    if (tid < 2 * KK * 4) {  // both bounds buffers start empty
        const int e = tid & 3;
        (&s_bounds[0][0][0])[tid] = (e == 0 || e == 2) ? INT32_MAX : INT32_MIN;
    }
    __syncthreads();

    for (int d = d_begin; d < d_end; ++d) {
        const int cur = (d - d_begin) & 1;
        // ---- P1: sampling positions + footprint box
        if (K > 0) {
            const float dval = depth[(size_t)n * D + d];
#pragma unroll
            for (int it = 0; it < (K * kTilePix + kThreads - 1) / kThreads; ++it) {
                // 128 pixels = 2 whole waves per neighbour: j is wave-uniform
                const int j = __builtin_amdgcn_readfirstlane((it * kThreads + tid) / kTilePix);
                if (j < K) {
                    const int p = tid % kTilePix;
                    const int x = tx0 + (p % TW), y = ty0 + (p / TW);
                    TapXY t;
                    if (x < W && y < H) {
                        t = compute_taps_xy(proj + ((size_t)n * K + j) * 16, (float)x, (float)y, dval, H, W);
                    } else {  // pixel outside the image: no taps, no footprint
                        t.x0 = t.y0 = 0;
                        t.w = make_float4(0.f, 0.f, 0.f, 0.f);
                        t.xlo = t.ylo = INT32_MAX;
                        t.xhi = t.yhi = INT32_MIN;
                    }
                    s_xy[j][p] = make_int2(t.x0, t.y0);
                    s_w[j][p] = t.w;
                    const int xlo = wave_reduce<true>(t.xlo), xhi = wave_reduce<false>(t.xhi);
                    const int ylo = wave_reduce<true>(t.ylo), yhi = wave_reduce<false>(t.yhi);
                    if (lane == 0) {
                        atomicMin(&s_bounds[cur][j][0], xlo);
                        atomicMax(&s_bounds[cur][j][1], xhi);
                        atomicMin(&s_bounds[cur][j][2], ylo);
                        atomicMax(&s_bounds[cur][j][3], yhi);
                    }
                }
            }
            if (tid < KK * 4) {  // reset the other buffer for the next plane (its readers are behind a barrier)
                const int e = tid & 3;
                (&s_bounds[cur ^ 1][0][0])[tid] = (e == 0 || e == 2) ? INT32_MAX : INT32_MIN;
            }
        }
        __syncthreads();

        // footprint boxes (block-uniform scalars)
        int bx0[KK], bx1[KK], by0[KK], by1[KK], nc[KK], nr[KK];
        bool staged[KK];
#pragma unroll
        for (int j = 0; j < K; ++j) {
            bx0[j] = __builtin_amdgcn_readfirstlane(s_bounds[cur][j][0]);
            bx1[j] = __builtin_amdgcn_readfirstlane(s_bounds[cur][j][1]);
            by0[j] = __builtin_amdgcn_readfirstlane(s_bounds[cur][j][2]);
            by1[j] = __builtin_amdgcn_readfirstlane(s_bounds[cur][j][3]);
            nc[j] = bx1[j] - bx0[j] + 1;
            nr[j] = by1[j] - by0[j] + 1;
            staged[j] = (bx1[j] >= bx0[j]) && (by1[j] >= by0[j]) && (nc[j] * nr[j] <= kBoxCap);
        }
        // ---- P1b: tap offsets (float4 units, lane slot g not yet added) inside the box / the slab image
        if (K > 0) {
#pragma unroll
            for (int it = 0; it < (K * kTilePix + kThreads - 1) / kThreads; ++it) {
                const int j = __builtin_amdgcn_readfirstlane((it * kThreads + tid) / kTilePix);
                if (j < K) {
                    const int p = tid % kTilePix;
                    const int2 xy = s_xy[j][p];
                    int lox = 0, hix = W - 1, loy = 0, hiy = H - 1, pitch = W;
#pragma unroll
                    for (int jj = 0; jj < K; ++jj)
                        if (jj == j && staged[jj]) { lox = bx0[jj]; hix = bx1[jj]; loy = by0[jj]; hiy = by1[jj]; pitch = nc[jj]; }
                    const int xa = clampi(xy.x, lox, hix) - lox, xb = clampi(xy.x + 1, lox, hix) - lox;
                    const int ya = (clampi(xy.y, loy, hiy) - loy) * pitch, yb = (clampi(xy.y + 1, loy, hiy) - loy) * pitch;
                    s_off[j][p] = make_int4((ya + xa) * 8, (ya + xb) * 8, (yb + xa) * 8, (yb + xb) * 8);
                }
            }
        }

        float S_[4][4], Q_[4][4];
#pragma unroll
        for (int s = 0; s < 4; ++s) {
            S_[s][0] = f[s].x; S_[s][1] = f[s].y; S_[s][2] = f[s].z; S_[s][3] = f[s].w;
            Q_[s][0] = f[s].x * f[s].x; Q_[s][1] = f[s].y * f[s].y; Q_[s][2] = f[s].z * f[s].z; Q_[s][3] = f[s].w * f[s].w;
        }
#pragma unroll
        for (int j = 0; j < K; ++j) {
            if (staged[j]) {
                // ---- P2: every wave copies whole rows of the box: one contiguous nc*128-byte run each
                const int row_f4 = nc[j] * 8;
                for (int row = wave; row < nr[j]; row += 4) {
                    const float4* src = nb_img[j] + ((size_t)(by0[j] + row) * W + bx0[j]) * 8;  // wave-uniform
                    float4* dst = s_box + row * row_f4;
                    if (DMA) {
                        // LDS-DMA: each wave-instruction moves 1 KiB global -> LDS without touching VGPRs; the LDS
                        // address is the wave-uniform base + 16*lane, the global address is per lane.  The DMA counts
                        // on vmcnt, which the __syncthreads() below drains.
                        for (int q0 = 0; q0 < row_f4; q0 += 64)
                            if (q0 + lane < row_f4)
                                __builtin_amdgcn_global_load_lds(
                                    (const __attribute__((address_space(1))) void*)(src + q0 + lane),
                                    (__attribute__((address_space(3))) void*)(dst + q0), 16, 0, 0);
                    } else {
                        for (int q = lane; q < row_f4; q += 64) dst[q] = src[q];
                    }
                }
            }
            __syncthreads();  // box (and, for the first neighbour, the tap offsets) visible
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
            __syncthreads();  // box fully read before it is overwritten (next neighbour / output tile)
        }
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
        __syncthreads();
        // ---- P5: wave w stores channel rows w, w+4, ..: scalar row base, per-lane pixel offset
        {
            const float* t = s_tile + wave * kTileStride + lane;
#pragma unroll
            for (int k = 0; k < 8; ++k) {
                const int c = slab * kSlab + wave + 4 * k;  // wave-uniform
                if (c < C) {
                    float* row = var + (((size_t)n * C + c) * D + d) * HW;
#pragma unroll
                    for (int h = 0; h < 2; ++h) {
                        const float v = t[4 * k * kTileStride + 64 * h];
                        if (st_ok[h]) {
                            // written once, never re-read here: keep the stream from evicting the source slabs
                            if (NT) __builtin_nontemporal_store(v, row + st_off[h]);
                            else row[st_off[h]] = v;
                        }
                    }
                }
            }
        }
        // The next plane writes s_box again only after the barriers that follow its P1, which every thread
        // reaches after its tile reads above -- unless K == 0, where P4 follows directly.
        if (K == 0) __syncthreads();
    }
}

}  // namespace mvsdet
