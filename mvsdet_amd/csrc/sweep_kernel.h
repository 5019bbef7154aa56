// a3+a4 fused: plane-sweep variance (included by planesweep.hip).  Two kernels:
//
// (1) plane_sweep_coords_kernel -- everything that depends on (view, neighbour, plane, pixel) but NOT on the
//     channel: the sampling position of mvs_models/module.py:116-143, reduced to a 16-byte table entry
//     {tap origin x0,y0; fractional weights wx,wy; validity bits} plus, per (tile, neighbour, plane), the
//     bounding box of the valid taps.  The reference builds its sampling grid once per plane too.
//     Cost: N*K*D*H*W entries (1.6 GB at the 64-plane shape, 3 % of the cost volume), written once.
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
//         P1  scalar-load the footprint boxes; start the LDS-DMA of neighbour 0's box; one thread per (pixel,
//             neighbour) decodes its table entry into 4 weights + 4 tap offsets inside the box (or inside the
//             slab image when the footprint does not fit in LDS and that neighbour gathers from global memory)
//         per neighbour j
//           P2  LDS-DMA of the box rows (contiguous nc*128-byte runs of the slab image), 1 KiB per wave-instruction
//           P3  taps = 4 x ds_read_b128 per step; fma chain -> S, Q
//         P4  variance -> LDS tile [32 channels][128 pixels] (aliases the box storage)
//         P5  tile rows -> global as 16-byte non-temporal stores, 128 B contiguous per (channel, tile row)
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
constexpr int kBoxPool = 352;       // texels (128 B each) of the LDS pool that holds the footprint boxes of one plane: 44 KiB

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

// Table entry: x = (x0+1) | (y0+1) << 16 with the clamped tap origin of compute_taps_xy(); y, z = bit patterns
// of wx, wy; w = validity bits (1: column x0 inside, 2: column x0+1 inside, 4: row y0 inside, 8: row y0+1 inside).
__device__ __forceinline__ uint4 encode_taps(const float* __restrict__ P, float x, float y, float d, int H, int W,
                                             int& xlo, int& xhi, int& ylo, int& yhi) {
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
    const bool x0in = (x0 >= 0.0f) && (x0 <= (float)(W - 1));
    const bool x1in = (x0 >= -1.0f) && (x0 <= (float)(W - 2));
    const bool y0in = (y0 >= 0.0f) && (y0 <= (float)(H - 1));
    const bool y1in = (y0 >= -1.0f) && (y0 <= (float)(H - 2));
    const int xi = (int)fminf(fmaxf(x0, -1.0f), (float)(W - 1));  // NaN / Inf positions become finite indices
    const int yi = (int)fminf(fmaxf(y0, -1.0f), (float)(H - 1));
    const bool any = (x0in || x1in) && (y0in || y1in);
    xlo = any ? (x0in ? xi : xi + 1) : INT32_MAX;
    xhi = any ? (x1in ? xi + 1 : xi) : INT32_MIN;
    ylo = any ? (y0in ? yi : yi + 1) : INT32_MAX;
    yhi = any ? (y1in ? yi + 1 : yi) : INT32_MIN;
    uint4 e;
    e.x = (unsigned)(xi + 1) | ((unsigned)(yi + 1) << 16);
    e.y = __float_as_uint(wx);
    e.z = __float_as_uint(wy);
    e.w = (x0in ? 1u : 0u) | (x1in ? 2u : 0u) | (y0in ? 4u : 0u) | (y1in ? 8u : 0u);
    return e;
}

// ---------------------------------------------------------------------------------------------
// (1) sampling table: block = (view, tile); thread = (neighbour, pixel); loops over the planes of its chunk.
//     table [((n*tiles + tile)*D + d)*K + j][128] uint4;  boxes [((n*tiles + tile)*D + d)*K + j] int4.
// ---------------------------------------------------------------------------------------------
template <int K, int TW>
__global__ __launch_bounds__(kThreads) void plane_sweep_coords_kernel(const float* __restrict__ proj,
                                                                       const float* __restrict__ depth,
                                                                       uint4* __restrict__ table, int4* __restrict__ boxes,
                                                                       int D, int H, int W, int tiles_x, int tiles,
                                                                       int d_per_block) {
    constexpr int TH = kTilePix / TW;
    constexpr int ITER = (K * kTilePix + kThreads - 1) / kThreads;
    __shared__ int s_red[2][K][2][4];  // [plane parity][neighbour][wave of the neighbour][xlo,xhi,ylo,yhi]
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
                uint4 e = make_uint4(0x00010001u, 0u, 0u, 0u);  // pixel outside the image: no taps, no footprint
                if (inside) e = encode_taps(proj + ((size_t)n * K + j) * 16, (float)x, (float)y, dval, H, W, xlo, xhi, ylo, yhi);
                table[(base + j) * kTilePix + p] = e;
                xlo = wave_reduce<true>(xlo);
                xhi = wave_reduce<false>(xhi);
                ylo = wave_reduce<true>(ylo);
                yhi = wave_reduce<false>(yhi);
                if (lane == 0) {
                    int* r = s_red[par][j][(tid >> 6) & 1];
                    r[0] = xlo; r[1] = xhi; r[2] = ylo; r[3] = yhi;
                }
            }
        }
        __syncthreads();  // one barrier per plane: s_red is double-buffered by plane parity
        if (tid < K) {
            const int* a = s_red[par][tid][0];
            const int* b = s_red[par][tid][1];
            boxes[base + tid] = make_int4(min(a[0], b[0]), max(a[1], b[1]), min(a[2], b[2]), max(a[3], b[3]));
        }
    }
}

// ---------------------------------------------------------------------------------------------
// (2) the slab kernel
// ---------------------------------------------------------------------------------------------
template <int K, int TW, bool NT, bool STAMP>
__global__ __launch_bounds__(kThreads, 3) void plane_sweep_variance_kernel(
    const float* __restrict__ packed, const int64_t* __restrict__ nbr, const uint4* __restrict__ table,
    const int4* __restrict__ boxes, float* __restrict__ var, int N, int C, int S, int D, int H, int W, int tiles_x,
    int tiles, int d_per_block, int box_cap, unsigned long long* __restrict__ stamps) {
    constexpr int KK = K > 0 ? K : 1;
    constexpr int TH = kTilePix / TW;
    constexpr int ITER = (KK * kTilePix + kThreads - 1) / kThreads;
    __shared__ float4 s_box[kBoxPool * 8];      // footprint boxes of all neighbours of the current plane, back to back
    __shared__ int4 s_off[KK][kTilePix];        // float4 index of the 4 taps (inside s_box or the slab image)
    __shared__ float4 s_w[KK][kTilePix];        // tap weights

    const int HW = H * W;
    const int id = blockIdx.x;
    const int slab = id % S;
    const int bt = id / S;  // n*tiles + tile
    const int tile = bt % tiles, n = bt / tiles;
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

    // This lane owns 4 CONSECUTIVE pixels of one tile row (float4 slot q of the 128-pixel tile, TW/4 slots per
    // row) for the 4 channels 8*i + g of the slab: its results go to global memory straight from registers as
    // 16-byte stores, and a wave-instruction writes 8 channel rows x 128 contiguous bytes.
    const int q = wave * 8 + ps;
    const int st_x = tx0 + (q % (TW / 4)) * 4, st_y = ty0 + q / (TW / 4);
    const int st_off = st_y * W + st_x;
    const int st_n = (st_y < H) ? max(0, min(4, W - st_x)) : 0;            // how many of the 4 pixels are inside the image
    const bool st_vec = (st_n == 4) && ((W & 3) == 0) && ((HW & 3) == 0);  // 16-byte aligned in every channel row
    // Number of vector-store instructions this wave issues per plane (wave-uniform), or -1 when some lane takes the
    // scalar edge path: lets the loop wait for the LDS-DMA with a COUNTED vmcnt that leaves the stores in flight.
    int nst = 0;
#pragma unroll
    for (int i = 0; i < 4; ++i) nst += (slab * kSlab + 8 * i < C) ? 1 : 0;
    if (!__all(st_vec)) nst = -1;
    nst = __builtin_amdgcn_readfirstlane(nst);
    // element offset of this lane's 4 pixels in channel row 8*i + g at plane d_begin; advanced by HW per plane
    size_t st_idx[4];
#pragma unroll
    for (int i = 0; i < 4; ++i)
        st_idx[i] = (((size_t)n * C + min(slab * kSlab + 8 * i + g, C - 1)) * D + d_begin) * HW + st_off;
    float4 f[4];  // reference features of the 4 pixels, kept across the depth loop
#pragma unroll
    for (int s = 0; s < 4; ++s) {
        const int pix = min(st_y, H - 1) * W + min(st_x + s, W - 1);
        f[s] = *reinterpret_cast<const float4*>(ref_img + (size_t)pix * kSlab + 4 * g);
    }

    // Footprint boxes of one plane: block-uniform scalars, computed once per tile by the coords kernel.  Plain
    // arrays + macros (no structs / lambdas): everything below must stay in SGPRs / VGPRs after unrolling.
#define MVS_LOAD_BOXES(DD, BV)                                                                                  \
    _Pragma("unroll") for (int j = 0; j < K; ++j) BV[j] = boxes[((size_t)bt * D + (DD)) * K + j];
#define MVS_UNPACK_BOXES(BV, BX0, BY0, BX1, BY1, NC, NR, BASE, STG)                                            \
    {                                                                                                          \
        int used_ = 0;                                                                                         \
        _Pragma("unroll") for (int j = 0; j < K; ++j) {                                                        \
            BX0[j] = __builtin_amdgcn_readfirstlane(BV[j].x);                                                  \
            BX1[j] = __builtin_amdgcn_readfirstlane(BV[j].y);                                                  \
            BY0[j] = __builtin_amdgcn_readfirstlane(BV[j].z);                                                  \
            BY1[j] = __builtin_amdgcn_readfirstlane(BV[j].w);                                                  \
            NC[j] = BX1[j] - BX0[j] + 1;                                                                       \
            NR[j] = BY1[j] - BY0[j] + 1;                                                                       \
            const int need_ = NC[j] * NR[j];                                                                   \
            STG[j] = (BX1[j] >= BX0[j]) && (BY1[j] >= BY0[j]) && (used_ + need_ <= box_cap);                   \
            BASE[j] = used_ * 8; /* float4 index of this neighbour's box inside the pool */                    \
            if (STG[j]) used_ += need_;                                                                        \
        }                                                                                                      \
    }
#define MVS_READ_BOXES(DD, BX0, BY0, BX1, BY1, NC, NR, BASE, STG)                                              \
    {                                                                                                          \
        int4 bv_[KK];                                                                                          \
        MVS_LOAD_BOXES(DD, bv_)                                                                                \
        MVS_UNPACK_BOXES(bv_, BX0, BY0, BX1, BY1, NC, NR, BASE, STG)                                           \
    }
    // LDS-DMA: each wave-instruction moves 1 KiB global -> LDS without touching VGPRs; the LDS address is the
    // wave-uniform base + 16*lane, the global address is per lane.  The DMA counts on vmcnt, which the next
    // __syncthreads() drains.  Every wave copies whole rows: one contiguous nc*128-byte run each.
#define MVS_ISSUE_DMA(BX0, BY0, NC, NR, BASE, STG)                                                             \
    _Pragma("unroll") for (int j = 0; j < K; ++j) {                                                            \
        if (STG[j]) {                                                                                          \
            const int row_f4_ = NC[j] * 8;                                                                     \
            for (int row = wave; row < NR[j]; row += 4) {                                                      \
                const float4* src_ = nb_img[j] + ((size_t)(BY0[j] + row) * W + BX0[j]) * 8; /* wave-uniform */ \
                float4* dst_ = s_box + BASE[j] + row * row_f4_;                                                \
                for (int q0 = 0; q0 < row_f4_; q0 += 64)                                                       \
                    if (q0 + lane < row_f4_)                                                                   \
                        __builtin_amdgcn_global_load_lds(                                                      \
                            (const __attribute__((address_space(1))) void*)(src_ + q0 + lane),                 \
                            (__attribute__((address_space(3))) void*)(dst_ + q0), 16, 0, 0);                   \
            }                                                                                                  \
        }                                                                                                      \
    }
    // table entry of (neighbour, pixel) = thread -> registers
#define MVS_LOAD_ENTRIES(DD, E)                                                                                \
    _Pragma("unroll") for (int it = 0; it < ITER; ++it) {                                                      \
        const int j_ = __builtin_amdgcn_readfirstlane((it * kThreads + tid) / kTilePix);                       \
        E[it] = make_uint4(0x00010001u, 0u, 0u, 0u);                                                           \
        if (j_ < K) E[it] = table[(((size_t)bt * D + (DD)) * K + j_) * kTilePix + (tid % kTilePix)];           \
    }
    // entry -> weights + tap offsets (float4 units, lane slot g not yet added) in LDS
#define MVS_DECODE(E, BX0, BY0, BX1, BY1, NC, BASE, STG)                                                       \
    _Pragma("unroll") for (int it = 0; it < ITER; ++it) {                                                      \
        const int j_ = __builtin_amdgcn_readfirstlane((it * kThreads + tid) / kTilePix);                       \
        if (j_ < K) {                                                                                          \
            const int p_ = tid % kTilePix;                                                                     \
            const uint4 e_ = E[it];                                                                            \
            const int x0_ = (int)(e_.x & 0xffffu) - 1, y0_ = (int)(e_.x >> 16) - 1;                            \
            const float wx_ = __uint_as_float(e_.y), wy_ = __uint_as_float(e_.z);                             \
            const float ex_ = 1.0f - wx_, sy_ = 1.0f - wy_;                                                    \
            const float wnw_ = sy_ * ex_, wne_ = sy_ * wx_, wsw_ = wy_ * ex_, wse_ = wy_ * wx_;                \
            float4 w_; /* an outside tap carries weight*0, so Inf/NaN positions give NaN as ATen-CPU does */   \
            w_.x = ((e_.w & 5u) == 5u) ? wnw_ : wnw_ * 0.0f;                                                   \
            w_.y = ((e_.w & 6u) == 6u) ? wne_ : wne_ * 0.0f;                                                   \
            w_.z = ((e_.w & 9u) == 9u) ? wsw_ : wsw_ * 0.0f;                                                   \
            w_.w = ((e_.w & 10u) == 10u) ? wse_ : wse_ * 0.0f;                                                 \
            s_w[j_][p_] = w_;                                                                                  \
            int lox_ = 0, hix_ = W - 1, loy_ = 0, hiy_ = H - 1, pitch_ = W, base_ = 0;                         \
            _Pragma("unroll") for (int jj = 0; jj < K; ++jj) if (jj == j_ && STG[jj]) {                        \
                lox_ = BX0[jj]; hix_ = BX1[jj]; loy_ = BY0[jj]; hiy_ = BY1[jj]; pitch_ = NC[jj]; base_ = BASE[jj]; \
            }                                                                                                  \
            const int xa_ = clampi(x0_, lox_, hix_) - lox_, xb_ = clampi(x0_ + 1, lox_, hix_) - lox_;          \
            const int ya_ = (clampi(y0_, loy_, hiy_) - loy_) * pitch_;                                         \
            const int yb_ = (clampi(y0_ + 1, loy_, hiy_) - loy_) * pitch_;                                     \
            s_off[j_][p_] = make_int4(base_ + (ya_ + xa_) * 8, base_ + (ya_ + xb_) * 8, base_ + (yb_ + xa_) * 8, \
                                      base_ + (yb_ + xb_) * 8);                                                \
        }                                                                                                      \
    }

    // c_*: boxes of the plane being computed; n_*: boxes of the next plane (its DMA is issued one plane ahead);
    // en: table entries of the next plane, loaded two vm-operations ahead of the result stores so that no wait
    // for them ever has to wait for a store acknowledgement (vmcnt retires in order).
    int c_bx0[KK], c_by0[KK], c_bx1[KK], c_by1[KK], c_nc[KK], c_nr[KK], c_base[KK];
    bool c_stg[KK];
    int n_bx0[KK], n_by0[KK], n_bx1[KK], n_by1[KK], n_nc[KK], n_nr[KK], n_base[KK];
    bool n_stg[KK];
    uint4 en[ITER];
#pragma unroll
    for (int j = 0; j < KK; ++j) {
        c_bx0[j] = c_by0[j] = n_bx0[j] = n_by0[j] = 0;
        c_bx1[j] = c_by1[j] = n_bx1[j] = n_by1[j] = -1;
        c_nc[j] = c_nr[j] = c_base[j] = n_nc[j] = n_nr[j] = n_base[j] = 0;
        c_stg[j] = n_stg[j] = false;
    }
#pragma unroll
    for (int it = 0; it < ITER; ++it) en[it] = make_uint4(0x00010001u, 0u, 0u, 0u);
    // ---- prologue: boxes, DMA and tables of the first plane; boxes and entries of the second
    if (K > 0 && d_begin < d_end) {
        uint4 e0[ITER];
        MVS_READ_BOXES(d_begin, c_bx0, c_by0, c_bx1, c_by1, c_nc, c_nr, c_base, c_stg)
        MVS_LOAD_ENTRIES(d_begin, e0)
        MVS_ISSUE_DMA(c_bx0, c_by0, c_nc, c_nr, c_base, c_stg)
        MVS_DECODE(e0, c_bx0, c_by0, c_bx1, c_by1, c_nc, c_base, c_stg)
        if (d_begin + 1 < d_end) {
            MVS_READ_BOXES(d_begin + 1, n_bx0, n_by0, n_bx1, n_by1, n_nc, n_nr, n_base, n_stg)
            MVS_LOAD_ENTRIES(d_begin + 1, en)
        }
    }
    // diagnostic build only (STAMP): cycles spent per loop segment, summed per wave
    unsigned long long tacc[6] = {0, 0, 0, 0, 0, 0}, tprev = 0;
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
        // ---- barrier 1 of 2: the boxes (DMA drained) and the tables of plane d are visible.
        // Memory operations of this wave still in flight here, oldest first: the DMA of plane d, then the nst
        // result stores of plane d-1.  vmcnt(nst) retires the DMA and leaves the
        // stores alone (waiting for their acknowledgements every plane was the longest stall of the loop).  hipcc
        // does not carry a pending LDS-DMA across the loop back-edge, and with an inline wait in front of the
        // barrier it also left out the lgkmcnt(0) for the table writes (rare stale reads): both are written out.
        if (d == d_begin || nst < 0) asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");
        else if (nst == 4) asm volatile("s_waitcnt vmcnt(4) lgkmcnt(0)" ::: "memory");
        else if (nst == 3) asm volatile("s_waitcnt vmcnt(3) lgkmcnt(0)" ::: "memory");
        else if (nst == 2) asm volatile("s_waitcnt vmcnt(2) lgkmcnt(0)" ::: "memory");
        else if (nst == 1) asm volatile("s_waitcnt vmcnt(1) lgkmcnt(0)" ::: "memory");
        else asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");
        __builtin_amdgcn_s_barrier();
        asm volatile("" ::: "memory");
        MVS_STAMP(0)
        const bool more = (d + 1 < d_end);
        // boxes / table entries of plane d+2: requested here, so their latency hides behind the taps of plane d and
        // the only older memory operations they can be queued behind are the (long finished) stores of plane d-1
        int4 bm[KK];
        uint4 em[ITER];
        const bool more2 = (K > 0) && (d + 2 < d_end);
        if (more2) {
            MVS_LOAD_BOXES(d + 2, bm)
            MVS_LOAD_ENTRIES(d + 2, em)
        }
        float S_[4][4], Q_[4][4];
#pragma unroll
        for (int s = 0; s < 4; ++s) {
            S_[s][0] = f[s].x; S_[s][1] = f[s].y; S_[s][2] = f[s].z; S_[s][3] = f[s].w;
            Q_[s][0] = f[s].x * f[s].x; Q_[s][1] = f[s].y * f[s].y; Q_[s][2] = f[s].z * f[s].z; Q_[s][3] = f[s].w * f[s].w;
        }
        // ---- taps -> warped value -> running sums, all neighbours back to back.  The LDS and the global variant
        // are separate loops on purpose: a per-tap select between them makes the compiler emit FLAT loads.
#define MVS_ACCUMULATE(T0, T1, T2, T3)                                                                  \
    {                                                                                                   \
        const float a0[4] = {T0.x, T0.y, T0.z, T0.w}, a1[4] = {T1.x, T1.y, T1.z, T1.w};                 \
        const float a2[4] = {T2.x, T2.y, T2.z, T2.w}, a3[4] = {T3.x, T3.y, T3.z, T3.w};                 \
        _Pragma("unroll") for (int i = 0; i < 4; ++i) {                                                 \
            float v = a0[i] * w.x;                                                                      \
            v = fmaf(a1[i], w.y, v);                                                                    \
            v = fmaf(a2[i], w.z, v);                                                                    \
            v = fmaf(a3[i], w.w, v);                                                                    \
            S_[s][i] = S_[s][i] + v;                                                                    \
            Q_[s][i] = fmaf(v, v, Q_[s][i]);                                                            \
        }                                                                                               \
    }
#pragma unroll
        for (int j = 0; j < K; ++j) {
            if (c_stg[j]) {  // block-uniform
#pragma unroll
                for (int s = 0; s < 4; ++s) {
                    const int p = 4 * q + s;
                    const int4 o = s_off[j][p];
                    const float4 w = s_w[j][p];
                    const float4 t0 = s_box[o.x + g], t1 = s_box[o.y + g], t2 = s_box[o.z + g], t3 = s_box[o.w + g];
                    MVS_ACCUMULATE(t0, t1, t2, t3)
                }
            } else {  // footprint too large for the pool (or empty): taps straight from the slab image (L2)
#pragma unroll
                for (int s = 0; s < 4; ++s) {
                    const int p = 4 * q + s;
                    const int4 o = s_off[j][p];
                    const float4 w = s_w[j][p];
                    const float4* b = nb_img[j] + g;
                    const float4 t0 = b[o.x], t1 = b[o.y], t2 = b[o.z], t3 = b[o.w];
                    MVS_ACCUMULATE(t0, t1, t2, t3)
                }
            }
        }
#undef MVS_ACCUMULATE
        MVS_STAMP(1)
        // ---- barrier 2 of 2: every wave is done with the boxes and tables of plane d
        int m_bx0[KK], m_by0[KK], m_bx1[KK], m_by1[KK], m_nc[KK], m_nr[KK], m_base[KK];
        bool m_stg[KK];
        if (more2) MVS_UNPACK_BOXES(bm, m_bx0, m_by0, m_bx1, m_by1, m_nc, m_nr, m_base, m_stg)
        if (K > 0 && more) {
            __syncthreads();
            MVS_STAMP(2)
            // next plane's boxes fly while this plane's results are stored and its tables decoded; nothing else is
            // requested from memory between here and the loop top (the counted vmcnt there relies on it)
            MVS_ISSUE_DMA(n_bx0, n_by0, n_nc, n_nr, n_base, n_stg)
        }
        MVS_STAMP(3)
        // ---- variance, stored straight from registers: channel 8*i + g, 4 consecutive pixels
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            const int c = slab * kSlab + 8 * i + g;
            float r_[4];
#pragma unroll
            for (int s = 0; s < 4; ++s) {
                const float m = S_[s][i] * rcp;
                r_[s] = fmaf(-m, m, Q_[s][i] * rcp);
            }
            if (c < C) {
                float* dst = var + st_idx[i];
                // written once, never re-read here: non-temporal keeps the stream from evicting the source slabs
                if (st_vec) {
                    typedef float v4f __attribute__((ext_vector_type(4)));
                    const v4f vv = {r_[0], r_[1], r_[2], r_[3]};
                    if (NT) __builtin_nontemporal_store(vv, reinterpret_cast<v4f*>(dst));
                    else *reinterpret_cast<v4f*>(dst) = vv;
                } else {
#pragma unroll
                    for (int s = 0; s < 4; ++s)
                        if (s < st_n) dst[s] = r_[s];
                }
            }
        }
#pragma unroll
        for (int i = 0; i < 4; ++i) st_idx[i] += HW;
        MVS_STAMP(4)
        if (K > 0 && more) {
            MVS_DECODE(en, n_bx0, n_by0, n_bx1, n_by1, n_nc, n_base, n_stg)
#pragma unroll
            for (int j = 0; j < K; ++j) {
                c_bx0[j] = n_bx0[j]; c_by0[j] = n_by0[j]; c_bx1[j] = n_bx1[j]; c_by1[j] = n_by1[j];
                c_nc[j] = n_nc[j]; c_nr[j] = n_nr[j]; c_base[j] = n_base[j]; c_stg[j] = n_stg[j];
            }
            if (more2) {
#pragma unroll
                for (int j = 0; j < K; ++j) {
                    n_bx0[j] = m_bx0[j]; n_by0[j] = m_by0[j]; n_bx1[j] = m_bx1[j]; n_by1[j] = m_by1[j];
                    n_nc[j] = m_nc[j]; n_nr[j] = m_nr[j]; n_base[j] = m_base[j]; n_stg[j] = m_stg[j];
                }
#pragma unroll
                for (int it = 0; it < ITER; ++it) en[it] = em[it];
            }
        }
        MVS_STAMP(5)
    }
    if (STAMP && stamps && lane == 0 && blockIdx.x < 65536) {
#pragma unroll
        for (int k = 0; k < 6; ++k) stamps[((size_t)blockIdx.x * 4 + wave) * 8 + k] = tacc[k];
    }
#undef MVS_STAMP
#undef MVS_READ_BOXES
#undef MVS_LOAD_BOXES
#undef MVS_UNPACK_BOXES
#undef MVS_ISSUE_DMA
#undef MVS_LOAD_ENTRIES
#undef MVS_DECODE
}

}  // namespace mvsdet
