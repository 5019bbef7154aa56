// Weight gradient of Conv3d(kernel 3, stride 1, padding 1) of the cost regularisation network (mvs_models/mvsnet.py:76-82)
// on the bf16 matrix cores with three-term split operands (see costreg_bf16.hip for the arithmetic):
//
//     dW[o][c][kd,kh,kw] = sum over views and voxels (d,h,w) of  dY[n,o,d,h,w] * X[n,c,d+kd-1,h+kh-1,w+kw-1]
//
// A GEMM per tap with the VOXELS as the reduction: D[o][c] += A[o][k] * B[k][c], the 16 k of v_mfma_f32_32x32x16_bf16 being
// 16 consecutive voxels of one row.  Both operands arrive as fp32 NCDHW, are cut into bf16 pieces on the way into the LDS
// and are stored voxel-contiguous, so that a lane's fragment (8 voxels of one channel) is one ds_read_b128.
//   block = 32 output x 32 input channels; 8 multiplying waves = 8 of the 9 (kd,kh) with their 3 kw each, the ninth pair's
//           three taps on waves 0..2; 4 staging waves (global loads, cut into pieces, LDS stores), one per SIMD
//   tile  = 4 rows x 16 voxels of dY and the 3 x 6 x 18 halo of X; the columns of tiles are walked along d as one stream of
//           planes with a ring of four X planes and two dY buffers (see the kernel)
//   loads : rows as float4 (W a multiple of 4), two steps ahead in registers; padding read from a zero word so that no load
//           sits under a branch
//   kw    : a lane reads its 8 voxels, the other half's and the edge pairs (three ds_read_b128, conflict-free at the channel
//           pitch) and forms the shifted fragments with five v_alignbit_b32
//   per row of the tile and wave: 2 A + 6 B ds_read_b128 feed 9 MFMAs
// partial[split][o][c][27] as the fp32 kernel: the caller adds the splits up.
#include "common.h"

namespace mvsdet {

typedef short dwb_bf16x8 __attribute__((ext_vector_type(8)));
typedef float dwb_f32x16 __attribute__((ext_vector_type(16)));
typedef unsigned dwb_u32x4 __attribute__((ext_vector_type(4)));
typedef float dwb_f32x4 __attribute__((ext_vector_type(4)));

constexpr int kDbTH = 4, kDbTW = 16;
constexpr int kDbComputeWaves = 8, kDbLoaderWaves = 4;
constexpr int kDbComputeThreads = kDbComputeWaves * 64, kDbLoaders = kDbLoaderWaves * 64;
constexpr int kDbThreads = kDbComputeThreads + kDbLoaders;       // 768
constexpr int kDbXRowB = 48;                                      // [w0..w0+7][w0+8..w0+15][w0-2,w0-1 | w0+16,w0+17][pad 8]
constexpr int kDbXPlaneB = (kDbTH + 2) * kDbXRowB;                // 288
constexpr int kDbXChanB = 4 * kDbXPlaneB + 16;                    // 1168 B = 292 words = 36 mod 64: conflict-free b128 reads
constexpr int kDbXPieceB = 32 * kDbXChanB;
constexpr int kDbYChanB = kDbTH * 32 + 16;                        // 144 B = 36 words: conflict-free b128 reads
constexpr int kDbYPieceB = 32 * kDbYChanB;
constexpr int kDbYBufB = 2 * kDbYPieceB;
constexpr int kDbDepth = 2;                                       // steps of global loads in flight
constexpr int kDbTabCap = 4096;                                   // columns of one split (host: nsplit >= ncols / 4096)
constexpr int kDbTabOff = 2 * kDbXPieceB + 2 * kDbYBufB;          // 93184
constexpr int kDbLdsB = kDbTabOff + kDbTabCap * 8;                // 125952: one block of 12 waves per CU
constexpr int kDbXRoles = 32 * (kDbTH + 2), kDbYRoles = 32 * kDbTH;   // 192 (input channel, halo row), 128 (output channel, row)
static_assert(kDbXRoles <= kDbLoaders && kDbYRoles <= kDbLoaders, "one loader thread per row of either operand");

__device__ float4 g_dwb_zero;   // zero-initialised: the source of every padding element

__device__ __forceinline__ unsigned dwb_pack(__bf16 a, __bf16 b) {
    return (unsigned)__builtin_bit_cast(unsigned short, a) | ((unsigned)__builtin_bit_cast(unsigned short, b) << 16);
}
// two fp32 -> the packed bf16 pair of their leading pieces and of their remainders
__device__ __forceinline__ void dwb_split2(float f0, float f1, unsigned& hi, unsigned& mid) {
    const __bf16 a0 = (__bf16)f0, a1 = (__bf16)f1;                       // round to nearest even
    hi = dwb_pack(a0, a1);
    mid = dwb_pack((__bf16)(f0 - (float)a0), (__bf16)(f1 - (float)a1));  // exact difference, rounded once
}
__device__ __forceinline__ void dwb_split8(const dwb_f32x4& a, const dwb_f32x4& b, uint4& hi, uint4& mid) {
    dwb_split2(a.x, a.y, hi.x, mid.x);
    dwb_split2(a.z, a.w, hi.y, mid.y);
    dwb_split2(b.x, b.y, hi.z, mid.z);
    dwb_split2(b.z, b.w, hi.w, mid.w);
}

// The columns of a split are walked as ONE stream of planes: position q = (column i, plane pos), pos = 0 .. D, where the
// plane pos = D is all zero and serves as plane D of column i and as plane -1 of column i+1.  Every step one X plane (two
// ahead of the one being multiplied) is committed into a ring of four and one dY tile into one of two buffers, the ones two
// steps further are fetched into registers, and planes q-1, q, q+1 are multiplied: one barrier per step, no start-up cost
// per column, two steps of global loads in flight (a step is shorter than the memory latency).
//
// Waves 0..7 only multiply (the zero plane between two columns carries a zero dY tile and is multiplied like any other:
// 1/(D+1) more matrix work, no branch): wave w owns the (kd,kh) pair number w with its three kw, and waves 0..2 one kw each
// of the ninth pair -- a workgroup's waves go to the SIMDs round-robin (w, w+4, w+8 share one: tools/micro/wave_simd.hip), so
// every SIMD multiplies 7, 7, 7 or 6 of the 27 taps.  Waves 8..11 only stage, one per SIMD: a SIMD's vector issue is shared
// by its waves (an MFMA holds it for 8 of its 32 cycles), and with the staging on the multiplying waves a step took the SUM
// of the two (6.3 ms at the conv0 shape for 2.5 ms of staging and 4.4 ms of MFMAs with their fragment reads).  A loader thread
// owns one (input channel, halo row) -- six float4 loads = the 24 voxels w0-4 .. w0+19, of which w0-2 .. w0+17 make the
// 48-byte LDS row -- and / or one (output channel, row) of dY.  The staging step is straight-line: padding and idle roles read
// a zero word (a load whose value is only selected under a condition is moved under a branch with its own s_waitcnt; a branch
// on the role hides from the compiler how many loads are in flight), the conversion is pinned behind the barrier (else hipcc
// converts right after the load to save registers and waits for it there), the stream position advances without a division
// (column table in the LDS).
__global__ __launch_bounds__(kDbThreads) void conv3d_k3_dw_bf16x3_kernel(const float* __restrict__ x, const float* __restrict__ gy,
                                                                         float* __restrict__ partial, int N, int Cin, int Cout,
                                                                         int D, int H, int W, int tiles_w, int tiles_h, int ncols,
                                                                         int nsplit) {
    extern __shared__ uint4 s_dwb[];
    char* const sx = reinterpret_cast<char*>(s_dwb);
    char* const sy = sx + 2 * kDbXPieceB;
    int2* const tab = reinterpret_cast<int2*>(sx + kDbTabOff);   // column i of this split -> (view, tile origin h0 << 16 | w0)
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const bool loader = wave >= kDbComputeWaves;
    const int r32 = lane & 31, hh = lane >> 5;
    const int kd = wave / 3, kh = wave - 3 * kd;   // multiplying waves: the (kd,kh) pair number `wave`
    // The blocks of one split (all channel blocks) read the same dY tiles and, per input-channel block, the same X planes at
    // the same time: they are placed on ONE XCD (block id modulo 8 picks the XCD), so that its L2 serves all but the first.
    const int ncb = (Cin + 31) / 32, nob = (Cout + 31) / 32, J = ncb * nob;
    int split, j;
    if (nsplit % 8 == 0) {
        const int k = blockIdx.x >> 3;
        j = k % J;
        split = (k / J) * 8 + (blockIdx.x & 7);
    } else {
        j = blockIdx.x % J;
        split = blockIdx.x / J;
    }
    const int c0 = (j / nob) * 32, o0 = (j % nob) * 32;
    const size_t vol = (size_t)D * H * W;
    const int HW = H * W;
    const float* const zero = reinterpret_cast<const float*>(&g_dwb_zero);

    const int cols_per_view = tiles_h * tiles_w;
    const int mine = split < ncols ? (ncols - split + nsplit - 1) / nsplit : 0;   // columns of this split (host: <= kDbTabCap)
    const int Q = mine * (D + 1);                                                 // stream positions
    for (int i = tid; i < mine; i += kDbThreads) {
        const int cidx = split + i * nsplit, n = cidx / cols_per_view, t2 = cidx - n * cols_per_view;
        tab[i] = make_int2(n, ((t2 / tiles_w) * kDbTH) << 16 | ((t2 % tiles_w) * kDbTW));
    }
    __syncthreads();

    if (loader) {
        // ---------------------------------------------------------------------------------------------- staging waves
        const int lt = tid - kDbComputeThreads;
        const bool xthread = lt < kDbXRoles, ythread = lt >= kDbLoaders - kDbYRoles;   // threads 128..191 carry both roles
        const int xl = min(lt, kDbXRoles - 1), yl = max(lt - (kDbLoaders - kDbYRoles), 0);
        const int xc = xl / (kDbTH + 2), xrow = xl - xc * (kDbTH + 2);
        const int yo = (yl >> 2) & 31, yrow = yl & 3;
        const int x_lds = xc * kDbXChanB + xrow * kDbXRowB;
        const int y_lds = yo * kDbYChanB + yrow * 32;
        const float* const xchan = x + (size_t)min(c0 + xc, Cin - 1) * vol;
        const float* const ychan = gy + (size_t)min(o0 + yo, Cout - 1) * vol;
        const bool xc_ok = xthread && c0 + xc < Cin, yo_ok = ythread && o0 + yo < Cout;

        struct Regs { dwb_f32x4 xv[6], yv[4]; };
        struct Pos { int i, d; };   // column of the split, plane (d == D: the zero plane)
        Regs sets[kDbDepth];
        // X plane at stream position px and dY tile at position py (a position beyond the stream or a plane D: zeros)
        auto fetch = [&](Regs& g, Pos px, Pos py) {
            {
                const int2 e = tab[min(px.i, mine - 1)];
                const int h0 = e.y >> 16, w0 = e.y & 0xffff;
                const int h = h0 - 1 + xrow;
                const bool ok = (px.i < mine) & (px.d < D) & xc_ok & (h >= 0) & (h < H);   // '&': no short-circuit branches
                const float* row = xchan + (size_t)e.x * Cin * vol + (size_t)min(px.d, D - 1) * HW + min(max(h, 0), H - 1) * W;
#pragma unroll
                for (int k = 0; k < 6; ++k) {
                    const int w = w0 - 4 + 4 * k;
                    g.xv[k] = *reinterpret_cast<const dwb_f32x4*>((ok & (w >= 0) & (w < W)) ? row + w : zero);
                }
            }
            {
                const int2 e = tab[min(py.i, mine - 1)];
                const int h0 = e.y >> 16, w0 = e.y & 0xffff;
                const int h = h0 + yrow;
                const bool ok = (py.i < mine) & (py.d < D) & yo_ok & (h < H);
                const float* row = ychan + (size_t)e.x * Cout * vol + (size_t)min(py.d, D - 1) * HW + min(h, H - 1) * W;
#pragma unroll
                for (int k = 0; k < 4; ++k) {
                    const int w = w0 + 4 * k;
                    g.yv[k] = *reinterpret_cast<const dwb_f32x4*>((ok & (w < W)) ? row + w : zero);
                }
            }
        };
        auto commit = [&](Regs& g, int slot, int ybuf) {
            // the values pass through an opaque statement here, after the barrier (see the kernel's comment)
#pragma unroll
            for (int k = 0; k < 6; ++k) asm volatile("" : "+v"(g.xv[k]));
#pragma unroll
            for (int k = 0; k < 4; ++k) asm volatile("" : "+v"(g.yv[k]));
            uint4 hi, mid;
            if (xthread) {   // whole waves: 8, 9, 10
                char* dx = sx + x_lds + slot * kDbXPlaneB;
                dwb_split8(g.xv[1], g.xv[2], hi, mid);
                *reinterpret_cast<uint4*>(dx) = hi;
                *reinterpret_cast<uint4*>(dx + kDbXPieceB) = mid;
                dwb_split8(g.xv[3], g.xv[4], hi, mid);
                *reinterpret_cast<uint4*>(dx + 16) = hi;
                *reinterpret_cast<uint4*>(dx + kDbXPieceB + 16) = mid;
                // (w0-2, w0-1) = xv[0].zw, (w0+16, w0+17) = xv[5].xy; the 16-byte store also covers the row's padding
                dwb_split8(dwb_f32x4{g.xv[0].z, g.xv[0].w, g.xv[5].x, g.xv[5].y}, dwb_f32x4{0.f, 0.f, 0.f, 0.f}, hi, mid);
                *reinterpret_cast<uint4*>(dx + 32) = hi;
                *reinterpret_cast<uint4*>(dx + kDbXPieceB + 32) = mid;
            }
            if (ythread) {   // whole waves: 10, 11
                char* dy = sy + ybuf * kDbYBufB + y_lds;
                dwb_split8(g.yv[0], g.yv[1], hi, mid);
                *reinterpret_cast<uint4*>(dy) = hi;
                *reinterpret_cast<uint4*>(dy + kDbYPieceB) = mid;
                dwb_split8(g.yv[2], g.yv[3], hi, mid);
                *reinterpret_cast<uint4*>(dy + 16) = hi;
                *reinterpret_cast<uint4*>(dy + kDbYPieceB + 16) = mid;
            }
        };
        auto at = [&](int q) { return Pos{q / (D + 1), q % (D + 1)}; };
        auto next = [&](Pos p) { return p.d == D ? Pos{p.i + 1, 0} : Pos{p.i, p.d + 1}; };

        Pos fx = at(2 + kDbDepth), fy = at(1 + kDbDepth);   // the positions the first step fetches
        if (Q > 0) {
            // plane -1 of the first column (slot 3) is zero; planes 0, 1 and the first dY tile are committed before the loop
#pragma unroll
            for (int k = 0; k < 6; ++k) sets[0].xv[k] = dwb_f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
            for (int k = 0; k < 4; ++k) sets[0].yv[k] = dwb_f32x4{0.f, 0.f, 0.f, 0.f};
            commit(sets[0], 3, 1);
            fetch(sets[0], at(0), at(0));
            commit(sets[0], 0, 0);
            fetch(sets[0], at(1), Pos{mine, 0});
            commit(sets[0], 1, 1);
#pragma unroll
            for (int k = 0; k < kDbDepth; ++k) fetch(sets[k], at(2 + k), at(1 + k));
        }
        auto step = [&](int q, Regs& g) {
            __syncthreads();   // step q-1 fully consumed (its oldest plane and its dY buffer may be replaced); commits of q-1 visible
            commit(g, (q + 2) & 3, (q + 1) & 1);
            fetch(g, fx, fy);
            fy = fx;
            fx = next(fx);
        };
        for (int q = 0; q < Q; q += kDbDepth) {
#pragma unroll
            for (int k = 0; k < kDbDepth; ++k) step(q + k, sets[k]);
        }
        return;
    }

    // -------------------------------------------------------------------------------------------------- multiplying waves
    dwb_f32x16 acc[4];   // the three kw of the wave's (kd,kh) pair; waves 0..2: tap kw = wave of the pair (2,2)
#pragma unroll
    for (int i = 0; i < 4; ++i)
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[i][r] = 0.0f;
    const bool extra = wave < 3;
    const char* const ya = sy + r32 * kDbYChanB + hh * 16;
    const char* const xbase = sx + r32 * kDbXChanB + kh * kDbXRowB;
    const int own_off = hh * 16, oth_off = 16 - hh * 16;
    // the extra tap (2,2,kw = wave): its row of the X plane after the current one, and the one neighbour word it needs
    const char* const xbase22 = sx + r32 * kDbXChanB + 2 * kDbXRowB;
    const int side_off = wave == 0 ? (hh ? 0 : 32) : (hh ? 32 : 16);   // kw = 0: the voxel before own; kw = 2: the one after

    // one row of the tile: 8 ds_read_b128, 10 v_alignbit, 9 MFMAs
    auto row_mfma = [&](const char* xs, const char* xs22, const char* ys, int r) {
        const dwb_bf16x8 a_hi = *reinterpret_cast<const dwb_bf16x8*>(ys + r * 32);
        const dwb_bf16x8 a_mid = *reinterpret_cast<const dwb_bf16x8*>(ys + kDbYPieceB + r * 32);
        dwb_bf16x8 bq[3][2];
#pragma unroll
        for (int p = 0; p < 2; ++p) {
            const char* rowp = xs + p * kDbXPieceB + r * kDbXRowB;
            const dwb_u32x4 own = *reinterpret_cast<const dwb_u32x4*>(rowp + own_off);   // this half's 8 voxels
            dwb_u32x4 oth = *reinterpret_cast<const dwb_u32x4*>(rowp + oth_off);         // the other half's
            dwb_u32x4 edg = *reinterpret_cast<const dwb_u32x4*>(rowp + 32);              // (w0-2,w0-1), (w0+16,w0+17)
            // one word of each is used: keep the whole ds_read_b128 (conflict-free at this channel pitch) -- narrowed
            // to a ds_read_b32 it banks modulo 32 and the 32 channels of a half wave fall on 8 banks
            asm volatile("" : "+v"(oth));
            asm volatile("" : "+v"(edg));
            const unsigned L = hh ? oth.w : edg.x;     // high half = the voxel before own
            const unsigned R = hh ? edg.y : oth.x;     // low half = the voxel after own
            const unsigned t1 = __builtin_amdgcn_alignbit(own.y, own.x, 16), t2 = __builtin_amdgcn_alignbit(own.z, own.y, 16),
                           t3 = __builtin_amdgcn_alignbit(own.w, own.z, 16);
            bq[0][p] = __builtin_bit_cast(dwb_bf16x8, (dwb_u32x4{__builtin_amdgcn_alignbit(own.x, L, 16), t1, t2, t3}));   // w-1
            bq[1][p] = __builtin_bit_cast(dwb_bf16x8, own);                                                               // w
            bq[2][p] = __builtin_bit_cast(dwb_bf16x8, (dwb_u32x4{t1, t2, t3, __builtin_amdgcn_alignbit(R, own.w, 16)}));   // w+1
        }
        // consecutive MFMAs go to different accumulators
#pragma unroll
        for (int kw = 0; kw < 3; ++kw) acc[kw] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a_hi, bq[kw][0], acc[kw], 0, 0, 0);
#pragma unroll
        for (int kw = 0; kw < 3; ++kw) acc[kw] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a_hi, bq[kw][1], acc[kw], 0, 0, 0);
#pragma unroll
        for (int kw = 0; kw < 3; ++kw) acc[kw] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a_mid, bq[kw][0], acc[kw], 0, 0, 0);
        if (extra) {   // wave-uniform
            dwb_bf16x8 be[2];
#pragma unroll
            for (int p = 0; p < 2; ++p) {
                const char* rowp = xs22 + p * kDbXPieceB + r * kDbXRowB;
                const dwb_u32x4 own = *reinterpret_cast<const dwb_u32x4*>(rowp + own_off);
                dwb_u32x4 side = *reinterpret_cast<const dwb_u32x4*>(rowp + side_off);
                asm volatile("" : "+v"(side));
                const unsigned t1 = __builtin_amdgcn_alignbit(own.y, own.x, 16), t2 = __builtin_amdgcn_alignbit(own.z, own.y, 16),
                               t3 = __builtin_amdgcn_alignbit(own.w, own.z, 16);
                dwb_u32x4 v = own;                                                                      // kw = 1
                if (wave == 0) v = dwb_u32x4{__builtin_amdgcn_alignbit(own.x, hh ? side.w : side.x, 16), t1, t2, t3};   // w-1
                if (wave == 2) v = dwb_u32x4{t1, t2, t3, __builtin_amdgcn_alignbit(hh ? side.y : side.x, own.w, 16)};   // w+1
                be[p] = __builtin_bit_cast(dwb_bf16x8, v);
            }
            acc[3] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a_hi, be[0], acc[3], 0, 0, 0);
            acc[3] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a_hi, be[1], acc[3], 0, 0, 0);
            acc[3] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a_mid, be[0], acc[3], 0, 0, 0);
        }
    };
    for (int q = 0; q < (Q + kDbDepth - 1) / kDbDepth * kDbDepth; ++q) {   // as many barriers as the staging waves run
        __syncthreads();   // the loaders' commits of step q-1 are visible
        const char* xs = xbase + ((q + kd + 3) & 3) * kDbXPlaneB;
        const char* xs22 = xbase22 + ((q + 1) & 3) * kDbXPlaneB;   // kd = 2: the plane after the current one
        const char* ys = ya + (q & 1) * kDbYBufB;
#pragma unroll
        for (int r = 0; r < kDbTH; ++r) row_mfma(xs, xs22, ys, r);
    }
    // partial[split][o][c][tap]; C/D map: column = lane & 31 (c), row = (reg & 3) + 8*(reg >> 2) + 4*(lane >> 5) (o)
#pragma unroll
    for (int kw = 0; kw < 3; ++kw) {
        const int t = (kd * 3 + kh) * 3 + kw;
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            const int o = o0 + (r & 3) + 8 * (r >> 2) + 4 * hh, c = c0 + r32;
            if (o < Cout && c < Cin) partial[(((size_t)split * Cout + o) * Cin + c) * 27 + t] = acc[kw][r];
        }
    }
    if (extra) {
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            const int o = o0 + (r & 3) + 8 * (r >> 2) + 4 * hh, c = c0 + r32;
            if (o < Cout && c < Cin) partial[(((size_t)split * Cout + o) * Cin + c) * 27 + 24 + wave] = acc[3][r];
        }
    }
}

}  // namespace mvsdet

using namespace mvsdet;

extern "C" size_t mvsdet_conv3d_k3_dw_partial_bytes(int Cin, int Cout, int nsplit);

extern "C" int mvsdet_conv3d_k3_dw_bf16x3(const float* x, const float* grad_out, float* partial, size_t partial_bytes, int nsplit,
                                          int N, int Cin, int Cout, int D, int H, int W, mvsdet_stream_t stream) {
    MVS_REQUIRE(x && grad_out && partial, "conv3d_k3_dw_bf16x3: NULL pointer");
    MVS_REQUIRE(N > 0 && Cin > 0 && Cout > 0 && D > 0 && H > 0 && W > 0, "conv3d_k3_dw_bf16x3: bad shape");
    MVS_REQUIRE(nsplit > 0 && nsplit <= 65535, "conv3d_k3_dw_bf16x3: nsplit=%d outside [1,65535]", nsplit);
    MVS_REQUIRE((long long)D * H * W < INT32_MAX, "conv3d_k3_dw_bf16x3: one channel volume exceeds 2^31 elements");
    MVS_REQUIRE(W % 4 == 0 && (uintptr_t)x % 16 == 0 && (uintptr_t)grad_out % 16 == 0,
                "conv3d_k3_dw_bf16x3: rows are read as float4 (W=%d must be a multiple of 4, tensors 16-byte aligned)", W);
    if (partial_bytes < mvsdet_conv3d_k3_dw_partial_bytes(Cin, Cout, nsplit)) {
        set_error("conv3d_k3_dw_bf16x3: partial buffer %zu B < %zu B", partial_bytes,
                  mvsdet_conv3d_k3_dw_partial_bytes(Cin, Cout, nsplit));
        return MVSDET_ERR_WORKSPACE;
    }
    const int tiles_w = (W + kDbTW - 1) / kDbTW, tiles_h = (H + kDbTH - 1) / kDbTH;
    const long long ncols = (long long)N * tiles_h * tiles_w;   // (view, h-tile, w-tile) columns, walked along d
    MVS_REQUIRE(ncols < INT32_MAX, "conv3d_k3_dw_bf16x3: too many tiles");
    MVS_REQUIRE((ncols + nsplit - 1) / nsplit <= kDbTabCap, "conv3d_k3_dw_bf16x3: %lld tile columns need nsplit >= %lld", ncols,
                (ncols + kDbTabCap - 1) / kDbTabCap);
    MVS_REQUIRE(H < 32768 && W < 65536, "conv3d_k3_dw_bf16x3: H=%d, W=%d too large", H, W);
    const long long nblocks = (long long)nsplit * ((Cin + 31) / 32) * ((Cout + 31) / 32);
    MVS_REQUIRE(nblocks < INT32_MAX, "conv3d_k3_dw_bf16x3: too many blocks");
    dim3 grid((unsigned)nblocks);
    if (hipFuncSetAttribute(reinterpret_cast<const void*>(conv3d_k3_dw_bf16x3_kernel), hipFuncAttributeMaxDynamicSharedMemorySize,
                            kDbLdsB) != hipSuccess) {
        set_error("conv3d_k3_dw_bf16x3: hipFuncSetAttribute(MaxDynamicSharedMemorySize) failed");
        return MVSDET_ERR_HIP;
    }
    hipLaunchKernelGGL(conv3d_k3_dw_bf16x3_kernel, grid, dim3(kDbThreads), kDbLdsB, (hipStream_t)stream, x, grad_out, partial, N, Cin, Cout,
                       D, H, W, tiles_w, tiles_h, (int)ncols, nsplit);
    MVS_LAUNCH_CHECK("conv3d_k3_dw_bf16x3");
    return MVSDET_OK;
}
