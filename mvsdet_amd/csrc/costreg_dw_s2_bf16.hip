// Weight gradient of the stride-2 layers of the cost regularisation network (mvs_models/mvsnet.py:77,80: conv1, conv3) and,
// with the two tensors exchanged, of its transposed layers (mvsnet.py:92-100: conv9, conv11), on the bf16 matrix cores with
// three-term split operands (the arithmetic of costreg_bf16.hip / costreg_dw_bf16.hip):
//
//     dW[o][c][kd,kh,kw] = sum over views and COARSE voxels (d,h,w) of  Y[n,o,d,h,w] * X[n,c,2d+kd-1,2h+kh-1,2w+kw-1]
//
// Y = the coarse tensor (N,Co,D/2,H/2,W/2), X = the fine one (N,Ci,D,H,W).  A GEMM per tap with the coarse voxels as the
// reduction, on v_mfma_f32_16x16x32_bf16: the 32 k of one instruction are 4 coarse rows x 8 coarse voxels (k-group =
// lane >> 4 = the row), M = 16 coarse channels, N = 16 fine channels.
//   block : 64 coarse x 16 fine channels -- the FINE tensor is the large one (8 x the voxels), and every block of coarse
//           channels stages it again: 64 coarse channels per block halve that against a 32 x 32 block, and the fine planes
//           of 16 channels fit the LDS twice over (the ring below)
//   tile  : 4 coarse rows x 8 coarse voxels of Y; of X the fine rows 2h0-1 .. 2h0+7 (9) and fine voxels 2w0-1 .. 2w0+15,
//           DE-INTERLEAVED along w while they are cut into bf16 pieces: per fine row three 16-byte units per channel,
//           even[j] = x[2w0+2j] (tap kw=1), odd[j] = x[2w0+2j+1] (kw=2), oddm[j] = x[2w0+2j-1] (kw=0), j = 0..7 -- every
//           tap's fragment is ONE aligned ds_read_b128, no shifting in the multiplying waves
//   LDS   : fine  [pair 2][plane 2][piece 2][fine row 9][unit 3][channel 16][16 B]   channel-fastest: a b128 read of 16
//           coarse [buffer 3][piece 2][row 4][channel 64][16 B]                       channels x any rows is conflict-free
//   stream: the tile columns of a split are walked along the coarse d.  Step (column, s) multiplies the fine planes 2s
//           (kd=1) and 2s+1 (kd=2 with Y[s]; kd=0 with Y[s+1]: fine plane 2(s+1)-1) -- so a column takes exactly D/2 steps,
//           plane -1 is never staged (it is padding) and the waves of kd=0 sit out the last step of a column.
//           Two fine plane pairs (one multiplied, one being committed) and three Y tiles (s, s+1, s+2 being committed).
//   waves : 8 multiplying waves = the (kd,kh) pairs 0..7 with their three kw, the ninth pair's taps one each on waves 0..2
//           (7,7,7,6 taps per SIMD); 8 staging waves (round 5; 4 until then): six stage the fine planes, two the Y tile, global
//           loads three steps ahead in registers, cut, LDS stores.  Their loads are COALESCED: four consecutive lanes read the 64
//           contiguous bytes of a (plane, fine row, channel) -- a thread per row (five float4 of its own) made every lane of a
//           load a 64-byte request of its own, and the address path bounded the kernel: conv1 0.74 -> 0.63 ms, conv3 0.39 -> 0.335.
// partial[split][o][c][27] as the other weight-gradient kernels: the caller adds the splits up.
#include "common.h"

namespace mvsdet {

typedef short ds2_bf16x8 __attribute__((ext_vector_type(8)));
typedef float ds2_f32x4 __attribute__((ext_vector_type(4)));

#ifndef MVS_S2DW_FINE
#define MVS_S2DW_FINE 6
#endif
#ifndef MVS_S2DW_DEPTH
#define MVS_S2DW_DEPTH 3
#endif
constexpr int kS2dRows = 4, kS2dW = 8;                    // coarse tile
constexpr int kS2dFRows = 2 * kS2dRows + 1;               // 9 fine rows
constexpr int kS2dCo = 64, kS2dCi = 16;                   // channels of a block: coarse (M, four groups of 16), fine (N)
// staging waves: six that stage the fine planes (1152 items (plane, fine row, channel, quarter row): three per thread and step) and
// two that stage the Y tile (512 items (channel, row, half row): four per thread and step); 16 waves = four per SIMD, 128 VGPRs
constexpr int kS2dComputeWaves = 8, kS2dFineWaves = MVS_S2DW_FINE, kS2dYWaves = 2, kS2dLoaderWaves = kS2dFineWaves + kS2dYWaves;
constexpr int kS2dComputeThreads = kS2dComputeWaves * 64, kS2dLoaders = kS2dLoaderWaves * 64;
constexpr int kS2dThreads = kS2dComputeThreads + kS2dLoaders;          // 1024
constexpr int kS2dFUnitB = kS2dCi * 16;                                // 256: one (fine row, unit) of 16 channels
constexpr int kS2dFRowB = 3 * kS2dFUnitB;                              // 768
constexpr int kS2dFPieceB = kS2dFRows * kS2dFRowB;                     // 6912: one plane, one piece
constexpr int kS2dFPlaneB = 2 * kS2dFPieceB;
constexpr int kS2dFPairB = 2 * kS2dFPlaneB;                            // 27648
constexpr int kS2dYRowB = kS2dCo * 16;                                 // 1024
constexpr int kS2dYPieceB = kS2dRows * kS2dYRowB;                      // 4096
constexpr int kS2dYBufB = 2 * kS2dYPieceB;
constexpr int kS2dDepth = MVS_S2DW_DEPTH;                              // steps of global loads in flight
constexpr int kS2dTabCap = 4096;
constexpr int kS2dYOff = 2 * kS2dFPairB;                               // 55296
constexpr int kS2dTabOff = kS2dYOff + 3 * kS2dYBufB;                   // 79872
constexpr int kS2dLdsB = kS2dTabOff + kS2dTabCap * 8;                  // 112640: one block of 12 waves per CU
constexpr int kS2dFRoles = 2 * kS2dFRows * kS2dCi;                     // 288 (plane, fine row, channel)
constexpr int kS2dYRoles = kS2dCo * kS2dRows;                          // 256 (channel, row)
static_assert(kS2dFRoles <= kS2dFineWaves * 64 && kS2dYRoles == 2 * kS2dYWaves * 64, "a fine role per thread of the fine waves, two Y rows per thread of the Y waves");

__device__ float4 g_ds2_zero;   // zero-initialised: the source of every padding element

__device__ __forceinline__ unsigned ds2_pack(__bf16 a, __bf16 b) {
    return (unsigned)__builtin_bit_cast(unsigned short, a) | ((unsigned)__builtin_bit_cast(unsigned short, b) << 16);
}
__device__ __forceinline__ void ds2_split2(float f0, float f1, unsigned& hi, unsigned& mid) {
    const __bf16 a0 = (__bf16)f0, a1 = (__bf16)f1;                       // round to nearest even
    hi = ds2_pack(a0, a1);
    mid = ds2_pack((__bf16)(f0 - (float)a0), (__bf16)(f1 - (float)a1));  // exact difference, rounded once
}

__global__ __launch_bounds__(kS2dThreads) void conv3d_k3_s2_dw_bf16x3_kernel(const float* __restrict__ x, const float* __restrict__ gy,
                                                                            float* __restrict__ partial, int N, int Cin, int Cout,
                                                                            int D, int H, int W, int tiles_w, int tiles_h,
                                                                            int ncols, int nsplit) {
    extern __shared__ uint4 s_ds2[];
    char* const sf = reinterpret_cast<char*>(s_ds2);
    char* const sy = sf + kS2dYOff;
    int2* const tab = reinterpret_cast<int2*>(sf + kS2dTabOff);   // column i of this split -> (view, coarse origin h0 << 16 | w0)
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const bool loader = wave >= kS2dComputeWaves;
    const int Dc = D >> 1, Hc = H >> 1, Wc = W >> 1;
    // blocks of one split on ONE XCD (block id modulo 8 picks it): they read the same tiles at the same time
    const int ncb = (Cin + kS2dCi - 1) / kS2dCi, nob = (Cout + kS2dCo - 1) / kS2dCo, J = ncb * nob;
    int split, j;
    if (nsplit % 8 == 0) {
        const int k = blockIdx.x >> 3;
        j = k % J;
        split = (k / J) * 8 + (blockIdx.x & 7);
    } else {
        j = blockIdx.x % J;
        split = blockIdx.x / J;
    }
    const int c0 = (j / nob) * kS2dCi, o0 = (j % nob) * kS2dCo;
    const size_t fvol = (size_t)D * H * W, cvol = (size_t)Dc * Hc * Wc;
    const int HW = H * W, HWc = Hc * Wc;
    const float* const zero = reinterpret_cast<const float*>(&g_ds2_zero);

    const int cols_per_view = tiles_h * tiles_w;
    const int mine = split < ncols ? (ncols - split + nsplit - 1) / nsplit : 0;   // columns of this split (host: <= kS2dTabCap)
    const int Q = mine * Dc;                                                      // stream positions
    const int steps = (Q + kS2dDepth - 1) / kS2dDepth * kS2dDepth;                // barriers both kinds of wave run
    for (int i = tid; i < mine; i += kS2dThreads) {
        const int cidx = split + i * nsplit, n = cidx / cols_per_view, t2 = cidx - n * cols_per_view;
        tab[i] = make_int2(n, ((t2 / tiles_w) * kS2dRows) << 16 | ((t2 % tiles_w) * kS2dW));
    }
    __syncthreads();

    if (loader) {
        // ---------------------------------------------------------------------------------------------- staging waves
        const int lt = tid - kS2dComputeThreads;
        const bool fine_wave = wave < kS2dComputeWaves + kS2dFineWaves;   // wave-uniform: the two kinds run separate loops
        struct Pos { int i, s; };
        auto at = [&](int q) { return Pos{q / Dc, q % Dc}; };
        auto next = [&](Pos p) { return p.s + 1 == Dc ? Pos{p.i + 1, 0} : Pos{p.i, p.s + 1}; };

        if (fine_wave) {
            // COALESCED staging.  An item = (role (plane of the pair, fine row, channel), quarter k of the row's 16 fine voxels
            // x[2w0 .. 2w0+15]): four consecutive lanes read the 64 contiguous, 64-byte-aligned bytes of a role -- with a role per
            // thread (five float4 of ITS row) every lane of a load was a 64-byte request of its own and the address path, not the
            // latency, bounded the kernel (what-if build with contiguous addresses: conv1 0.72 -> 0.48 ms).  The de-interleave
            // needs no exchange: quarter k holds even[2k], even[2k+1] (elements 0, 2), odd[2k], odd[2k+1] (1, 3) and, with the
            // element to its left -- the previous lane's last one by DPP, x[2w0-1] from memory for k = 0 --, oddm[2k], oddm[2k+1]:
            // three aligned 4-byte stores per piece.  1152 items on 384 threads: kS2dFPasses = 3 per step.
            constexpr int kItems = kS2dFRoles * 4, kFThreads = kS2dFineWaves * 64;
            constexpr int kS2dFPasses = (kItems + kFThreads - 1) / kFThreads;     // 3
            int it_lds[kS2dFPasses], it_frow[kS2dFPasses], it_pl[kS2dFPasses], it_k[kS2dFPasses];
            unsigned it_chan[kS2dFPasses];   // element offset of the item's channel inside a view (< 2^31: host check)
            bool it_ok[kS2dFPasses], it_live[kS2dFPasses];
#pragma unroll
            for (int ps = 0; ps < kS2dFPasses; ++ps) {
                const int item = lt + ps * kFThreads;
                it_live[ps] = item < kItems;
                const int role = min(item, kItems - 1) >> 2;
                it_k[ps] = item & 3;
                const int ch = role & (kS2dCi - 1), pr = role >> 4;
                it_pl[ps] = pr >= kS2dFRows;
                it_frow[ps] = pr - it_pl[ps] * kS2dFRows;
                it_lds[ps] = it_pl[ps] * kS2dFPlaneB + it_frow[ps] * kS2dFRowB + ch * 16 + it_k[ps] * 4;
                it_chan[ps] = (unsigned)((size_t)min(c0 + ch, Cin - 1) * fvol);
                it_ok[ps] = it_live[ps] && c0 + ch < Cin;
            }
            struct FSet { ds2_f32x4 v[kS2dFPasses]; float left[kS2dFPasses]; };
            FSet sets[kS2dDepth];
            auto fetch_fine = [&](FSet& g, Pos p) {
                const int2 e = tab[min(p.i, mine - 1)];
                const int h0 = e.y >> 16, w0 = e.y & 0xffff;
                const float* const view = x + (size_t)e.x * Cin * fvol;
#pragma unroll
                for (int ps = 0; ps < kS2dFPasses; ++ps) {
                    const int h = 2 * h0 - 1 + it_frow[ps], d = 2 * p.s + it_pl[ps];
                    const bool ok = (p.i < mine) & it_ok[ps] & (h >= 0) & (h < H);   // '&': no short-circuit branches
                    const float* row = view + it_chan[ps] + (size_t)d * HW + min(max(h, 0), H - 1) * W;
                    const int w = 2 * w0 + 4 * it_k[ps];
                    g.v[ps] = *reinterpret_cast<const ds2_f32x4*>((ok & (w < W)) ? row + w : zero);
                    g.left[ps] = *((ok & (w0 > 0) & (2 * w0 <= W)) ? row + 2 * w0 - 1 : zero);   // x[2w0-1]: used by quarter 0
                }
            };
            auto commit_fine = [&](FSet& g, int pair) {
#pragma unroll
                for (int ps = 0; ps < kS2dFPasses; ++ps) {
                    asm volatile("" : "+v"(g.v[ps]));   // conversion pinned behind the barrier (see costreg_dw_bf16.hip)
                    asm volatile("" : "+v"(g.left[ps]));
                }
#pragma unroll
                for (int ps = 0; ps < kS2dFPasses; ++ps) {
                    // the element to the left of the quarter: the previous lane's last one (same role: lanes 4r .. 4r+3)
                    const float prev = __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(g.v[ps].w), 0x111 /* row_shr:1 */, 0xf, 0xf, true));
                    const float lft = it_k[ps] == 0 ? g.left[ps] : prev;
                    unsigned eh, em, oh, om, mh, mm;
                    ds2_split2(g.v[ps].x, g.v[ps].z, eh, em);
                    ds2_split2(g.v[ps].y, g.v[ps].w, oh, om);
                    ds2_split2(lft, g.v[ps].y, mh, mm);
                    if (it_live[ps]) {
                        char* dst = sf + pair * kS2dFPairB + it_lds[ps];
                        *reinterpret_cast<unsigned*>(dst) = eh;
                        *reinterpret_cast<unsigned*>(dst + kS2dFUnitB) = oh;
                        *reinterpret_cast<unsigned*>(dst + 2 * kS2dFUnitB) = mh;
                        *reinterpret_cast<unsigned*>(dst + kS2dFPieceB) = em;
                        *reinterpret_cast<unsigned*>(dst + kS2dFPieceB + kS2dFUnitB) = om;
                        *reinterpret_cast<unsigned*>(dst + kS2dFPieceB + 2 * kS2dFUnitB) = mm;
                    }
                }
            };
            // before the loop: fine pair 0 -> slot 0; then the loads of the first kS2dDepth steps (step q commits position q + 1)
            if (steps > 0) {
                fetch_fine(sets[0], at(0));
                commit_fine(sets[0], 0);
#pragma unroll
                for (int k = 0; k < kS2dDepth; ++k) fetch_fine(sets[k], at(1 + k));
            }
            Pos pf = at(1 + kS2dDepth);
            for (int q = 0; q < steps; q += kS2dDepth) {
#pragma unroll
                for (int k = 0; k < kS2dDepth; ++k) {
                    __syncthreads();   // step q+k-1 fully consumed; its commits visible
                    commit_fine(sets[k], (q + k + 1) & 1);
                    fetch_fine(sets[k], pf);
                    pf = next(pf);
                }
            }
            return;
        }

        // the Y tile, coalesced the same way: item = (role (channel, row), half of the row's 8 coarse voxels): two lanes read a role's
        // 32 contiguous bytes and store 8 bytes per piece; 512 items on 128 threads: four per step
        constexpr int kYThreads = kS2dYWaves * 64, kS2dYPasses = 2 * kS2dYRoles / kYThreads;   // 4
        const int yt = lt - kS2dFineWaves * 64;
        int y_lds[kS2dYPasses], y_row[kS2dYPasses], y_half[kS2dYPasses];
        unsigned y_chan[kS2dYPasses];
        bool y_ok[kS2dYPasses];
#pragma unroll
        for (int ps = 0; ps < kS2dYPasses; ++ps) {
            const int item = yt + ps * kYThreads, role = item >> 1;
            const int yo = role >> 2;
            y_half[ps] = item & 1;
            y_row[ps] = role & 3;
            y_lds[ps] = y_row[ps] * kS2dYRowB + yo * 16 + y_half[ps] * 8;
            y_chan[ps] = (unsigned)((size_t)min(o0 + yo, Cout - 1) * cvol);
            y_ok[ps] = o0 + yo < Cout;
        }
        ds2_f32x4 ysets[kS2dDepth][kS2dYPasses];
        auto fetch_y = [&](ds2_f32x4* yv, Pos p) {
            const int2 e = tab[min(p.i, mine - 1)];
            const int h0 = e.y >> 16, w0 = e.y & 0xffff;
            const float* const view = gy + (size_t)e.x * Cout * cvol;
#pragma unroll
            for (int ps = 0; ps < kS2dYPasses; ++ps) {
                const int h = h0 + y_row[ps], w = w0 + 4 * y_half[ps];
                const bool ok = (p.i < mine) & y_ok[ps] & (h < Hc) & (w < Wc);
                const float* row = view + y_chan[ps] + (size_t)p.s * HWc + min(h, Hc - 1) * Wc;
                yv[ps] = *reinterpret_cast<const ds2_f32x4*>(ok ? row + w : zero);
            }
        };
        auto commit_y = [&](ds2_f32x4* yv, int buf) {
#pragma unroll
            for (int ps = 0; ps < kS2dYPasses; ++ps) asm volatile("" : "+v"(yv[ps]));
#pragma unroll
            for (int ps = 0; ps < kS2dYPasses; ++ps) {
                uint2 hi, mid;
                ds2_split2(yv[ps].x, yv[ps].y, hi.x, mid.x);
                ds2_split2(yv[ps].z, yv[ps].w, hi.y, mid.y);
                char* dst = sy + buf * kS2dYBufB + y_lds[ps];
                *reinterpret_cast<uint2*>(dst) = hi;
                *reinterpret_cast<uint2*>(dst + kS2dYPieceB) = mid;
            }
        };
        // before the loop: Y tiles 0, 1 -> buffers 0, 1; then the loads of the first kS2dDepth steps (step q commits position q + 2)
        if (steps > 0) {
            fetch_y(ysets[0], at(0));
            fetch_y(ysets[1], at(1));
            commit_y(ysets[0], 0);
            commit_y(ysets[1], 1);
#pragma unroll
            for (int k = 0; k < kS2dDepth; ++k) fetch_y(ysets[k], at(2 + k));
        }
        Pos py = at(2 + kS2dDepth);
        int ybuf = 2;                                         // (q + 2) mod 3
        for (int q = 0; q < steps; q += kS2dDepth) {
#pragma unroll
            for (int k = 0; k < kS2dDepth; ++k) {
                __syncthreads();
                commit_y(ysets[k], ybuf);
                ybuf = ybuf == 2 ? 0 : ybuf + 1;
                fetch_y(ysets[k], py);
                py = next(py);
            }
        }
        return;
    }

    // -------------------------------------------------------------------------------------------------- multiplying waves
    typedef float f32x4 __attribute__((ext_vector_type(4)));
    f32x4 acc[4][4];   // [the three kw of the wave's (kd,kh) pair; waves 0..2: tap kw = wave of the pair (2,2)][coarse channel group]
#pragma unroll
    for (int t = 0; t < 4; ++t)
#pragma unroll
        for (int a = 0; a < 4; ++a) acc[t][a] = f32x4{0.f, 0.f, 0.f, 0.f};
    const int kd = wave / 3, kh = wave - 3 * kd;
    const bool extra = wave < 3;
    const int l16 = lane & 15, g = lane >> 4;
    const int pl = kd != 1;                                     // kd = 1: the even plane of the pair; kd = 0, 2: the odd one
    const char* const fbase = sf + pl * kS2dFPlaneB + (2 * g + kh) * kS2dFRowB + l16 * 16;
    const char* const fbase22 = sf + kS2dFPlaneB + (2 * g + 2) * kS2dFRowB + (wave == 0 ? 2 : wave == 1 ? 0 : 1) * kS2dFUnitB + l16 * 16;
    const char* const ybase = sy + g * kS2dYRowB + l16 * 16;
    constexpr int kUnit[3] = {2, 0, 1};                         // kw -> oddm, even, odd

    int s = 0, yb = 0;                                          // plane of the column, q mod 3
    for (int q = 0; q < steps; ++q) {
        __syncthreads();   // the loaders' commits of step q-1 are visible
        const int yb1 = yb == 2 ? 0 : yb + 1;
        const char* const fs = fbase + (q & 1) * kS2dFPairB;
        const char* const ycur = ybase + yb * kS2dYBufB;
        if (kd != 0 || s + 1 < Dc) {                            // wave-uniform
            const char* const ys = kd == 0 ? ybase + yb1 * kS2dYBufB : ycur;
            ds2_bf16x8 ah[4], am[4], bh[3], bm[3];
#pragma unroll
            for (int a = 0; a < 4; ++a) {
                ah[a] = *reinterpret_cast<const ds2_bf16x8*>(ys + a * 256);
                am[a] = *reinterpret_cast<const ds2_bf16x8*>(ys + kS2dYPieceB + a * 256);
            }
#pragma unroll
            for (int kw = 0; kw < 3; ++kw) {
                bh[kw] = *reinterpret_cast<const ds2_bf16x8*>(fs + kUnit[kw] * kS2dFUnitB);
                bm[kw] = *reinterpret_cast<const ds2_bf16x8*>(fs + kS2dFPieceB + kUnit[kw] * kS2dFUnitB);
            }
            // consecutive MFMAs go to different accumulators
#pragma unroll
            for (int kw = 0; kw < 3; ++kw)
#pragma unroll
                for (int a = 0; a < 4; ++a) acc[kw][a] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(ah[a], bh[kw], acc[kw][a], 0, 0, 0);
#pragma unroll
            for (int kw = 0; kw < 3; ++kw)
#pragma unroll
                for (int a = 0; a < 4; ++a) acc[kw][a] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(ah[a], bm[kw], acc[kw][a], 0, 0, 0);
#pragma unroll
            for (int kw = 0; kw < 3; ++kw)
#pragma unroll
                for (int a = 0; a < 4; ++a) acc[kw][a] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(am[a], bh[kw], acc[kw][a], 0, 0, 0);
        }
        if (extra) {                                            // tap (2,2,kw = wave): odd plane, Y[s]
            const char* const fe = fbase22 + (q & 1) * kS2dFPairB;
            ds2_bf16x8 ah[4], am[4];
#pragma unroll
            for (int a = 0; a < 4; ++a) {
                ah[a] = *reinterpret_cast<const ds2_bf16x8*>(ycur + a * 256);
                am[a] = *reinterpret_cast<const ds2_bf16x8*>(ycur + kS2dYPieceB + a * 256);
            }
            const ds2_bf16x8 bh = *reinterpret_cast<const ds2_bf16x8*>(fe);
            const ds2_bf16x8 bm = *reinterpret_cast<const ds2_bf16x8*>(fe + kS2dFPieceB);
#pragma unroll
            for (int a = 0; a < 4; ++a) acc[3][a] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(ah[a], bh, acc[3][a], 0, 0, 0);
#pragma unroll
            for (int a = 0; a < 4; ++a) acc[3][a] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(ah[a], bm, acc[3][a], 0, 0, 0);
#pragma unroll
            for (int a = 0; a < 4; ++a) acc[3][a] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(am[a], bh, acc[3][a], 0, 0, 0);
        }
        s = s + 1 == Dc ? 0 : s + 1;
        yb = yb1;
    }
    // partial[split][o][c][tap]; C/D map of the 16x16 tile: column = lane & 15 (c), row = 4 * (lane >> 4) + reg (o)
    const int c = c0 + l16;
#pragma unroll
    for (int t = 0; t < 4; ++t) {
        if (t == 3 && !extra) break;
        const int tap = t < 3 ? (kd * 3 + kh) * 3 + t : 24 + wave;
#pragma unroll
        for (int a = 0; a < 4; ++a)
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const int o = o0 + a * 16 + 4 * g + r;
                if (o < Cout && c < Cin) partial[(((size_t)split * Cout + o) * Cin + c) * 27 + tap] = acc[t][a][r];
            }
    }
}

}  // namespace mvsdet

using namespace mvsdet;

extern "C" size_t mvsdet_conv3d_k3_dw_partial_bytes(int Cin, int Cout, int nsplit);

extern "C" int mvsdet_conv3d_k3_s2_dw_bf16x3(const float* x, const float* grad_out, float* partial, size_t partial_bytes, int nsplit,
                                             int N, int Cin, int Cout, int D, int H, int W, mvsdet_stream_t stream) {
    MVS_REQUIRE(x && grad_out && partial, "conv3d_k3_s2_dw_bf16x3: NULL pointer");
    MVS_REQUIRE(N > 0 && Cin > 0 && Cout > 0 && D > 0 && H > 0 && W > 0, "conv3d_k3_s2_dw_bf16x3: bad shape");
    MVS_REQUIRE(D % 2 == 0 && H % 2 == 0, "conv3d_k3_s2_dw_bf16x3: D=%d, H=%d must be even", D, H);
    MVS_REQUIRE(nsplit > 0 && nsplit <= 65535, "conv3d_k3_s2_dw_bf16x3: nsplit=%d outside [1,65535]", nsplit);
    MVS_REQUIRE((long long)D * H * W < INT32_MAX, "conv3d_k3_s2_dw_bf16x3: one channel volume exceeds 2^31 elements");
    // the staging waves keep a channel's offset inside its view as 32 bits
    MVS_REQUIRE((long long)Cin * D * H * W < (1LL << 32) && (long long)Cout * (D / 2) * (H / 2) * (W / 2) < (1LL << 32),
                "conv3d_k3_s2_dw_bf16x3: one view of either tensor exceeds 2^32 elements");
    MVS_REQUIRE(W % 8 == 0 && (uintptr_t)x % 16 == 0 && (uintptr_t)grad_out % 16 == 0,
                "conv3d_k3_s2_dw_bf16x3: rows of both tensors are read as float4 (W=%d must be a multiple of 8, tensors 16-byte aligned)", W);
    if (partial_bytes < mvsdet_conv3d_k3_dw_partial_bytes(Cin, Cout, nsplit)) {
        set_error("conv3d_k3_s2_dw_bf16x3: partial buffer %zu B < %zu B", partial_bytes,
                  mvsdet_conv3d_k3_dw_partial_bytes(Cin, Cout, nsplit));
        return MVSDET_ERR_WORKSPACE;
    }
    const int Hc = H / 2, Wc = W / 2;
    const int tiles_w = (Wc + kS2dW - 1) / kS2dW, tiles_h = (Hc + kS2dRows - 1) / kS2dRows;
    const long long ncols = (long long)N * tiles_h * tiles_w;   // (view, h-tile, w-tile) columns, walked along the coarse d
    MVS_REQUIRE(ncols < INT32_MAX, "conv3d_k3_s2_dw_bf16x3: too many tiles");
    MVS_REQUIRE((ncols + nsplit - 1) / nsplit <= kS2dTabCap, "conv3d_k3_s2_dw_bf16x3: %lld tile columns need nsplit >= %lld", ncols,
                (ncols + kS2dTabCap - 1) / kS2dTabCap);
    MVS_REQUIRE(H < 65536 && W < 65536, "conv3d_k3_s2_dw_bf16x3: H=%d, W=%d too large", H, W);
    const long long nblocks = (long long)nsplit * ((Cin + kS2dCi - 1) / kS2dCi) * ((Cout + kS2dCo - 1) / kS2dCo);
    MVS_REQUIRE(nblocks < INT32_MAX, "conv3d_k3_s2_dw_bf16x3: too many blocks");
    if (hipFuncSetAttribute(reinterpret_cast<const void*>(conv3d_k3_s2_dw_bf16x3_kernel), hipFuncAttributeMaxDynamicSharedMemorySize,
                            kS2dLdsB) != hipSuccess) {
        set_error("conv3d_k3_s2_dw_bf16x3: hipFuncSetAttribute(MaxDynamicSharedMemorySize) failed");
        return MVSDET_ERR_HIP;
    }
    hipLaunchKernelGGL(conv3d_k3_s2_dw_bf16x3_kernel, dim3((unsigned)nblocks), dim3(kS2dThreads), kS2dLdsB, (hipStream_t)stream, x,
                       grad_out, partial, N, Cin, Cout, D, H, W, tiles_w, tiles_h, (int)ncols, nsplit);
    MVS_LAUNCH_CHECK("conv3d_k3_s2_dw_bf16x3");
    return MVSDET_OK;
}
