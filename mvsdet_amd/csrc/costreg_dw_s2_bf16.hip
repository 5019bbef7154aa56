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
//           (7,7,7,6 taps per SIMD); 4 staging waves, one per SIMD: global loads two steps ahead in registers, cut, LDS stores.
// partial[split][o][c][27] as the other weight-gradient kernels: the caller adds the splits up.
#include "common.h"

namespace mvsdet {

typedef short ds2_bf16x8 __attribute__((ext_vector_type(8)));
typedef float ds2_f32x4 __attribute__((ext_vector_type(4)));

constexpr int kS2dRows = 4, kS2dW = 8;                    // coarse tile
constexpr int kS2dFRows = 2 * kS2dRows + 1;               // 9 fine rows
constexpr int kS2dCo = 64, kS2dCi = 16;                   // channels of a block: coarse (M, four groups of 16), fine (N)
constexpr int kS2dComputeWaves = 8, kS2dLoaderWaves = 4;
constexpr int kS2dComputeThreads = kS2dComputeWaves * 64, kS2dLoaders = kS2dLoaderWaves * 64;
constexpr int kS2dThreads = kS2dComputeThreads + kS2dLoaders;          // 768
constexpr int kS2dFUnitB = kS2dCi * 16;                                // 256: one (fine row, unit) of 16 channels
constexpr int kS2dFRowB = 3 * kS2dFUnitB;                              // 768
constexpr int kS2dFPieceB = kS2dFRows * kS2dFRowB;                     // 6912: one plane, one piece
constexpr int kS2dFPlaneB = 2 * kS2dFPieceB;
constexpr int kS2dFPairB = 2 * kS2dFPlaneB;                            // 27648
constexpr int kS2dYRowB = kS2dCo * 16;                                 // 1024
constexpr int kS2dYPieceB = kS2dRows * kS2dYRowB;                      // 4096
constexpr int kS2dYBufB = 2 * kS2dYPieceB;
constexpr int kS2dDepth = 2;                                           // steps of global loads in flight
constexpr int kS2dTabCap = 4096;
constexpr int kS2dYOff = 2 * kS2dFPairB;                               // 55296
constexpr int kS2dTabOff = kS2dYOff + 3 * kS2dYBufB;                   // 79872
constexpr int kS2dLdsB = kS2dTabOff + kS2dTabCap * 8;                  // 112640: one block of 12 waves per CU
constexpr int kS2dFRoles = 2 * kS2dFRows * kS2dCi;                     // 288 (plane, fine row, channel)
constexpr int kS2dYRoles = kS2dCo * kS2dRows;                          // 256 (channel, row)
static_assert(kS2dYRoles == kS2dLoaders && kS2dFRoles <= kS2dLoaders + 64, "one Y role per loader thread; the first staging wave carries a second fine role");

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
        const bool first = wave == kS2dComputeWaves;                   // carries the fine roles 256 .. 287 on lanes 0..31
        struct FRole { int ch, pl, frow, lds; const float* chan; bool ok, live; };
        auto frole = [&](int fr, bool live) {
            FRole r;
            r.ch = fr & (kS2dCi - 1);
            const int pr = fr >> 4;
            r.pl = pr >= kS2dFRows;
            r.frow = pr - r.pl * kS2dFRows;
            r.lds = r.pl * kS2dFPlaneB + r.frow * kS2dFRowB + r.ch * 16;
            r.chan = x + (size_t)min(c0 + r.ch, Cin - 1) * fvol;
            r.ok = live && c0 + r.ch < Cin;
            r.live = live;
            return r;
        };
        const FRole fa = frole(lt, true);
        const FRole fb = frole(min(256 + (lt & 63), kS2dFRoles - 1), first && lane < kS2dFRoles - 256);
        const int yo = lt >> 2, yrow = lt & 3;
        const int y_lds = yrow * kS2dYRowB + yo * 16;
        const float* const ychan = gy + (size_t)min(o0 + yo, Cout - 1) * cvol;
        const bool yo_ok = o0 + yo < Cout;

        struct Pos { int i, s; };
        auto at = [&](int q) { return Pos{q / Dc, q % Dc}; };
        auto next = [&](Pos p) { return p.s + 1 == Dc ? Pos{p.i + 1, 0} : Pos{p.i, p.s + 1}; };
        struct Regs { ds2_f32x4 fv[5], yv[2]; };
        struct RegsB { ds2_f32x4 fv[5]; };
        Regs sets[kS2dDepth];
        RegsB setsb[kS2dDepth];

        // fine row of role r at stream position p: x[2w0-4 .. 2w0+15] as five float4 (padding and idle roles: the zero word)
        auto fetch_fine = [&](ds2_f32x4* fv, const FRole& r, Pos p) {
            const int2 e = tab[min(p.i, mine - 1)];
            const int h0 = e.y >> 16, w0 = e.y & 0xffff;
            const int h = 2 * h0 - 1 + r.frow, d = 2 * p.s + r.pl;
            const bool ok = (p.i < mine) & r.ok & (h >= 0) & (h < H);   // '&': no short-circuit branches
            const float* row = r.chan + (size_t)e.x * Cin * fvol + (size_t)d * HW + min(max(h, 0), H - 1) * W;
#pragma unroll
            for (int k = 0; k < 5; ++k) {
                const int w = 2 * w0 - 4 + 4 * k;
                fv[k] = *reinterpret_cast<const ds2_f32x4*>((ok & (w >= 0) & (w < W)) ? row + w : zero);
            }
        };
        auto fetch_y = [&](ds2_f32x4* yv, Pos p) {
            const int2 e = tab[min(p.i, mine - 1)];
            const int h0 = e.y >> 16, w0 = e.y & 0xffff;
            const int h = h0 + yrow;
            const bool ok = (p.i < mine) & yo_ok & (h < Hc);
            const float* row = ychan + (size_t)e.x * Cout * cvol + (size_t)p.s * HWc + min(h, Hc - 1) * Wc;
#pragma unroll
            for (int k = 0; k < 2; ++k) {
                const int w = w0 + 4 * k;
                yv[k] = *reinterpret_cast<const ds2_f32x4*>((ok & (w < Wc)) ? row + w : zero);
            }
        };
        // five float4 -> the even / odd / shifted-odd units of both pieces
        auto commit_fine = [&](ds2_f32x4* fv, const FRole& r, int pair) {
#pragma unroll
            for (int k = 0; k < 5; ++k) asm volatile("" : "+v"(fv[k]));   // conversion pinned behind the barrier (see costreg_dw_bf16.hip)
            uint4 eh, em, oh, om;
            ds2_split2(fv[1].x, fv[1].z, eh.x, em.x);
            ds2_split2(fv[2].x, fv[2].z, eh.y, em.y);
            ds2_split2(fv[3].x, fv[3].z, eh.z, em.z);
            ds2_split2(fv[4].x, fv[4].z, eh.w, em.w);
            ds2_split2(fv[1].y, fv[1].w, oh.x, om.x);
            ds2_split2(fv[2].y, fv[2].w, oh.y, om.y);
            ds2_split2(fv[3].y, fv[3].w, oh.z, om.z);
            ds2_split2(fv[4].y, fv[4].w, oh.w, om.w);
            unsigned lh, lm;                                             // x[2w0-1] in the HIGH half
            ds2_split2(0.0f, fv[0].w, lh, lm);
            const uint4 mh = make_uint4(__builtin_amdgcn_alignbit(oh.x, lh, 16), __builtin_amdgcn_alignbit(oh.y, oh.x, 16),
                                        __builtin_amdgcn_alignbit(oh.z, oh.y, 16), __builtin_amdgcn_alignbit(oh.w, oh.z, 16));
            const uint4 mm = make_uint4(__builtin_amdgcn_alignbit(om.x, lm, 16), __builtin_amdgcn_alignbit(om.y, om.x, 16),
                                        __builtin_amdgcn_alignbit(om.z, om.y, 16), __builtin_amdgcn_alignbit(om.w, om.z, 16));
            char* dst = sf + pair * kS2dFPairB + r.lds;
            *reinterpret_cast<uint4*>(dst) = eh;
            *reinterpret_cast<uint4*>(dst + kS2dFUnitB) = oh;
            *reinterpret_cast<uint4*>(dst + 2 * kS2dFUnitB) = mh;
            *reinterpret_cast<uint4*>(dst + kS2dFPieceB) = em;
            *reinterpret_cast<uint4*>(dst + kS2dFPieceB + kS2dFUnitB) = om;
            *reinterpret_cast<uint4*>(dst + kS2dFPieceB + 2 * kS2dFUnitB) = mm;
        };
        auto commit_y = [&](ds2_f32x4* yv, int buf) {
#pragma unroll
            for (int k = 0; k < 2; ++k) asm volatile("" : "+v"(yv[k]));
            uint4 hi, mid;
            ds2_split2(yv[0].x, yv[0].y, hi.x, mid.x);
            ds2_split2(yv[0].z, yv[0].w, hi.y, mid.y);
            ds2_split2(yv[1].x, yv[1].y, hi.z, mid.z);
            ds2_split2(yv[1].z, yv[1].w, hi.w, mid.w);
            char* dst = sy + buf * kS2dYBufB + y_lds;
            *reinterpret_cast<uint4*>(dst) = hi;
            *reinterpret_cast<uint4*>(dst + kS2dYPieceB) = mid;
        };

        // before the loop: fine pair 0 -> slot 0, Y tiles 0, 1 -> buffers 0, 1; then the loads of the first two steps
        if (steps > 0) {
            fetch_fine(sets[0].fv, fa, at(0));
            fetch_y(sets[0].yv, at(0));
            fetch_y(sets[1].yv, at(1));
            if (first) fetch_fine(setsb[0].fv, fb, at(0));
            commit_fine(sets[0].fv, fa, 0);
            commit_y(sets[0].yv, 0);
            commit_y(sets[1].yv, 1);
            if (first && fb.live) commit_fine(setsb[0].fv, fb, 0);
#pragma unroll
            for (int k = 0; k < kS2dDepth; ++k) {
                fetch_fine(sets[k].fv, fa, at(1 + k));
                fetch_y(sets[k].yv, at(2 + k));
                if (first) fetch_fine(setsb[k].fv, fb, at(1 + k));
            }
        }
        Pos pf = at(1 + kS2dDepth), py = at(2 + kS2dDepth);   // what the first step fetches
        int ybuf = 2;                                         // (q + 2) mod 3
        auto step = [&](int q, Regs& g, RegsB& gb) {
            __syncthreads();   // step q-1 fully consumed; its commits visible
            commit_fine(g.fv, fa, (q + 1) & 1);
            commit_y(g.yv, ybuf);
            if (first && fb.live) commit_fine(gb.fv, fb, (q + 1) & 1);
            ybuf = ybuf == 2 ? 0 : ybuf + 1;
            fetch_fine(g.fv, fa, pf);
            fetch_y(g.yv, py);
            if (first) fetch_fine(gb.fv, fb, pf);
            pf = next(pf);
            py = next(py);
        };
        for (int q = 0; q < steps; q += kS2dDepth) {
#pragma unroll
            for (int k = 0; k < kS2dDepth; ++k) step(q + k, sets[k], setsb[k]);
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
