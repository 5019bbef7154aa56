// 3x3x3 convolutions of the cost regularisation network (mvs_models/mvsnet.py:76-82,104-108) on the bf16 matrix cores
// with fp32 operands cut into bf16 pieces ("bf16x3"):
//
//     x = x_hi + x_mid (+ 2^-16 |x|),  w = w_hi + w_mid (+ ...),   x*w ~ x_hi*w_hi + x_hi*w_mid + x_mid*w_hi
//
// Every product of two bf16 pieces is exact in fp32 and v_mfma_f32_32x32x16_bf16 accumulates in fp32, so the three
// MFMAs leave out only the terms of relative size 2^-16 (x_mid*w_mid, x_lo*w_hi, x_hi*w_lo): measured on the network
// (tools/study/split_bf16_emulation.py, G8 fixture) 2e-6 .. 4e-6 on the logits, 5e-7 on the depth probabilities -- the
// north_star bar is 1e-4 -- where plain bf16 operands give 1e-3.  Three bf16 MFMAs cost 3/16 of the fp32 MFMA
// (v_mfma_f32_32x32x2_f32 runs at the vector rate: MI355X_MICROARCH.md, Matrix cores).
//
// Data layout ("split channel-last", SCL): a producer cuts the activations once,
//     xs[piece][n][c8][dp][hp][wp][8]   bf16, piece = hi | mid, c8 = channel / 8, (dp,hp,wp) = (d,h,w) + 1,
// with a zero border of one voxel (and up to a tile beyond D, H, W) around every volume, so the convolution reads
// whole halo tiles without a bounds test.  One voxel's 8 channels of one piece are 16 bytes: one lane of an LDS-DMA
// (global_load_lds_dwordx4), one ds_read_b128, one MFMA B fragment.
//
// Implicit GEMM, D[o][v] += A[o][k] * B[k][v]:  the 16 k of an MFMA are (tap parity, 8 channels): lane half h = lane >> 5
// reads tap 2p + h of tap pair p -- the 27 taps make 14 pairs, the last one half empty (zero weights): 27/28 of the
// matrix work is useful and a stage is 8 channels deep, which is what lets two stages of input fit the LDS.
//   block  = (view, TD x TH x TW output voxels, 64 output channels), TD*TH*TW/64 waves: 4x8x16 (8 waves), 4x12x16 (12: 60 rows =
//            5 tiles, three waves on every SIMD); the fp32-input form also 3x16x8 (6) and 8x8x8 (8), the stride-2 and transposed
//            kernels 3x16x8 -- whichever pads the volume least (6x30x40 and 3x15x20 are padded 1.7x and 2.3x by 4x8x16)
//   wave   = 64 voxels (two column groups of 32 = 2 h-rows x 16 or 4 h-rows x 8) x 64 channels (two row groups) = 4 accumulators 32x32
//   LDS    = input halo tile of 8 channels, both pieces, double buffered  +  the weights of 5 tap pairs x 64 x 8,
//            both pieces, double buffered (three weight sub-stages per 8 channels); all of it filled by LDS-DMA
//   per tap pair and wave: 4 A + 4 B ds_read_b128 feed 12 MFMAs (2 x 2 accumulators x 3 terms)
// One barrier per weight sub-stage: wait for the own DMAs of the stage, barrier, issue the DMAs of the next stage into
// the buffers the barrier has just freed, compute.
#include "common.h"

#include <algorithm>

namespace mvsdet {

typedef short bf16x8 __attribute__((ext_vector_type(8)));
typedef float f32x16b __attribute__((ext_vector_type(16)));

constexpr int kBfW = 16;             // tile width (voxels along w): one MFMA column group = 2 h-rows of 16
constexpr int kBfPairs = 14;         // tap pairs (27 taps + 1 empty)
constexpr int kBfSubPairs = 5;       // tap pairs per weight sub-stage (5 + 5 + 4); the kernels take it as a template parameter:
                                     // 2 (seven sub-stages) makes the LDS of a 3x16x8 tile 77 KiB, so that two blocks share a CU
__host__ __device__ constexpr int bf_w_slots(int subp) { return subp * 2 * 2 * 64; }   // 16-byte slots of one weight buffer: [pair][row group][piece][lane]

__host__ __device__ constexpr int bf_in_slots(int TD, int TH, int TW = kBfW) {   // 16-byte slots of one piece of one input buffer
    return ((TD + 2) * (TH + 2) * (TW + 2) + 63) / 64 * 64;
}
__host__ __device__ constexpr size_t bf_lds_bytes(int TD, int TH, int TW = kBfW, int subp = kBfSubPairs) {
    return (size_t)(2 * 2 * bf_in_slots(TD, TH, TW) + 2 * bf_w_slots(subp)) * 16;
}

// Which output channel an accumulator register holds.  The 32x32 MFMA leaves row (r & 3) + 8 * (r >> 2) + 4 * hh of a row group in
// register r of lane half hh = lane >> 5; the weights are laid out (split_conv_weight_kernel) so that row carries channel
//     8 * (2 * (r >> 3) + hh) + (r & 7)
// of the group: registers 8q .. 8q+7 of a lane are then EIGHT CONSECUTIVE channels = one 16-byte unit of the SCL form, and a
// layer can hand its output to the next one already cut into bf16 pieces (one uint4 store per piece and unit instead of eight
// 4-byte stores, no packing pass, and the consumer feeds its LDS by DMA).
__host__ __device__ constexpr int bf_row_channel(int r, int hh) { return 8 * (2 * (r >> 3) + hh) + (r & 7); }
__host__ __device__ constexpr int bf_mfma_row_channel(int m) {   // the same map from the MFMA row m = lane & 31 of an A fragment
    return 8 * ((m >> 4) * 2 + ((m >> 2) & 1)) + (m & 3) + 4 * ((m >> 3) & 1);
}

// Where a convolution leaves its result: any combination of the fp32 (N,Cout,D,H,W) tensor, its SCL form and its
// parity-split SCL form ("PSCL": the eight (d,h,w)-parity classes of the volume as eight compact SCL volumes, what a
// stride-2 consumer reads tile by tile: conv3d_k3_s2_bf16x3_kernel).
struct BfOut {
    float* f32;            // or nullptr
    uint4* scl;            // or nullptr; [piece][n][c8][Dp][Hp][Wp]
    uint4* pscl;           // or nullptr; [piece][class 8][n][c8][cDp][cHp][cWp]
    int N;                 // views (the class index of the PSCL form is outside the view index)
    int Dp, Hp, Wp;        // padded extents of the SCL form
    int cDp, cHp, cWp;     // padded extents of one parity class
    size_t piece, cpiece;  // units per piece (SCL), per piece (PSCL: 8 classes)
};

// Blocks are dealt to the 8 XCDs round-robin by their linear id.  With `xcd_map` XCD x takes the x-th EIGHTH of the grid in its
// natural order instead of every eighth block: neighbouring tiles -- which share their halo voxels -- and the blocks of output
// channels that read the same tile meet in one L2 (conv2 of the cost network 0.81 -> 0.74 ms).  Host: the grid is a multiple of 8.
__device__ __forceinline__ void xcd_block_id(int xcd_map, unsigned& bx, unsigned& by, unsigned& bz) {
    bx = blockIdx.x; by = blockIdx.y; bz = blockIdx.z;
    if (xcd_map) {
        const unsigned total = gridDim.x * gridDim.y * gridDim.z;
        const unsigned lin = bx + gridDim.x * (by + gridDim.y * bz);
        const unsigned moved = (lin & 7u) * (total >> 3) + (lin >> 3);
        bx = moved % gridDim.x;
        by = (moved / gridDim.x) % gridDim.y;
        bz = moved / (gridDim.x * gridDim.y);
    }
}
// fp32 x 8 -> the two bf16 pieces (round to nearest even; the remainder is exact in fp32 and rounded once)
__device__ __forceinline__ void bf_cut8(const float (&v)[8], uint4& hi, uint4& mid) {
    unsigned h[4], m[4];
#pragma unroll
    for (int j = 0; j < 4; ++j) {
        unsigned short hb[2], mb[2];
#pragma unroll
        for (int e = 0; e < 2; ++e) {
            const __bf16 a = (__bf16)v[2 * j + e];
            const __bf16 b = (__bf16)(v[2 * j + e] - (float)a);
            hb[e] = __builtin_bit_cast(unsigned short, a);
            mb[e] = __builtin_bit_cast(unsigned short, b);
        }
        h[j] = (unsigned)hb[0] | ((unsigned)hb[1] << 16);
        m[j] = (unsigned)mb[0] | ((unsigned)mb[1] << 16);
    }
    hi = make_uint4(h[0], h[1], h[2], h[3]);
    mid = make_uint4(m[0], m[1], m[2], m[3]);
}

// The 8 finished values (channels 8*c8 .. 8*c8+7 of view n at voxel (d,h,w)) into the SCL and / or PSCL form.
__device__ __forceinline__ void bf_store_units(const BfOut& o, const float (&v)[8], int n, int C8o, int c8, int d, int h, int w) {
    uint4 hi, mid;
    bf_cut8(v, hi, mid);
    if (o.scl) {
        const size_t u = (((size_t)n * C8o + c8) * o.Dp + (d + 1)) * o.Hp * o.Wp + (size_t)(h + 1) * o.Wp + (w + 1);
        o.scl[u] = hi;
        o.scl[o.piece + u] = mid;
    }
    if (o.pscl) {
        const int cls = ((d & 1) << 2) | ((h & 1) << 1) | (w & 1);
        const size_t u = ((((size_t)cls * o.N + n) * C8o + c8) * o.cDp + ((d >> 1) + 1)) * o.cHp * o.cWp +
                         (size_t)((h >> 1) + 1) * o.cWp + ((w >> 1) + 1);
        o.pscl[u] = hi;
        o.pscl[o.cpiece + u] = mid;
    }
}

// fp32 NCDHW -> SCL (both pieces), interior voxels only: the border stays as the caller zeroed it.
// thread = one voxel x 8 channels: 8 coalesced channel-row reads, two 16-byte stores.
__global__ __launch_bounds__(kThreads) void scl_pack_kernel(const float* __restrict__ x, long long sN, long long sC, long long sD,
                                                            long long sH, uint4* __restrict__ xs, int C, int C8, int D, int H, int W,
                                                            int Dp, int Hp, int Wp, size_t piece_stride) {
    const size_t vol = (size_t)D * H * W;
    const size_t v = (size_t)blockIdx.x * kThreads + threadIdx.x;
    if (v >= vol) return;
    const int c8 = blockIdx.y, n = blockIdx.z;
    const int d = (int)(v / ((size_t)H * W)), r = (int)(v - (size_t)d * H * W), h = r / W, w = r - h * W;
    // element (n,c,d,h,w) at x[n*sN + c*sC + d*sD + h*sH + w]: a pitched cost volume (rows padded to whole 128-byte lines)
    // is read in place
    const float* src = x + (size_t)n * sN + (size_t)c8 * 8 * sC + (size_t)d * sD + (size_t)h * sH + w;
    unsigned hi[4], mid[4];
#pragma unroll
    for (int j = 0; j < 4; ++j) {
        float f[2];
#pragma unroll
        for (int e = 0; e < 2; ++e) f[e] = (c8 * 8 + 2 * j + e < C) ? src[(size_t)(2 * j + e) * sC] : 0.0f;
        unsigned short hb[2], mb[2];
#pragma unroll
        for (int e = 0; e < 2; ++e) {
            const __bf16 a = (__bf16)f[e];               // round to nearest even (v_cvt_pk_bf16_f32)
            const __bf16 b = (__bf16)(f[e] - (float)a);  // exact difference, rounded once
            hb[e] = __builtin_bit_cast(unsigned short, a);
            mb[e] = __builtin_bit_cast(unsigned short, b);
        }
        hi[j] = (unsigned)hb[0] | ((unsigned)hb[1] << 16);
        mid[j] = (unsigned)mb[0] | ((unsigned)mb[1] << 16);
    }
    const size_t o = (((size_t)n * C8 + c8) * Dp + (d + 1)) * Hp * Wp + (size_t)(h + 1) * Wp + (w + 1);
    xs[o] = make_uint4(hi[0], hi[1], hi[2], hi[3]);
    xs[piece_stride + o] = make_uint4(mid[0], mid[1], mid[2], mid[3]);
}

// The same cut over the PADDED volume: a thread whose voxel lies in the border stores zeros, so the buffer needs no history
// (zero_border = 2: a buffer fresh from an allocator, nothing cleared beforehand; interior values bit for bit those above).
__global__ __launch_bounds__(kThreads) void scl_pack_padded_kernel(const float* __restrict__ x, long long sN, long long sC, long long sD,
                                                                   long long sH, uint4* __restrict__ xs, int C, int C8, int D, int H,
                                                                   int W, int Dp, int Hp, int Wp, size_t piece_stride) {
    const size_t volp = (size_t)Dp * Hp * Wp;
    const size_t v = (size_t)blockIdx.x * kThreads + threadIdx.x;
    if (v >= volp) return;
    const int c8 = blockIdx.y, n = blockIdx.z;
    const int dp = (int)(v / ((size_t)Hp * Wp)), r = (int)(v - (size_t)dp * Hp * Wp), hp = r / Wp, wp = r - hp * Wp;
    const int d = dp - 1, h = hp - 1, w = wp - 1;
    unsigned hi[4] = {0u, 0u, 0u, 0u}, mid[4] = {0u, 0u, 0u, 0u};
    if (d >= 0 && d < D && h >= 0 && h < H && w >= 0 && w < W) {
        const float* src = x + (size_t)n * sN + (size_t)c8 * 8 * sC + (size_t)d * sD + (size_t)h * sH + w;
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            unsigned short hb[2], mb[2];
#pragma unroll
            for (int e = 0; e < 2; ++e) {
                const float f = (c8 * 8 + 2 * j + e < C) ? src[(size_t)(2 * j + e) * sC] : 0.0f;
                const __bf16 a = (__bf16)f;
                const __bf16 b = (__bf16)(f - (float)a);
                hb[e] = __builtin_bit_cast(unsigned short, a);
                mb[e] = __builtin_bit_cast(unsigned short, b);
            }
            hi[j] = (unsigned)hb[0] | ((unsigned)hb[1] << 16);
            mid[j] = (unsigned)mb[0] | ((unsigned)mb[1] << 16);
        }
    }
    const size_t o = ((size_t)n * C8 + c8) * volp + v;
    xs[o] = make_uint4(hi[0], hi[1], hi[2], hi[3]);
    xs[piece_stride + o] = make_uint4(mid[0], mid[1], mid[2], mid[3]);
}

// Conv3d weight (Cout,Cin,27) fp32 -> [Cout/64][c8][14][2 row groups][2 pieces][64 lanes][8] bf16 (see the entry point; lane =
// 32 * (half of the pair) + MFMA row m, which carries output channel bf_mfma_row_channel(m) of its row group):
// thread = one 16-byte unit.  A few hundred thousand elements: run on every call, so the kernel never multiplies a stale
// copy of weights that were updated in place.
struct TapTable { signed char t[2 * kBfPairs]; };   // the 3x3x3 tap (0..26) of every half of the 14 tap pairs; -1 = empty half

// 16x16x32 layout (order 3): [Cout/64][c8][k-step 7][row group 4][piece 2][64 lanes][8]: lane = 16 * (tap of the k-step) + row m,
// row m of row group rg = channel 32*(rg >> 1) + 8*(m >> 2) + 4*(rg & 1) + (m & 3), tap = 4*ks + (lane >> 4) (27: empty)
__device__ __forceinline__ void bf_m16_unit(size_t u, int C8, int& o, int& t, int& c8) {
    const int lane = (int)(u & 63), rg = (int)((u >> 7) & 3), m = lane & 15;
    const size_t r = u >> 9;
    const int ks = (int)(r % 7), ob = (int)(r / ((size_t)7 * C8));
    c8 = (int)((r / 7) % C8);
    o = ob * 64 + 32 * (rg >> 1) + 8 * (m >> 2) + 4 * (rg & 1) + (m & 3);
    t = 4 * ks + (lane >> 4);
    if (t > 26) t = -1;
}

__global__ __launch_bounds__(kThreads) void split_conv_weight_kernel(const float* __restrict__ w, uint4* __restrict__ out, int Cin,
                                                                     int C8, size_t units, TapTable taps, long long so, long long sc,
                                                                     int m16) {
    const size_t u = (size_t)blockIdx.x * kThreads + threadIdx.x;
    if (u >= units) return;
    int lane = (int)(u & 63), piece = (int)((u >> 6) & 1), a = (int)((u >> 7) & 1);
    const size_t r = u >> 8;
    int p = (int)(r % kBfPairs), c8 = (int)((r / kBfPairs) % C8), ob = (int)(r / ((size_t)kBfPairs * C8));
    int o = ob * 64 + a * 32 + bf_mfma_row_channel(lane & 31), t = taps.t[2 * p + (lane >> 5)];
    if (m16) bf_m16_unit(u, C8, o, t, c8);
    unsigned short b[8];
#pragma unroll
    for (int j = 0; j < 8; ++j) {
        const int c = c8 * 8 + j;
        // element (output o, input c, tap t) at w[o*so + c*sc + t]: (Cout,Cin,27) for Conv3d, (Cin,Cout,27) for ConvTranspose3d
        const float f = (t >= 0 && c < Cin) ? w[(size_t)o * so + (size_t)c * sc + t] : 0.0f;
        const __bf16 hi = (__bf16)f;
        const __bf16 v = piece ? (__bf16)(f - (float)hi) : hi;
        b[j] = __builtin_bit_cast(unsigned short, v);
    }
    out[u] = make_uint4((unsigned)b[0] | ((unsigned)b[1] << 16), (unsigned)b[2] | ((unsigned)b[3] << 16),
                        (unsigned)b[4] | ((unsigned)b[5] << 16), (unsigned)b[6] | ((unsigned)b[7] << 16));
}

// All weight tensors of a network in ONE launch (the cost network has seven: a launch each was 0.07 ms of kernels and twice that
// of launch gaps per scene, and caching the pieces instead would miss an in-place update through `.data`).
constexpr int kSplitBatchMax = 8;
struct SplitBatch {
    const float* w[kSplitBatchMax];
    uint4* out[kSplitBatchMax];
    unsigned long long first[kSplitBatchMax + 1];   // first unit of tensor i in the launch's flat unit index
    int Cin[kSplitBatchMax], order[kSplitBatchMax], Cout[kSplitBatchMax];
    TapTable taps[3];
    int n;
};
__global__ __launch_bounds__(kThreads) void split_conv_weight_batched_kernel(SplitBatch b) {
    const size_t g = (size_t)blockIdx.x * kThreads + threadIdx.x;
    if (g >= b.first[b.n]) return;
    int i = 0;
#pragma unroll
    for (int k = 1; k < kSplitBatchMax; ++k)
        if (k < b.n && g >= b.first[k]) i = k;
    const size_t u = g - b.first[i];
    const int Cin = b.Cin[i], C8 = (Cin + 7) / 8, order = b.order[i];
    const long long so = order == 2 ? 27 : (long long)Cin * 27, sc = order == 2 ? (long long)b.Cout[i] * 27 : 27;
    const int lane = (int)(u & 63), piece = (int)((u >> 6) & 1), a = (int)((u >> 7) & 1);
    const size_t r = u >> 8;
    const int p = (int)(r % kBfPairs), ob = (int)(r / ((size_t)kBfPairs * C8));
    int c8 = (int)((r / kBfPairs) % C8);
    int o = ob * 64 + a * 32 + bf_mfma_row_channel(lane & 31), t = b.taps[order == 3 ? 0 : order].t[2 * p + (lane >> 5)];
    if (order == 3) bf_m16_unit(u, C8, o, t, c8);
    const float* w = b.w[i];
    unsigned short bb[8];
#pragma unroll
    for (int j = 0; j < 8; ++j) {
        const int c = c8 * 8 + j;
        const float f = (t >= 0 && c < Cin) ? w[(size_t)o * so + (size_t)c * sc + t] : 0.0f;
        const __bf16 hi = (__bf16)f;
        const __bf16 v = piece ? (__bf16)(f - (float)hi) : hi;
        bb[j] = __builtin_bit_cast(unsigned short, v);
    }
    b.out[i][u] = make_uint4((unsigned)bb[0] | ((unsigned)bb[1] << 16), (unsigned)bb[2] | ((unsigned)bb[3] << 16),
                             (unsigned)bb[4] | ((unsigned)bb[5] << 16), (unsigned)bb[6] | ((unsigned)bb[7] << 16));
}

// tap t of the 3x3x3 kernel as an offset in halo voxels; tap 27 (the empty half of pair 13) aliases tap 26
template <int HH, int HW = kBfW + 2>
__host__ __device__ constexpr int bf_tap_off(int t) {
    const int u = t > 26 ? 26 : t;
    return ((u / 9) * HH + (u / 3) % 3) * HW + u % 3;
}

// F32IN: the input is the fp32 (N,C,D,H,W) tensor itself (element strides sN, sC, sD, sH; w stride 1 -- a row-pitched cost
// volume included) instead of its SCL form: every thread fetches the 8 channels of its 2-3 halo voxels of the NEXT channel
// group into registers while the current group is multiplied (coalesced along w), cuts them into the two bf16 pieces and
// writes the same LDS image the DMA route fills -- no packing pass and no second copy of the activation in HBM (the 2.4 GB
// variance volume: 1.0 ms of packing and a 3 GB buffer).  Out-of-volume halo voxels are zeros by predicate.
// TW: tile width.  16: an MFMA column group (32 voxels) = 2 h-rows of 16; 8 (fp32-input form only): 4 h-rows of 8 -- the tiles
// 3 x 16 x 8 and 8 x 8 x 8 fit the half- and quarter-resolution volumes of the cost network (6 x 30 x 40, 3 x 15 x 20) and the
// neck's 40 x 40 x 16 level, which 4 x 8 x 16 tiles pad 1.7x, 2.3x and 1.2x.
// M16: v_mfma_f32_16x16x32_bf16 instead of v_mfma_f32_32x32x16_bf16.  The 32 k of one instruction are FOUR taps of 8 channels
// (lane group lane >> 4 = tap 4*ks + group of k-step ks; 27 taps = 7 k-steps, one slot empty), a wave's 64 voxels x 64 channels
// are 4 x 4 accumulators of 16 x 16, per k-step 8 A + 8 B fragments feed 48 instructions: the same LDS reads, matrix cycles and
// sums per output as the 32 x 32 form, but MFMA-dense loops hold a higher clock on this shape (MI355X_MICROARCH.md, DVFS
// give-back item 7: ~1.12 - 1.15 x the FLOP/s at equal cycles per FLOP).  Weights: split order 3 (k-step-major, SUBP = 6 "pairs"
// = 3 k-steps per sub-stage: 3 + 3 + 1).
// CGN (16x16x32 form): column groups of 16 voxels per wave.  4: a wave = 64 voxels; 2: 32 voxels, twice the waves -- the
// 3 x 16 x 8 tile then runs on 12 waves (three on every SIMD) instead of 6 (2/2/1/1 on the one block a CU holds).
template <int TD, int TH, bool F32IN, int TW = kBfW, int SUBP = kBfSubPairs, bool M16 = false, int CGN = 4>
__global__ __launch_bounds__(TD * TH * TW * 4 / CGN) void conv3d_k3_bf16x3_kernel(
    const uint4* __restrict__ xs, const float* __restrict__ xf, long long sN, long long sC, long long sD, long long sH, int Cin,
    const uint4* __restrict__ wq, const float* __restrict__ scale, const float* __restrict__ shift,
    const float* __restrict__ residual, BfOut dst, int C8, int Cout, int D, int H, int W, int Dp, int Hp, int Wp,
    size_t piece_stride, int tiles_w, int relu, int nsplit, float* __restrict__ partial, size_t total, int xcd_map,
    double2* __restrict__ stats, const float* __restrict__ stats_pivot) {
    float* __restrict__ out = dst.f32;
    constexpr int RG = 32 / TW;                           // h-rows of one column group
    static_assert(TH % RG == 0 && (TD * TH * TW) % 64 == 0, "whole column groups, two per wave");
    static_assert(CGN == 4 || (M16 && CGN == 2), "half-size waves: 16x16x32 form only");
    constexpr int NW = TD * TH * TW / (16 * CGN);         // waves
    constexpr int NT = 64 * NW;                           // threads
    constexpr int HD = TD + 2, HH = TH + 2, HW = TW + 2;
    constexpr int NVOX = HD * HH * HW;
    constexpr int INS = bf_in_slots(TD, TH, TW);          // slots per piece
    constexpr int IN_DMA = INS / 64;                      // wave-instructions per piece and stage
    constexpr int IN_PER_WAVE = (2 * IN_DMA + NW - 1) / NW;
    constexpr int NSUB = (kBfPairs + SUBP - 1) / SUBP;    // weight sub-stages per channel group
    constexpr int kBfWSlots = bf_w_slots(SUBP);
    static_assert(NSUB >= 2, "the fp32 fetch / cut needs two sub-stages");
    static_assert(!M16 || SUBP % 2 == 0, "16x16x32: a k-step is two pair slots of weights");
    extern __shared__ uint4 s_bf[];   // [2 stages][2 pieces][INS] input, then [2 stages][kBfWSlots] weights
    uint4* s_in = s_bf;
    uint4* s_w = s_bf + 2 * 2 * INS;

    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    unsigned bx, by, bz;
    xcd_block_id(xcd_map, bx, by, bz);
    const int bw = bx % tiles_w, bh = bx / tiles_w;
    // bz = (view, block of 64 output channels, split of the channel groups): a small volume (the 3-D neck's 20x20x8 and
    // 10x10x4 levels: 1-12 tiles) is split over the input channels so that the grid fills the chip; each split writes raw
    // partial sums that splitk_epilogue_kernel adds up (ascending split order) before the affine, residual and ReLU
    const int nob = Cout / 64;
    const int split = bz % nsplit, zo = bz / nsplit;
    const int n = zo / nob, ob64 = zo % nob;
    const int c8_begin = (int)((long long)C8 * split / nsplit), c8_end = (int)((long long)C8 * (split + 1) / nsplit);
    const int w0 = bw * TW, h0 = bh * TH, d0 = by * TD;
    const int col = lane & 31, hh = lane >> 5;

    // ---- DMA plans.  Input: wave-instruction i of a stage = (piece i / IN_DMA, slots 64*(i % IN_DMA) ..); the lane's slot
    // is a halo voxel (dz, hy, wx) -> its address in the padded volume (slots beyond the halo re-read voxel 0).
    const size_t c8_stride = (size_t)Dp * Hp * Wp;
    const uint4* xn = xs + ((size_t)n * C8) * c8_stride + ((size_t)d0 * Hp + h0) * Wp + w0;
    unsigned in_src[IN_PER_WAVE];   // 16-byte units relative to xn (piece and channel group added per stage)
#pragma unroll
    for (int k = 0; k < IN_PER_WAVE; ++k) {
        const int i = wave + k * NW;
        const int slot = (i % IN_DMA) * 64 + lane;
        const int sv = slot < NVOX ? slot : 0;
        const int dz = sv / (HH * HW), r = sv - dz * (HH * HW), hy = r / HW, wx = r - hy * HW;
        in_src[k] = (unsigned)(((size_t)dz * Hp + hy) * Wp + wx);
    }
    auto dma_input = [&](int c8, int buf) {
#pragma unroll
        for (int k = 0; k < IN_PER_WAVE; ++k) {
            const int i = wave + k * NW;   // wave-uniform
            if (i < 2 * IN_DMA) {
                const int piece = i / IN_DMA;
                const uint4* src = xn + (size_t)piece * piece_stride + (size_t)c8 * c8_stride + in_src[k];
                uint4* dst = s_in + (size_t)(buf * 2 + piece) * INS + (i % IN_DMA) * 64;
                __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)src,
                                                 (__attribute__((address_space(3))) void*)dst, 16, 0, 0);
            }
        }
    };
    // ---- fp32 input (F32IN): voxel slots tid, tid + NT, .. of the halo tile; offset inside one channel's volume or -1
    constexpr int NV = F32IN ? (NVOX + NT - 1) / NT : 1;
    int f_off[NV];
    float f_reg[NV][8];
    const float* xfn = F32IN ? xf + (size_t)n * sN : nullptr;
    if constexpr (F32IN) {
#pragma unroll
        for (int k = 0; k < NV; ++k) {
            const int slot = tid + k * NT;
            const int dz = slot / (HH * HW), r = slot - dz * (HH * HW), hy = r / HW, wx = r - hy * HW;
            const int d = d0 + dz - 1, h = h0 + hy - 1, w = w0 + wx - 1;
            const bool ok = slot < NVOX && d >= 0 && d < D && h >= 0 && h < H && w >= 0 && w < W;
            f_off[k] = ok ? (int)((long long)d * sD + (long long)h * sH + w) : -1;
        }
    }
    // ("load or zero" per element: measured faster than unconditional loads from a clamped address with the zeroing moved to
    // the cut -- 4.79 against 4.91 ms at conv0 -- although hipcc branches around every such load)
    auto fetch_f32 = [&](int c8) {
        if constexpr (F32IN) {
#pragma unroll
            for (int k = 0; k < NV; ++k)
#pragma unroll
                for (int j = 0; j < 8; ++j) {
                    const int c = c8 * 8 + j;
                    f_reg[k][j] = (f_off[k] >= 0 && c < Cin) ? xfn[(size_t)c * sC + f_off[k]] : 0.0f;
                }
        }
    };
    auto stage_f32 = [&](int buf) {
        if constexpr (F32IN) {
#pragma unroll
            for (int k = 0; k < NV; ++k) {
                const int slot = tid + k * NT;
                unsigned hi[4], mid[4];
#pragma unroll
                for (int j = 0; j < 4; ++j) {
                    unsigned short hb[2], mb[2];
#pragma unroll
                    for (int e = 0; e < 2; ++e) {
                        const float f = f_reg[k][2 * j + e];
                        const __bf16 a = (__bf16)f;
                        const __bf16 b = (__bf16)(f - (float)a);
                        hb[e] = __builtin_bit_cast(unsigned short, a);
                        mb[e] = __builtin_bit_cast(unsigned short, b);
                    }
                    hi[j] = (unsigned)hb[0] | ((unsigned)hb[1] << 16);
                    mid[j] = (unsigned)mb[0] | ((unsigned)mb[1] << 16);
                }
                if (slot < NVOX) {
                    s_in[(size_t)(buf * 2 + 0) * INS + slot] = make_uint4(hi[0], hi[1], hi[2], hi[3]);
                    s_in[(size_t)(buf * 2 + 1) * INS + slot] = make_uint4(mid[0], mid[1], mid[2], mid[3]);
                }
            }
        }
    };
    // Weights: wq[ob64][c8][pair 14][row group 2][piece 2][lane 64] in 16-byte units; sub-stage s = pairs 5s .. 5s+4
    const uint4* wn = wq + (size_t)ob64 * C8 * (kBfPairs * 4 * 64);
    auto dma_weights = [&](int c8, int s, int buf) {
        const int ninstr = min(SUBP, kBfPairs - s * SUBP) * 4;
        const uint4* src0 = wn + ((size_t)c8 * kBfPairs + s * SUBP) * (4 * 64) + lane;
        for (int i = wave; i < ninstr; i += NW) {
            uint4* dst = s_w + (size_t)buf * kBfWSlots + i * 64;
            __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)(src0 + i * 64),
                                             (__attribute__((address_space(3))) void*)dst, 16, 0, 0);
        }
    };

    // ---- the wave's two column groups: g = 2*wave + b, plane g / (TH/RG), h-rows RG*(g % (TH/RG)) + {0..RG-1}; column c of a
    // group = voxel (row c / TW, w = c % TW)
    int vb[2];
#pragma unroll
    for (int b = 0; b < 2; ++b) {
        const int g = 2 * wave + b;
        const int dz = g / (TH / RG), hy = RG * (g % (TH / RG)) + col / TW;
        vb[b] = (dz * HH + hy) * HW + col % TW;
    }
    // M16: four column groups of 16 voxels, g16 = 4*wave + cg: RG16 = 16 / TW h-rows each; lane = (voxel lane & 15, tap group lane >> 4)
    constexpr int RG16 = 16 / TW > 0 ? 16 / TW : 1;
    const int col16 = lane & 15, kg = lane >> 4;
    int vb16[CGN], toffs[7];
#pragma unroll
    for (int cg = 0; cg < CGN; ++cg) {
        const int g = CGN * wave + cg;
        const int dz = g / (TH / RG16), hy = RG16 * (g % (TH / RG16)) + col16 / TW;
        vb16[cg] = (dz * HH + hy) * HW + col16 % TW;
    }
#pragma unroll
    for (int ks = 0; ks < 7; ++ks)
        toffs[ks] = kg == 0 ? bf_tap_off<HH, HW>(4 * ks) : kg == 1 ? bf_tap_off<HH, HW>(4 * ks + 1)
                  : kg == 2 ? bf_tap_off<HH, HW>(4 * ks + 2) : bf_tap_off<HH, HW>(4 * ks + 3);

    f32x16b acc[2][2];   // [row group (32 output channels)][column group]
    typedef float f32x4b __attribute__((ext_vector_type(4)));
    f32x4b acc16[4][CGN];  // M16: [row group (16 output channels)][column group (16 voxels)]
#pragma unroll
    for (int a = 0; a < 2; ++a)
#pragma unroll
        for (int b = 0; b < 2; ++b)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[a][b][r] = 0.0f;
#pragma unroll
    for (int a = 0; a < 4; ++a)
#pragma unroll
        for (int b = 0; b < CGN; ++b)
#pragma unroll
            for (int r = 0; r < 4; ++r) acc16[a][b][r] = 0.0f;

    const bf16x8* s_in8 = reinterpret_cast<const bf16x8*>(s_in);
    const bf16x8* s_w8 = reinterpret_cast<const bf16x8*>(s_w);

    auto compute = [&](auto sc, int ibuf, int wbuf) {
        constexpr int s = decltype(sc)::value;
        constexpr int np = (kBfPairs - s * SUBP < SUBP ? kBfPairs - s * SUBP : SUBP);
        const bf16x8* bin = s_in8 + (size_t)(ibuf * 2) * INS;
        const bf16x8* ain = s_w8 + (size_t)wbuf * kBfWSlots + lane;
        if constexpr (M16) {
#pragma unroll
            for (int kl = 0; kl < np / 2; ++kl) {
                const int toff = toffs[s * (SUBP / 2) + kl];
                bf16x8 A[4][2], B[CGN][2];   // [row / column group][piece]
#pragma unroll
                for (int a = 0; a < 4; ++a)
#pragma unroll
                    for (int q = 0; q < 2; ++q) A[a][q] = ain[((kl * 4 + a) * 2 + q) * 64];
#pragma unroll
                for (int b = 0; b < CGN; ++b)
#pragma unroll
                    for (int q = 0; q < 2; ++q) B[b][q] = bin[(size_t)q * INS + vb16[b] + toff];
#pragma unroll
                for (int a = 0; a < 4; ++a)
#pragma unroll
                    for (int b = 0; b < CGN; ++b) {
                        acc16[a][b] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(A[a][1], B[b][0], acc16[a][b], 0, 0, 0);   // w_mid * x_hi
                        acc16[a][b] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(A[a][0], B[b][1], acc16[a][b], 0, 0, 0);   // w_hi * x_mid
                        acc16[a][b] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(A[a][0], B[b][0], acc16[a][b], 0, 0, 0);   // w_hi * x_hi
                    }
            }
        } else {
#pragma unroll
            for (int pl = 0; pl < np; ++pl) {
                const int p = s * SUBP + pl;
                const int toff = hh ? bf_tap_off<HH, HW>(2 * p + 1) : bf_tap_off<HH, HW>(2 * p);
                bf16x8 A[2][2], B[2][2];   // [row / column group][piece]
#pragma unroll
                for (int a = 0; a < 2; ++a)
#pragma unroll
                    for (int q = 0; q < 2; ++q) A[a][q] = ain[((pl * 2 + a) * 2 + q) * 64];
#pragma unroll
                for (int b = 0; b < 2; ++b)
#pragma unroll
                    for (int q = 0; q < 2; ++q) B[b][q] = bin[(size_t)q * INS + vb[b] + toff];
#pragma unroll
                for (int a = 0; a < 2; ++a)
#pragma unroll
                    for (int b = 0; b < 2; ++b) {
                        acc[a][b] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(A[a][1], B[b][0], acc[a][b], 0, 0, 0);   // w_mid * x_hi
                        acc[a][b] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(A[a][0], B[b][1], acc[a][b], 0, 0, 0);   // w_hi * x_mid
                        acc[a][b] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(A[a][0], B[b][0], acc[a][b], 0, 0, 0);   // w_hi * x_hi
                    }
            }
        }
    };

    // ---- pipeline over the flat stage index q = 3*c8 + s
    if (c8_begin < c8_end) {
        if constexpr (F32IN) {
            fetch_f32(c8_begin);
            stage_f32(0);
            __builtin_amdgcn_s_waitcnt(0xc07f);   // lgkmcnt(0): this wave's LDS writes of the first input stage are done
        } else {
            dma_input(c8_begin, 0);
        }
        dma_weights(c8_begin, 0, 0);
    }
    // one sub-stage: wait for the own transfers, barrier, start the next sub-stage's transfers, multiply
    auto substage = [&](auto sc, int c8, int ibuf, int q) {
        constexpr int s = decltype(sc)::value;
        if constexpr (s < NSUB) {
            const int wbuf = q & 1;
            // this wave's DMAs (and, F32IN, register fetches) of the current stage have landed; after the barrier
            // everybody's have, and everybody is done reading the buffers the next stage's transfers (issued right below)
            // overwrite
            __builtin_amdgcn_s_waitcnt(0x0f70);   // vmcnt(0) only (expcnt 7, lgkmcnt 15 untouched)
            __builtin_amdgcn_s_barrier();
            if (s + 1 < NSUB) {
                dma_weights(c8, s + 1, wbuf ^ 1);
            } else if (c8 + 1 < c8_end) {
                dma_weights(c8 + 1, 0, wbuf ^ 1);
            }
            if constexpr (F32IN) {
                // next channel group: global -> registers during sub-stage 0, registers -> LDS (cut into pieces) at the
                // top of sub-stage 1; the LDS reads of sub-stage 1's first tap pair wait for lgkmcnt(0) behind the writes
                if (s == 0 && c8 + 1 < c8_end) fetch_f32(c8 + 1);
                if (s == 1 && c8 + 1 < c8_end) stage_f32(ibuf ^ 1);
            } else {
                if (s == 0 && c8 + 1 < c8_end) dma_input(c8 + 1, ibuf ^ 1);
            }
            compute(sc, ibuf, wbuf);
        }
    };
    for (int c8 = c8_begin; c8 < c8_end; ++c8) {
        const int ibuf = (c8 - c8_begin) & 1;
        const int q0 = (c8 - c8_begin) * NSUB;
        substage(std::integral_constant<int, 0>{}, c8, ibuf, q0);
        substage(std::integral_constant<int, 1>{}, c8, ibuf, q0 + 1);
        substage(std::integral_constant<int, 2>{}, c8, ibuf, q0 + 2);
        substage(std::integral_constant<int, 3>{}, c8, ibuf, q0 + 3);
        substage(std::integral_constant<int, 4>{}, c8, ibuf, q0 + 4);
        substage(std::integral_constant<int, 5>{}, c8, ibuf, q0 + 5);
        substage(std::integral_constant<int, 6>{}, c8, ibuf, q0 + 6);
    }

    // ---- epilogue: C/D map of the 32x32 MFMA: column = lane & 31 (voxel), register r of lane half hh = channel bf_row_channel(r, hh)
    const size_t plane = (size_t)H * W, vol = (size_t)D * plane;
    if constexpr (M16) {
        // 16x16: column = lane & 15 (voxel), register r of lane group kg = row 4*kg + r of the row group; the weights are laid out
        // so that row m of row group rg carries channel 32*(rg >> 1) + 8*(m >> 2) + 4*(rg & 1) + (m & 3): row groups 2q and 2q+1
        // together give a lane the eight consecutive channels 32q + 8kg .. + 7
        // stats (training-mode BatchNorm behind this layer, costreg_bn.hip): the lane's sums of its outputs and of their squares,
        // per channel, over its CGN voxels -- reduced over the block below, so that the BatchNorm needs no pass of its own over the
        // tensor for its statistics
        // (sums of v - pivot_c: E[v^2] - mean^2 on raw values cancels in fp32 when |mean| >> spread; the pivot is any value near the
        // channel's mean -- the BatchNorm's running mean -- and is requested here, ahead of the stores)
        float st_s[2][8], st_q[2][8], st_p[2][8];
#pragma unroll
        for (int q = 0; q < 2; ++q)
#pragma unroll
            for (int j = 0; j < 8; ++j) {
                st_s[q][j] = st_q[q][j] = 0.0f;
                st_p[q][j] = (stats && stats_pivot) ? stats_pivot[ob64 * 64 + 32 * q + 8 * kg + j] : 0.0f;
            }
#pragma unroll
        for (int cg = 0; cg < CGN; ++cg) {
            const int g = CGN * wave + cg;
            const int d = d0 + g / (TH / RG16), h = h0 + RG16 * (g % (TH / RG16)) + col16 / TW, w = w0 + col16 % TW;
            if (d >= D || h >= H || w >= W) continue;
#pragma unroll
            for (int q = 0; q < 2; ++q) {
                float sc[8], sh[8], rv[8], v[8];
                const int o0 = ob64 * 64 + 32 * q + 8 * kg;
                const size_t idx0 = ((size_t)n * Cout + o0) * vol + (size_t)d * plane + (size_t)h * W + w;
                const bool fin = nsplit == 1;
#pragma unroll
                for (int j = 0; j < 8; ++j) {
                    sc[j] = (fin && scale) ? scale[o0 + j] : 1.0f;
                    sh[j] = (fin && scale) ? shift[o0 + j] : 0.0f;
                    rv[j] = (fin && residual) ? residual[idx0 + (size_t)j * vol] : 0.0f;
                }
#pragma unroll
                for (int j = 0; j < 8; ++j) {
                    v[j] = j < 4 ? acc16[2 * q][cg][j & 3] : acc16[2 * q + 1][cg][j & 3];
                    if (nsplit > 1) {
                        partial[(size_t)split * total + idx0 + (size_t)j * vol] = v[j];
                        continue;
                    }
                    if (scale) v[j] = fmaf(v[j], sc[j], sh[j]);
                    if (residual) v[j] = v[j] + rv[j];
                    if (relu) v[j] = fmaxf(v[j], 0.0f);
                    if (out) out[idx0 + (size_t)j * vol] = v[j];
                }
                if (nsplit == 1 && (dst.scl || dst.pscl)) bf_store_units(dst, v, n, Cout / 8, ob64 * 8 + 4 * q + kg, d, h, w);
                if (stats) {
#pragma unroll
                    for (int j = 0; j < 8; ++j) {
                        const float dv = v[j] - st_p[q][j];
                        st_s[q][j] += dv;
                        st_q[q][j] = fmaf(dv, dv, st_q[q][j]);
                    }
                }
            }
        }
        if (stats) {   // block-uniform; the launcher allows it with nsplit == 1 and a raw fp32 output only
            // channel 32q + 8kg + j of the block has NW * 16 contributors (wave, voxel lane): all of them into the LDS the loop has
            // left, then four threads per channel add theirs up in double, in a fixed order (the same bits on every run)
            constexpr int NCON = NW * 16;
            static_assert(NT >= 256 && (size_t)64 * NCON * sizeof(float2) <= bf_lds_bytes(TD, TH, TW, SUBP), "the reduction's LDS image");
            float2* red = reinterpret_cast<float2*>(s_bf);
            __syncthreads();   // every wave is past its last read of the stage buffers
#pragma unroll
            for (int q = 0; q < 2; ++q)
#pragma unroll
                for (int j = 0; j < 8; ++j) red[(32 * q + 8 * kg + j) * NCON + wave * 16 + col16] = make_float2(st_s[q][j], st_q[q][j]);
            __syncthreads();
            if (tid < 256) {
                const int ch = tid >> 2, part = tid & 3;
                double a = 0.0, b = 0.0;
                for (int i = part * (NCON / 4); i < (part + 1) * (NCON / 4); ++i) {
                    const float2 e = red[ch * NCON + i];
                    a += (double)e.x;
                    b += (double)e.y;
                }
                a += __shfl_xor(a, 1, 64); b += __shfl_xor(b, 1, 64);
                a += __shfl_xor(a, 2, 64); b += __shfl_xor(b, 2, 64);
                if (part == 0) {
                    // partial sums of channel c: [c][view][tile]: one entry per block that holds outputs of the channel
                    const size_t tiles = (size_t)gridDim.x * gridDim.y;
                    const size_t bidx = ((size_t)n * gridDim.y + by) * gridDim.x + bx;
                    stats[(size_t)(ob64 * 64 + ch) * ((size_t)(gridDim.z / nob) * tiles) + bidx] = make_double2(a, b);
                }
            }
        }
        return;
    }
#pragma unroll
    for (int b = 0; b < 2; ++b) {
        const int g = 2 * wave + b;
        const int d = d0 + g / (TH / RG), h = h0 + RG * (g % (TH / RG)) + col / TW, w = w0 + col % TW;
        if (d >= D || h >= H || w >= W) continue;
#pragma unroll
        for (int a = 0; a < 2; ++a) {
            // affine and residual of 8 outputs (registers 8q .. 8q+7 = eight consecutive channels) are requested together, ahead
            // of the stores: loads and stores share one in-order counter, so a load issued behind a store waits for the store's
            // round trip as well (one element at a time this epilogue was a chain of 64 memory round trips per lane)
#pragma unroll
            for (int q = 0; q < 2; ++q) {
                float sc[8], sh[8], rv[8], v[8];
                const int o0 = ob64 * 64 + a * 32 + bf_row_channel(8 * q, hh);   // channels o0 .. o0 + 7
                const size_t idx0 = ((size_t)n * Cout + o0) * vol + (size_t)d * plane + (size_t)h * W + w;
                const bool fin = nsplit == 1;
#pragma unroll
                for (int j = 0; j < 8; ++j) {
                    sc[j] = (fin && scale) ? scale[o0 + j] : 1.0f;
                    sh[j] = (fin && scale) ? shift[o0 + j] : 0.0f;
                    rv[j] = (fin && residual) ? residual[idx0 + (size_t)j * vol] : 0.0f;
                }
#pragma unroll
                for (int j = 0; j < 8; ++j) {
                    v[j] = acc[a][b][8 * q + j];
                    if (nsplit > 1) {
                        partial[(size_t)split * total + idx0 + (size_t)j * vol] = v[j];
                        continue;
                    }
                    if (scale) v[j] = fmaf(v[j], sc[j], sh[j]);
                    if (residual) v[j] = v[j] + rv[j];
                    if (relu) v[j] = fmaxf(v[j], 0.0f);
                    if (out) out[idx0 + (size_t)j * vol] = v[j];
                }
                // the same values cut into bf16 pieces: one 16-byte unit per piece
                if (nsplit == 1 && (dst.scl || dst.pscl)) bf_store_units(dst, v, n, Cout / 8, ob64 * 8 + a * 4 + 2 * q + hh, d, h, w);
            }
        }
    }
}

// ---------------------------------------------------------------------------------------------------------------
// Stride-2 convolution (mvsnet.py:77,79: conv1 64 -> 128, conv3 128 -> 256; kernel 3, padding 1) on the same machinery.
// Output voxel o reads input 2o + k - 1 per dimension: k = 1 reads the EVEN input plane at coarse index o, k = 0 / k = 2
// the ODD plane at o - 1 / o.  So the input splits into 8 parity classes pi = (pd, ph, pw), each a stride-1 grid of the
// output's size, and the layer is the sum of 8 small stride-1 convolutions (1, 2, 2, 2, 4, 4, 4, 8 taps = 27) into the same
// accumulators.  A stage = (8 channels, one parity class): its tile of the class ((4+1) x (8+1) x (16+1) voxels) is
// fetched from the fp32 tensor (every second element along each odd/even axis), cut into bf16 pieces and written to LDS
// one stage ahead, the class's 1, 1, 2 or 4 tap pairs of weights arrive by LDS-DMA; 14 tap pairs per 8 channels, exactly
// the stride-1 kernel's MFMA work per output voxel.  x (N,Cin,Di,Hi,Wi) fp32 -> out (N,Cout,D,H,W), D = (Di-1)/2 + 1, ...
// ---------------------------------------------------------------------------------------------------------------
constexpr int kS2TD = 4, kS2TH = 8;
constexpr int kS2HH = kS2TH + 1, kS2HW = kBfW + 1;
constexpr int kS2WSlots = 4 * 2 * 2 * 64;                    // up to 4 tap pairs per stage
__host__ __device__ constexpr int s2_ins(int TD, int TH, int TW) { return ((TD + 1) * (TH + 1) * (TW + 1) + 63) / 64 * 64; }
__host__ __device__ constexpr size_t s2_lds_bytes(int TD = kS2TD, int TH = kS2TH, int TW = kBfW, int OB = 1) {
    return (size_t)(2 * 2 * s2_ins(TD, TH, TW) + 2 * OB * kS2WSlots) * 16;
}
__host__ __device__ constexpr int s2_pairs(int pi) { return (1 << ((pi >> 2) + ((pi >> 1) & 1) + (pi & 1))) / 2 == 0 ? 1 : (1 << ((pi >> 2) + ((pi >> 1) & 1) + (pi & 1))) / 2; }
__host__ __device__ constexpr int s2_first_pair(int pi) { int n = 0; for (int q = 0; q < pi; ++q) n += s2_pairs(q); return n; }
// tap j of class pi as an offset in the class tile: per odd dimension bit 0 -> k = 0 (coarse index o - 1 = tile index o),
// bit 1 -> k = 2 (coarse index o = tile index o + 1); even dimensions contribute nothing
template <int HH = kS2HH, int HW = kS2HW>
__host__ __device__ constexpr int s2_tap_off(int pi, int j) {
    const int pd = pi >> 2, ph = (pi >> 1) & 1, pw = pi & 1, nt = 1 << (pd + ph + pw);
    int bits = j < nt ? j : nt - 1, jw = 0, jh = 0, jd = 0;
    if (pw) { jw = bits & 1; bits >>= 1; }
    if (ph) { jh = bits & 1; bits >>= 1; }
    if (pd) { jd = bits & 1; }
    return (jd * HH + jh) * HW + jw;
}

// Output tile TD x TH x TW: 4 x 8 x 16, or 3 x 16 x 8 (column group = 4 h-rows of 8) for the outputs 6 x 30 x 40 and 3 x 15 x 20 of
// the cost network, which the former pads 1.7x and 2.3x.
// PIN: the input is the parity-split SCL form of the tensor (what the producing layer's epilogue wrote: BfOut::pscl) -- a stage's
// class tile then arrives by LDS-DMA like the stride-1 kernel's halo tile, and the fetch / cut / ds_write of the fp32 form
// (every second element of a row: half of every fetched sector unused, ~100 vector instructions per thread and stage beside
// 12 - 48 MFMAs) is gone.
// CG: column groups (32 output voxels) per wave: 2 = 64 voxels x 64 channels per wave; 1 = twice the waves of half the voxels
// (the 3 x 16 x 8 tile: 12 waves, three on every SIMD, instead of 6).
// OB: groups of 64 output channels per block.  2: a block's staged class tile feeds twice the MFMAs (this layer reads 8 x the
// voxels it writes: its input DMAs, not its matrix work, set its time) and a wave's B fragment serves 4 row groups.
template <int TD, int TH, int TW, bool PIN = false, int CG = 2, int OB = 1>
__global__ __launch_bounds__(TD * TH * TW * 2 / CG) void conv3d_k3_s2_bf16x3_kernel(
    const float* __restrict__ xf, long long sN, long long sC, long long sD, long long sH, int Cin, const uint4* __restrict__ xp,
    int cDp, int cHp, int cWp, size_t cpiece, int Nviews, const uint4* __restrict__ wq,
    const float* __restrict__ scale, const float* __restrict__ shift, BfOut dst, int C8, int Cout, int Di, int Hi,
    int Wi, int D, int H, int W, int tiles_w, int relu, int nsplit, float* __restrict__ partial, size_t total, int xcd_map) {
    float* __restrict__ out = dst.f32;
    constexpr int RG = 32 / TW;                // h-rows of one column group
    static_assert(TH % RG == 0 && (TD * TH * TW) % (32 * CG) == 0, "whole column groups, CG per wave");
    constexpr int NW = TD * TH * TW / (32 * CG), NT = 64 * NW;
    constexpr int HH = TH + 1, HW = TW + 1, NVOX = (TD + 1) * HH * HW, INS = s2_ins(TD, TH, TW);
    constexpr int NV = (NVOX + NT - 1) / NT;   // 2 voxel slots per thread and stage
    extern __shared__ uint4 s_bf[];            // [2 stages][2 pieces][INS] input, then [2 stages][OB][kS2WSlots] weights
    uint4* s_in = s_bf;
    uint4* s_w = s_bf + 2 * 2 * INS;

    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    unsigned bx, by, bz;
    xcd_block_id(xcd_map, bx, by, bz);
    const int bw = bx % tiles_w, bh = bx / tiles_w;
    const int nob = Cout / (64 * OB);
    // bz = (view, block of 64 * OB output channels, split of the channel groups): see conv3d_k3_bf16x3_kernel
    const int split = bz % nsplit, zo = bz / nsplit;
    const int n = zo / nob, ob64 = (zo % nob) * OB;   // first group of 64 channels of the block
    const int c8_begin = (int)((long long)C8 * split / nsplit), c8_end = (int)((long long)C8 * (split + 1) / nsplit);
    const int w0 = bw * TW, h0 = bh * TH, d0 = by * TD;
    const int col = lane & 31, hh = lane >> 5;
    const float* xfn = PIN ? nullptr : xf + (size_t)n * sN;

    // ---- PIN: DMA plan of a class tile.  Class pi = (pd, ph, pw) keeps input voxel (2i - pd, ...)... stored at index i + 1 - ...:
    // the PSCL form stores class coordinate c (input voxel 2c + parity) at index c + 1 (index 0 = the zero border the odd classes
    // read for c = -1).  Tile slot (dz, hy, wx) of class pi is class coordinate (d0 + dz - pd, ...) = stored index d0 + dz + 1 - pd.
    constexpr int IN_DMA = INS / 64;
    constexpr int IN_PER_WAVE = (2 * IN_DMA + NW - 1) / NW;
    const size_t c8_stride = (size_t)cDp * cHp * cWp, cls_stride = (size_t)Nviews * C8 * c8_stride;
    const uint4* xn = PIN ? xp + ((size_t)n * C8) * c8_stride + ((size_t)d0 * cHp + h0) * cWp + w0 : nullptr;
    unsigned in_src[IN_PER_WAVE];
    if constexpr (PIN) {
#pragma unroll
        for (int k = 0; k < IN_PER_WAVE; ++k) {
            const int i = wave + k * NW;
            const int slot = (i % IN_DMA) * 64 + lane;
            const int sv = slot < NVOX ? slot : 0;
            const int dz = sv / (HH * HW), r = sv - dz * (HH * HW), hy = r / HW, wx = r - hy * HW;
            in_src[k] = (unsigned)(((size_t)dz * cHp + hy) * cWp + wx);
        }
    }
    auto dma_input = [&](int c8, int pi, int buf) {
        if constexpr (PIN) {
            const int pd = pi >> 2, ph = (pi >> 1) & 1, pw = pi & 1;
            const uint4* base = xn + (size_t)pi * cls_stride + (size_t)c8 * c8_stride + ((size_t)(1 - pd) * cHp + (1 - ph)) * cWp + (1 - pw);
#pragma unroll
            for (int k = 0; k < IN_PER_WAVE; ++k) {
                const int i = wave + k * NW;   // wave-uniform
                if (i < 2 * IN_DMA) {
                    const int piece = i / IN_DMA;
                    const uint4* src = base + (size_t)piece * cpiece + in_src[k];
                    uint4* dstl = s_in + (size_t)(buf * 2 + piece) * INS + (i % IN_DMA) * 64;
                    __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)src,
                                                     (__attribute__((address_space(3))) void*)dstl, 16, 0, 0);
                }
            }
        }
    };

    // the thread's voxel slots of a class tile: tile index (dz, hy, wx) <-> input voxel 2*(d0 + dz) - pd, ... (class pi)
    int vz[NV], vy[NV], vx[NV];
#pragma unroll
    for (int k = 0; k < NV; ++k) {
        const int slot = tid + k * NT;
        const int sv = slot < NVOX ? slot : 0;
        vz[k] = 2 * (d0 + sv / (HH * HW));
        vy[k] = 2 * (h0 + (sv % (HH * HW)) / HW);
        vx[k] = 2 * (w0 + sv % HW);
    }
    // (A second register set with two stages of flight time was tried twice: beside an LDS-DMA hipcc drains vmcnt(0) before it uses
    // an ordinary load's result, so the extra stage is never granted; with the DMA hidden in inline assembly, every load
    // unconditional (padding from a zero word) and a counted wait at the top of the step the compiler's waits are counted, and
    // the layer is slower: conv1 1.03 against 0.92 ms.)
    float f_reg[1][NV][8];
    auto fetch = [&](auto setc, int c8, int pi) {
        constexpr int set = decltype(setc)::value;
        const int pd = pi >> 2, ph = (pi >> 1) & 1, pw = pi & 1;
#pragma unroll
        for (int k = 0; k < NV; ++k) {
            const int d = vz[k] - pd, h = vy[k] - ph, w = vx[k] - pw;
            const bool ok = (tid + k * NT < NVOX) && d >= 0 && d < Di && h >= 0 && h < Hi && w >= 0 && w < Wi;
            const size_t off = ok ? (size_t)((long long)d * sD + (long long)h * sH + w) : 0;
#pragma unroll
            for (int j = 0; j < 8; ++j) {
                const int c = c8 * 8 + j;
                f_reg[set][k][j] = (ok && c < Cin) ? xfn[(size_t)c * sC + off] : 0.0f;
            }
        }
    };
    auto stage = [&](auto setc, int buf) {
        constexpr int set = decltype(setc)::value;
#pragma unroll
        for (int k = 0; k < NV; ++k) {
            const int slot = tid + k * NT;
            unsigned hi[4], mid[4];
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                unsigned short hb[2], mb[2];
#pragma unroll
                for (int e = 0; e < 2; ++e) {
                    const float f = f_reg[set][k][2 * j + e];
                    const __bf16 a = (__bf16)f;
                    const __bf16 b = (__bf16)(f - (float)a);
                    hb[e] = __builtin_bit_cast(unsigned short, a);
                    mb[e] = __builtin_bit_cast(unsigned short, b);
                }
                hi[j] = (unsigned)hb[0] | ((unsigned)hb[1] << 16);
                mid[j] = (unsigned)mb[0] | ((unsigned)mb[1] << 16);
            }
            if (slot < NVOX) {
                s_in[(size_t)(buf * 2 + 0) * INS + slot] = make_uint4(hi[0], hi[1], hi[2], hi[3]);
                s_in[(size_t)(buf * 2 + 1) * INS + slot] = make_uint4(mid[0], mid[1], mid[2], mid[3]);
            }
        }
    };
    // weights: wq[ob64][c8][14 pairs in class order][row group][piece][lane]; stage (c8, pi) = pairs first(pi) .. + pairs(pi)
    const uint4* wn = wq + (size_t)ob64 * C8 * (kBfPairs * 4 * 64);
    auto dma_weights = [&](int c8, int pi, int buf) {
        int first = 0, np = 1;
#pragma unroll
        for (int q = 0; q < 8; ++q)
            if (q == pi) { first = s2_first_pair(q); np = s2_pairs(q); }
        const uint4* src0 = wn + ((size_t)c8 * kBfPairs + first) * (4 * 64) + lane;
        for (int i = wave; i < OB * np * 4; i += NW) {
            const int o = i / (np * 4), r = i - o * (np * 4);   // group of 64 channels, 64-unit chunk of the class's pairs
            uint4* dst = s_w + (size_t)(buf * OB + o) * kS2WSlots + r * 64;
            __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)(src0 + (size_t)o * C8 * (kBfPairs * 4 * 64) + r * 64),
                                             (__attribute__((address_space(3))) void*)dst, 16, 0, 0);
        }
    };

    int vb[CG];
#pragma unroll
    for (int b = 0; b < CG; ++b) {
        const int g = CG * wave + b;
        const int dz = g / (TH / RG), hy = RG * (g % (TH / RG)) + col / TW;
        vb[b] = (dz * HH + hy) * HW + col % TW;
    }
    f32x16b acc[2 * OB][CG];
#pragma unroll
    for (int a = 0; a < 2 * OB; ++a)
#pragma unroll
        for (int b = 0; b < CG; ++b)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[a][b][r] = 0.0f;
    const bf16x8* s_in8 = reinterpret_cast<const bf16x8*>(s_in);
    const bf16x8* s_w8 = reinterpret_cast<const bf16x8*>(s_w);

    auto compute = [&](auto pc, int buf) {
        constexpr int pi = decltype(pc)::value;
        constexpr int np = s2_pairs(pi);
        const bf16x8* bin = s_in8 + (size_t)(buf * 2) * INS;
        const bf16x8* ain = s_w8 + (size_t)buf * OB * kS2WSlots + lane;
#pragma unroll
        for (int pl = 0; pl < np; ++pl) {
            const int toff = hh ? s2_tap_off<HH, HW>(pi, 2 * pl + 1) : s2_tap_off<HH, HW>(pi, 2 * pl);
            bf16x8 A[2 * OB][2], B[CG][2];
#pragma unroll
            for (int a = 0; a < 2 * OB; ++a)
#pragma unroll
                for (int q = 0; q < 2; ++q) A[a][q] = ain[(size_t)(a >> 1) * kS2WSlots + ((pl * 2 + (a & 1)) * 2 + q) * 64];
#pragma unroll
            for (int b = 0; b < CG; ++b)
#pragma unroll
                for (int q = 0; q < 2; ++q) B[b][q] = bin[(size_t)q * INS + vb[b] + toff];
#pragma unroll
            for (int a = 0; a < 2 * OB; ++a)
#pragma unroll
                for (int b = 0; b < CG; ++b) {
                    acc[a][b] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(A[a][1], B[b][0], acc[a][b], 0, 0, 0);
                    acc[a][b] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(A[a][0], B[b][1], acc[a][b], 0, 0, 0);
                    acc[a][b] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(A[a][0], B[b][0], acc[a][b], 0, 0, 0);
                }
        }
    };

    // pipeline over the flat stage index q = 8*c8 + pi; LDS buffers alternate with q (= with pi).  Stage q: the registers
    // hold stage q+1's values (fetched during stage q-1): cut them into LDS buffer (q+1)&1 (free since stage q-1 ended),
    // start the fetch of stage q+2 and the weight DMA of stage q+1, multiply stage q.
    const int nq = (c8_end - c8_begin) * 8;
    using S0 = std::integral_constant<int, 0>;
    if (nq > 0) {
        if constexpr (PIN) {
            dma_input(c8_begin, 0, 0);
            dma_weights(c8_begin, 0, 0);
        } else {
            fetch(S0{}, c8_begin, 0);
            stage(S0{}, 0);
            dma_weights(c8_begin, 0, 0);
            fetch(S0{}, c8_begin, 1);
        }
    }
    auto step = [&](auto pc, int c8) {
        constexpr int pi = decltype(pc)::value;
        constexpr int buf = pi & 1;
        const int q = (c8 - c8_begin) * 8 + pi;
        if constexpr (PIN) __builtin_amdgcn_s_waitcnt(0x0f70);   // vmcnt(0): this wave's DMAs of the stage have landed
        else __builtin_amdgcn_s_waitcnt(0x0070);   // vmcnt(0) and lgkmcnt(0): own fetches, DMAs and LDS writes are done
        __builtin_amdgcn_s_barrier();
        if (q + 1 < nq) {
            if constexpr (PIN) dma_input(pi == 7 ? c8 + 1 : c8, (pi + 1) & 7, buf ^ 1);
            else stage(S0{}, buf ^ 1);
            dma_weights(pi == 7 ? c8 + 1 : c8, (pi + 1) & 7, buf ^ 1);
        }
        if constexpr (!PIN)
            if (q + 2 < nq) fetch(S0{}, pi >= 6 ? c8 + 1 : c8, (pi + 2) & 7);
        compute(pc, buf);
    };
    for (int c8 = c8_begin; c8 < c8_end; ++c8) {
        step(std::integral_constant<int, 0>{}, c8);
        step(std::integral_constant<int, 1>{}, c8);
        step(std::integral_constant<int, 2>{}, c8);
        step(std::integral_constant<int, 3>{}, c8);
        step(std::integral_constant<int, 4>{}, c8);
        step(std::integral_constant<int, 5>{}, c8);
        step(std::integral_constant<int, 6>{}, c8);
        step(std::integral_constant<int, 7>{}, c8);
    }

    const size_t plane = (size_t)H * W, vol = (size_t)D * plane;
#pragma unroll
    for (int b = 0; b < CG; ++b) {
        const int g = CG * wave + b;
        const int d = d0 + g / (TH / RG), h = h0 + RG * (g % (TH / RG)) + col / TW, w = w0 + col % TW;
        if (d >= D || h >= H || w >= W) continue;
#pragma unroll
        for (int a = 0; a < 2 * OB; ++a) {
#pragma unroll
            for (int q = 0; q < 2; ++q) {
                float sc[8], sh[8], v[8];   // requested ahead of the stores (see conv3d_k3_bf16x3_kernel)
                const int o0 = ob64 * 64 + a * 32 + bf_row_channel(8 * q, hh);
                const size_t idx0 = ((size_t)n * Cout + o0) * vol + (size_t)d * plane + (size_t)h * W + w;
#pragma unroll
                for (int j = 0; j < 8; ++j) {
                    sc[j] = (nsplit == 1 && scale) ? scale[o0 + j] : 1.0f;
                    sh[j] = (nsplit == 1 && scale) ? shift[o0 + j] : 0.0f;
                }
#pragma unroll
                for (int j = 0; j < 8; ++j) {
                    v[j] = acc[a][b][8 * q + j];
                    if (nsplit > 1) {
                        partial[(size_t)split * total + idx0 + (size_t)j * vol] = v[j];
                        continue;
                    }
                    if (scale) v[j] = fmaf(v[j], sc[j], sh[j]);
                    if (relu) v[j] = fmaxf(v[j], 0.0f);
                    if (out) out[idx0 + (size_t)j * vol] = v[j];
                }
                if (nsplit == 1 && (dst.scl || dst.pscl)) bf_store_units(dst, v, n, Cout / 8, ob64 * 8 + a * 4 + 2 * q + hh, d, h, w);
            }
        }
    }
}

// ---------------------------------------------------------------------------------------------------------------
// Transposed convolution (mvsnet.py:92-100: conv9 256 -> 128, conv11 128 -> 64; ConvTranspose3d kernel 3, stride 2, padding 1,
// output_padding 1) + affine + ReLU, then the skip tensor is added (mvsnet.py:110-111).  Output 2i + p per dimension sees
// tap k = 1 on input i (p = 0), or k = 0 on input i + 1 and k = 2 on input i (p = 1): each of the 8 output parity classes is
// a stride-1 convolution over the INPUT grid with 1, 2, 4 or 8 taps.  One block = (coarse tile of 4 x 8 x 16 input voxels,
// output parity (PD, PH), BOTH w parities, 64 output channels): 8 waves x (2 row groups x 2 column groups x 2 w parities) =
// 8 accumulators; the two neighbouring outputs of a lane leave as one float2.  A stage = 8 input channels: the halo tile
// (5 x 9 x 17 voxels, SCL form of the coarse input) and the 2 - 6 tap pairs of the two classes arrive by LDS-DMA, two
// stages ahead (three LDS buffers, counted vmcnt): the stages are as short as 24 MFMAs per wave.
// ---------------------------------------------------------------------------------------------------------------
__host__ __device__ constexpr int ct_pairs(int pd, int ph) { return s2_pairs(pd * 4 + ph * 2) + s2_pairs(pd * 4 + ph * 2 + 1); }
constexpr int kCtWSlots = 6 * 2 * 2 * 64;   // up to 6 tap pairs per stage
// slots of one piece of one input stage: the halo tile, rounded up until the 2 * INS / 64 DMAs of a stage divide evenly over the waves
__host__ __device__ constexpr int ct_ins(int TD, int TH, int TW, int CG = 2) {
    int ins = ((TD + 1) * (TH + 1) * (TW + 1) + 63) / 64 * 64;
    while ((2 * ins / 64) % (TD * TH * TW / (32 * CG))) ins += 64;
    return ins;
}
__host__ __device__ constexpr size_t ct_lds_bytes(int TD = kS2TD, int TH = kS2TH, int TW = kBfW, int CG = 2) {
    return (size_t)(3 * 2 * ct_ins(TD, TH, TW, CG) + 3 * kCtWSlots) * 16;
}
// tap j of output class pi as an offset in the halo tile (origin = the tile's first input voxel): per odd dimension bit 0 ->
// k = 0 reads input i + 1, bit 1 -> k = 2 reads input i
template <int HH = kS2HH, int HW = kS2HW>
__host__ __device__ constexpr int ct_tap_off(int pi, int j) {
    const int pd = pi >> 2, ph = (pi >> 1) & 1, pw = pi & 1, nt = 1 << (pd + ph + pw);
    int bits = j < nt ? j : nt - 1, jw = 0, jh = 0, jd = 0;
    if (pw) { jw = bits & 1; bits >>= 1; }
    if (ph) { jh = bits & 1; bits >>= 1; }
    if (pd) { jd = bits & 1; }
    const int od = (pd && !jd) ? 1 : 0, oh = (ph && !jh) ? 1 : 0, ow = (pw && !jw) ? 1 : 0;
    return (od * HH + oh) * HW + ow;
}

// CG: column groups (32 coarse voxels each) per wave.  2: a wave = 64 voxels x 64 channels x 2 w parities = 8 accumulators, the
// 3 x 16 x 8 tile = 6 waves, which sit 2/2/1/1 on the four SIMDs of the one block a CU holds; 1: 4 accumulators, 12 waves, three
// on every SIMD (each A fragment then feeds half the MFMAs: 6 LDS reads per 6 instead of 8 per 12).
template <int PD, int PH, int TD, int TH, int TW, int CG>
__device__ __forceinline__ void convT3d_k3_s2_bf16x3_body(
    const uint4* __restrict__ xs, const uint4* __restrict__ wq, const float* __restrict__ scale, const float* __restrict__ shift,
    const float* __restrict__ residual, const BfOut& dst, int C8, int Cout, int Di, int Hi, int Wi, int Dp, int Hp, int Wp,
    size_t piece_stride, int tiles_w, int relu, int tile_xy) {
    float* __restrict__ out = dst.f32;
    constexpr int RG = 32 / TW;                           // h-rows of one column group
    static_assert(TH % RG == 0 && (TD * TH * TW) % (32 * CG) == 0, "whole column groups, CG per wave");
    constexpr int NW = TD * TH * TW / (32 * CG);
    constexpr int HH = TH + 1, HW = TW + 1, NVOX = (TD + 1) * HH * HW, INS = ct_ins(TD, TH, TW, CG);
    constexpr int IN_DMA = INS / 64;                      // 12 wave-instructions per piece
    constexpr int IN_PER_WAVE = 2 * IN_DMA / NW;          // 3 (8 waves), 4 (6 waves) or 2 (12 waves)
    constexpr int PI0 = PD * 4 + PH * 2, NP0 = s2_pairs(PI0), NP1 = s2_pairs(PI0 + 1), NP = NP0 + NP1;
    constexpr int W_PER_WAVE = (NP * 4 + NW - 1) / NW;    // weight DMAs per wave and stage (the tail repeats earlier pieces)
    constexpr int DMA_PER_STAGE = IN_PER_WAVE + W_PER_WAVE;
    static_assert(2 * IN_DMA % NW == 0, "input DMAs must divide evenly over the waves");
    extern __shared__ uint4 s_bf[];   // [3 stages][2 pieces][INS] input, then [3 stages][kCtWSlots] weights
    uint4* s_in = s_bf;
    uint4* s_w = s_bf + 3 * 2 * INS;

    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int bw = tile_xy % tiles_w, bh = tile_xy / tiles_w;
    const int nob = Cout / 64;
    const int n = blockIdx.z / nob, ob64 = blockIdx.z % nob;
    const int w0 = bw * TW, h0 = bh * TH, d0 = blockIdx.y * TD;
    const int col = lane & 31, hh = lane >> 5;

    // input DMA plan: halo voxel (dz, hy, wx) = input (d0 + dz, h0 + hy, w0 + wx) = padded (+1, +1, +1)
    const size_t c8_stride = (size_t)Dp * Hp * Wp;
    const uint4* xn = xs + ((size_t)n * C8) * c8_stride + ((size_t)(d0 + 1) * Hp + (h0 + 1)) * Wp + (w0 + 1);
    unsigned in_src[IN_PER_WAVE];
#pragma unroll
    for (int k = 0; k < IN_PER_WAVE; ++k) {
        const int i = wave + k * NW;
        const int slot = (i % IN_DMA) * 64 + lane;
        const int sv = slot < NVOX ? slot : 0;
        const int dz = sv / (HH * HW), r = sv - dz * (HH * HW), hy = r / HW, wx = r - hy * HW;
        in_src[k] = (unsigned)(((size_t)dz * Hp + hy) * Wp + wx);
    }
    const uint4* wn = wq + (size_t)ob64 * C8 * (kBfPairs * 4 * 64) + (size_t)s2_first_pair(PI0) * (4 * 64) + lane;
    auto dma_stage = [&](int c8, int buf) {
#pragma unroll
        for (int k = 0; k < IN_PER_WAVE; ++k) {
            const int i = wave + k * NW;
            const int piece = i / IN_DMA;
            const uint4* src = xn + (size_t)piece * piece_stride + (size_t)c8 * c8_stride + in_src[k];
            uint4* dst = s_in + (size_t)(buf * 2 + piece) * INS + (i % IN_DMA) * 64;
            __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)src,
                                             (__attribute__((address_space(3))) void*)dst, 16, 0, 0);
        }
#pragma unroll
        for (int k = 0; k < W_PER_WAVE; ++k) {
            const int i = (wave + k * NW) % (NP * 4);   // the tail wraps: every wave issues the same number of DMAs
            uint4* dst = s_w + (size_t)buf * kCtWSlots + i * 64;
            __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)(wn + (size_t)c8 * (kBfPairs * 4 * 64) + i * 64),
                                             (__attribute__((address_space(3))) void*)dst, 16, 0, 0);
        }
    };

    int vb[CG];
#pragma unroll
    for (int b = 0; b < CG; ++b) {
        const int g = CG * wave + b;
        const int dz = g / (TH / RG), hy = RG * (g % (TH / RG)) + col / TW;
        vb[b] = (dz * HH + hy) * HW + col % TW;
    }
    f32x16b acc[2][2][CG];   // [w parity][row group][column group]
#pragma unroll
    for (int pw = 0; pw < 2; ++pw)
#pragma unroll
        for (int a = 0; a < 2; ++a)
#pragma unroll
            for (int b = 0; b < CG; ++b)
#pragma unroll
                for (int r = 0; r < 16; ++r) acc[pw][a][b][r] = 0.0f;
    const bf16x8* s_in8 = reinterpret_cast<const bf16x8*>(s_in);
    const bf16x8* s_w8 = reinterpret_cast<const bf16x8*>(s_w);

    auto compute = [&](int buf) {
        const bf16x8* bin = s_in8 + (size_t)(buf * 2) * INS;
        const bf16x8* ain = s_w8 + (size_t)buf * kCtWSlots + lane;
#pragma unroll
        for (int pl = 0; pl < NP; ++pl) {
            const int pw = pl < NP0 ? 0 : 1;
            const int pi = PI0 + pw, pj = pw ? pl - NP0 : pl;
            const int toff = hh ? ct_tap_off<HH, HW>(pi, 2 * pj + 1) : ct_tap_off<HH, HW>(pi, 2 * pj);
            bf16x8 A[2][2], B[CG][2];
#pragma unroll
            for (int a = 0; a < 2; ++a)
#pragma unroll
                for (int q = 0; q < 2; ++q) A[a][q] = ain[((pl * 2 + a) * 2 + q) * 64];
#pragma unroll
            for (int b = 0; b < CG; ++b)
#pragma unroll
                for (int q = 0; q < 2; ++q) B[b][q] = bin[(size_t)q * INS + vb[b] + toff];
#pragma unroll
            for (int a = 0; a < 2; ++a)
#pragma unroll
                for (int b = 0; b < CG; ++b) {
                    acc[pw][a][b] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(A[a][1], B[b][0], acc[pw][a][b], 0, 0, 0);
                    acc[pw][a][b] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(A[a][0], B[b][1], acc[pw][a][b], 0, 0, 0);
                    acc[pw][a][b] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(A[a][0], B[b][0], acc[pw][a][b], 0, 0, 0);
                }
        }
    };

    // three LDS stages: at stage c8 the DMAs of c8+2 are issued (into the buffer stage c8-1 used), and the wait at the top
    // leaves one stage's DMAs (those of c8+1) in flight
    dma_stage(0, 0);
    if (C8 > 1) dma_stage(1, 1);
    int buf = 0;
    for (int c8 = 0; c8 < C8; ++c8) {
        if (c8 + 1 < C8) __builtin_amdgcn_s_waitcnt(0x0f70 | (DMA_PER_STAGE & 15) | ((DMA_PER_STAGE >> 4) << 14));
        else __builtin_amdgcn_s_waitcnt(0x0f70);
        __builtin_amdgcn_s_barrier();
        if (c8 + 2 < C8) dma_stage(c8 + 2, buf == 0 ? 2 : buf - 1);
        compute(buf);
        buf = buf == 2 ? 0 : buf + 1;
    }

    // epilogue: lane = coarse voxel; the two w parities are neighbouring outputs -> one float2
    const int Do = 2 * Di, Ho = 2 * Hi, Wo = 2 * Wi;
    const size_t oplane = (size_t)Ho * Wo, ovol = (size_t)Do * oplane;
#pragma unroll
    for (int b = 0; b < CG; ++b) {
        const int g = CG * wave + b;
        const int di = d0 + g / (TH / RG), hi = h0 + RG * (g % (TH / RG)) + col / TW, wi = w0 + col % TW;
        if (di >= Di || hi >= Hi || wi >= Wi) continue;
        const size_t pos = (size_t)(2 * di + PD) * oplane + (size_t)(2 * hi + PH) * Wo + 2 * wi;
#pragma unroll
        for (int a = 0; a < 2; ++a) {
            // the 16 skip values of this (row group, column group) are requested together, ahead of the stores: loads and
            // stores share one in-order counter, so a load issued behind a store waits for the store's round trip as well
            // -- one value at a time, the epilogue took half of the layer's time
            if (CG == 2 && !(dst.scl || dst.pscl)) {   // (12-wave form: 168 registers per lane do not hold 16 at a time)
                // fp32 only: the 16 skip values of this (row group, column group) are requested together, ahead of the stores:
                // loads and stores share one in-order counter, so a load issued behind a store waits for the store's round trip
                // as well -- one value at a time, the epilogue took half of the layer's time (conv11: 1.2 GB of skip and output)
                float2 rv[16];
                float sc[16], sh[16];   // (the affine too: a load between two stores waits for the first store)
#pragma unroll
                for (int r = 0; r < 16; ++r) {
                    const int o = ob64 * 64 + a * 32 + bf_row_channel(r, hh);
                    rv[r] = residual ? *reinterpret_cast<const float2*>(residual + ((size_t)n * Cout + o) * ovol + pos) : make_float2(0.f, 0.f);
                    sc[r] = scale ? scale[o] : 1.0f;
                    sh[r] = scale ? shift[o] : 0.0f;
                }
#pragma unroll
                for (int r = 0; r < 16; ++r) {
                    const int o = ob64 * 64 + a * 32 + bf_row_channel(r, hh);
                    float2 v = make_float2(acc[0][a][b][r], acc[1][a][b][r]);
                    if (scale) {
                        v.x = fmaf(v.x, sc[r], sh[r]);
                        v.y = fmaf(v.y, sc[r], sh[r]);
                    }
                    if (relu) {
                        v.x = fmaxf(v.x, 0.0f);
                        v.y = fmaxf(v.y, 0.0f);
                    }
                    if (residual) {
                        v.x = rv[r].x + v.x;
                        v.y = rv[r].y + v.y;
                    }
                    *reinterpret_cast<float2*>(out + ((size_t)n * Cout + o) * ovol + pos) = v;
                }
                continue;
            }
#pragma unroll
            for (int q = 0; q < 2; ++q) {   // eight consecutive channels at a time: one unit per piece and w parity
                float2 rv[8];
                float sc[8], sh[8];
                const int o0 = ob64 * 64 + a * 32 + bf_row_channel(8 * q, hh);
                const size_t idx0 = ((size_t)n * Cout + o0) * ovol + pos;
#pragma unroll
                for (int j = 0; j < 8; ++j) {
                    rv[j] = residual ? *reinterpret_cast<const float2*>(residual + idx0 + (size_t)j * ovol) : make_float2(0.f, 0.f);
                    sc[j] = scale ? scale[o0 + j] : 1.0f;
                    sh[j] = scale ? shift[o0 + j] : 0.0f;
                }
                float v0[8], v1[8];
#pragma unroll
                for (int j = 0; j < 8; ++j) {
                    float2 v = make_float2(acc[0][a][b][8 * q + j], acc[1][a][b][8 * q + j]);
                    if (scale) {
                        v.x = fmaf(v.x, sc[j], sh[j]);
                        v.y = fmaf(v.y, sc[j], sh[j]);
                    }
                    if (relu) {
                        v.x = fmaxf(v.x, 0.0f);
                        v.y = fmaxf(v.y, 0.0f);
                    }
                    if (residual) {
                        v.x = rv[j].x + v.x;
                        v.y = rv[j].y + v.y;
                    }
                    v0[j] = v.x;
                    v1[j] = v.y;
                    if (out) *reinterpret_cast<float2*>(out + idx0 + (size_t)j * ovol) = v;
                }
                const int c8o = ob64 * 8 + a * 4 + 2 * q + hh;
                bf_store_units(dst, v0, n, Cout / 8, c8o, 2 * di + PD, 2 * hi + PH, 2 * wi);
                bf_store_units(dst, v1, n, Cout / 8, c8o, 2 * di + PD, 2 * hi + PH, 2 * wi + 1);
            }
        }
    }
}

// ---------------------------------------------------------------------------------------------------------------
// The transposed layer with ALL EIGHT output parity classes in one block (option "convT_fused"): block = (coarse tile of
// 3 x 16 x 8 input voxels, 32 output channels), 12 waves of one column group; a wave holds 8 accumulators = the 8 outputs
// 2i + (pd, ph, pw) of its 32 coarse voxels x 32 channels.  Against one block per (PD, PH): the halo tile of a stage is staged
// once for all classes (Cout / 32 times per tile instead of 4 x Cout / 64), and a stage is the 14 tap pairs of all classes =
// 42 MFMAs per wave behind one barrier instead of 6 - 18.  Every accumulator still sums the same (channel group, tap pair)
// sequence as in the per-class kernel: bit-identical.  LDS: 3 x (2 pieces x 768 slots input + 28 x 64 slots weights) = 159.8 KB.
// ---------------------------------------------------------------------------------------------------------------
constexpr int kCtfWSlots = kBfPairs * 2 * 64;   // 14 pairs x 2 pieces x one row group of 32 channels
__host__ __device__ constexpr size_t ctf_lds_bytes(int TD, int TH, int TW) {
    return (size_t)(3 * 2 * ct_ins(TD, TH, TW, 1) + 3 * kCtfWSlots) * 16;
}
// STATS (training: a BatchNorm on batch statistics follows, costreg_bn.hip): the raw fp32 output plus per-channel partial sums of
// (value - pivot_c) and of its square, one double2 per (channel, block) -- stats[c * parts + block], parts = views x tiles -- as the
// stride-1 kernel's statistics epilogue leaves them.  The epilogue has no registers to spare (168 + spills), so a lane keeps its 16
// channels' running sums in LDS slots of its own (the stage buffers are free by then): read - add - write per group of 8 values, no
// atomics, then four threads per channel add the 384 contributors in double, in a fixed order.
template <int TD, int TH, int TW, bool STATS = false>
__global__ __launch_bounds__(TD * TH * TW * 2) void convT3d_k3_s2_bf16x3_fused_kernel(
    const uint4* __restrict__ xs, const uint4* __restrict__ wq, const float* __restrict__ scale, const float* __restrict__ shift,
    const float* __restrict__ residual, BfOut dst, int C8, int Cout, int Di, int Hi, int Wi, int Dp, int Hp, int Wp,
    size_t piece_stride, int tiles_w, int relu, int xcd_map, double2* __restrict__ stats, const float* __restrict__ stats_pivot) {
    float* __restrict__ out = dst.f32;
    constexpr int RG = 32 / TW;
    static_assert(TH % RG == 0 && (TD * TH * TW) % 32 == 0, "whole column groups");
    constexpr int NW = TD * TH * TW / 32;
    constexpr int HH = TH + 1, HW = TW + 1, NVOX = (TD + 1) * HH * HW, INS = ct_ins(TD, TH, TW, 1);
    constexpr int IN_DMA = INS / 64, IN_PER_WAVE = 2 * IN_DMA / NW;
    constexpr int W_CHUNKS = kBfPairs * 2;                               // (pair, piece) chunks of 64 units
    constexpr int W_PER_WAVE = (W_CHUNKS + NW - 1) / NW;                  // the tail repeats earlier chunks
    constexpr int DMA_PER_STAGE = IN_PER_WAVE + W_PER_WAVE;
    static_assert(2 * IN_DMA % NW == 0, "input DMAs must divide evenly over the waves");
    extern __shared__ uint4 s_bf[];   // [3 stages][2 pieces][INS] input, then [3 stages][kCtfWSlots] weights
    uint4* s_in = s_bf;
    uint4* s_w = s_bf + 3 * 2 * INS;

    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    unsigned bx, by, bz;
    xcd_block_id(xcd_map, bx, by, bz);
    const int bw = bx % tiles_w, bh = bx / tiles_w;
    const int nob = Cout / 32;
    const int n = bz / nob, ob32 = bz % nob, ob64 = ob32 >> 1, rg = ob32 & 1;
    const int w0 = bw * TW, h0 = bh * TH, d0 = by * TD;
    const int col = lane & 31, hh = lane >> 5;

    const size_t c8_stride = (size_t)Dp * Hp * Wp;
    const uint4* xn = xs + ((size_t)n * C8) * c8_stride + ((size_t)(d0 + 1) * Hp + (h0 + 1)) * Wp + (w0 + 1);
    unsigned in_src[IN_PER_WAVE];
#pragma unroll
    for (int k = 0; k < IN_PER_WAVE; ++k) {
        const int i = wave + k * NW;
        const int slot = (i % IN_DMA) * 64 + lane;
        const int sv = slot < NVOX ? slot : 0;
        const int dz = sv / (HH * HW), r = sv - dz * (HH * HW), hy = r / HW, wx = r - hy * HW;
        in_src[k] = (unsigned)(((size_t)dz * Hp + hy) * Wp + wx);
    }
    // weights: wq[ob64][c8][pair 14][row group 2][piece 2][lane 64]; this block's row group only, as [pair][piece][lane]
    const uint4* wn = wq + (size_t)ob64 * C8 * (kBfPairs * 4 * 64) + (size_t)rg * (2 * 64) + lane;
    auto dma_stage = [&](int c8, int buf) {
#pragma unroll
        for (int k = 0; k < IN_PER_WAVE; ++k) {
            const int i = wave + k * NW;
            const int piece = i / IN_DMA;
            const uint4* src = xn + (size_t)piece * piece_stride + (size_t)c8 * c8_stride + in_src[k];
            uint4* dstl = s_in + (size_t)(buf * 2 + piece) * INS + (i % IN_DMA) * 64;
            __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)src,
                                             (__attribute__((address_space(3))) void*)dstl, 16, 0, 0);
        }
#pragma unroll
        for (int k = 0; k < W_PER_WAVE; ++k) {
            const int i = (wave + k * NW) % W_CHUNKS;   // chunk = (pair i >> 1, piece i & 1)
            uint4* dstl = s_w + (size_t)buf * kCtfWSlots + i * 64;
            __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)(wn + (size_t)c8 * (kBfPairs * 4 * 64) + (size_t)((i >> 1) * 4 + (i & 1)) * 64),
                                             (__attribute__((address_space(3))) void*)dstl, 16, 0, 0);
        }
    };

    const int g = wave;
    const int vb = ((g / (TH / RG)) * HH + RG * (g % (TH / RG)) + col / TW) * HW + col % TW;
    f32x16b acc[8];   // output parity class pi = 4 * pd + 2 * ph + pw
#pragma unroll
    for (int pi = 0; pi < 8; ++pi)
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[pi][r] = 0.0f;
    const bf16x8* s_in8 = reinterpret_cast<const bf16x8*>(s_in);
    const bf16x8* s_w8 = reinterpret_cast<const bf16x8*>(s_w);

    auto compute = [&](int buf) {
        const bf16x8* bin = s_in8 + (size_t)(buf * 2) * INS + vb;
        const bf16x8* ain = s_w8 + (size_t)buf * kCtfWSlots + lane;
#pragma unroll
        for (int pi = 0; pi < 8; ++pi) {
#pragma unroll
            for (int pj = 0; pj < s2_pairs(pi); ++pj) {
                const int pl = s2_first_pair(pi) + pj;
                const int toff = hh ? ct_tap_off<HH, HW>(pi, 2 * pj + 1) : ct_tap_off<HH, HW>(pi, 2 * pj);
                const bf16x8 a0 = ain[(pl * 2 + 0) * 64], a1 = ain[(pl * 2 + 1) * 64];
                const bf16x8 b0 = bin[toff], b1 = bin[(size_t)INS + toff];
                acc[pi] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a1, b0, acc[pi], 0, 0, 0);
                acc[pi] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a0, b1, acc[pi], 0, 0, 0);
                acc[pi] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a0, b0, acc[pi], 0, 0, 0);
            }
        }
    };

    dma_stage(0, 0);
    if (C8 > 1) dma_stage(1, 1);
    int buf = 0;
    for (int c8 = 0; c8 < C8; ++c8) {
        if (c8 + 1 < C8) __builtin_amdgcn_s_waitcnt(0x0f70 | (DMA_PER_STAGE & 15) | ((DMA_PER_STAGE >> 4) << 14));
        else __builtin_amdgcn_s_waitcnt(0x0f70);
        __builtin_amdgcn_s_barrier();
        if (c8 + 2 < C8) dma_stage(c8 + 2, buf == 0 ? 2 : buf - 1);
        compute(buf);
        buf = buf == 2 ? 0 : buf + 1;
    }

    // epilogue: lane = coarse voxel; per (pd, ph) the two w parities are neighbouring outputs -> one float2; eight channels at a time
    const int Do = 2 * Di, Ho = 2 * Hi, Wo = 2 * Wi;
    const size_t oplane = (size_t)Ho * Wo, ovol = (size_t)Do * oplane;
    const int di = d0 + g / (TH / RG), hi = h0 + RG * (g % (TH / RG)) + col / TW, wi = w0 + col % TW;
    const bool inside = di < Di && hi < Hi && wi < Wi;
    constexpr int NT = 64 * NW;
    float2* const slot = reinterpret_cast<float2*>(s_bf) + tid;            // [register r 0..15][thread]: the lane's (sum, sum of squares) of channel r
    float* const s_pivot = reinterpret_cast<float*>(reinterpret_cast<float2*>(s_bf) + 16 * NT);   // the block's 32 pivots
    if constexpr (STATS) {
        static_assert((size_t)16 * NT * sizeof(float2) + 32 * sizeof(float) <= ctf_lds_bytes(TD, TH, TW), "the statistics' LDS image");
        __syncthreads();   // every wave is past its last read of the stage buffers
        if (tid < 32) s_pivot[tid] = stats_pivot ? stats_pivot[ob64 * 64 + rg * 32 + tid] : 0.0f;
        __syncthreads();
    } else {
        if (!inside) return;
    }
#pragma unroll
    for (int cls = 0; cls < 4; ++cls) {
        const int pd = cls >> 1, ph = cls & 1;
        const size_t pos = (size_t)(2 * di + pd) * oplane + (size_t)(2 * hi + ph) * Wo + 2 * wi;
#pragma unroll
        for (int q = 0; q < 2; ++q) {
            float2 rv[8];
            float sc[8], sh[8];
            const int o0 = ob64 * 64 + rg * 32 + bf_row_channel(8 * q, hh);
            const size_t idx0 = ((size_t)n * Cout + o0) * ovol + pos;
            if constexpr (STATS) {   // raw outputs; sums of both w parities of the lane's 8 channels into its slots
#pragma unroll
                for (int j = 0; j < 8; ++j) {
                    const float2 v = make_float2(acc[2 * cls][8 * q + j], acc[2 * cls + 1][8 * q + j]);
                    const float pv = s_pivot[bf_row_channel(8 * q + j, hh)];
                    const float d0v = inside ? v.x - pv : 0.0f, d1v = inside ? v.y - pv : 0.0f;
                    float2 e = cls == 0 ? make_float2(0.0f, 0.0f) : slot[(8 * q + j) * NT];
                    e.x += d0v + d1v;
                    e.y = fmaf(d0v, d0v, fmaf(d1v, d1v, e.y));
                    slot[(8 * q + j) * NT] = e;
                    if (inside) *reinterpret_cast<float2*>(out + idx0 + (size_t)j * ovol) = v;
                }
                continue;
            }
#pragma unroll
            for (int j = 0; j < 8; ++j) {
                rv[j] = residual ? *reinterpret_cast<const float2*>(residual + idx0 + (size_t)j * ovol) : make_float2(0.f, 0.f);
                sc[j] = scale ? scale[o0 + j] : 1.0f;
                sh[j] = scale ? shift[o0 + j] : 0.0f;
            }
            float v0[8], v1[8];
#pragma unroll
            for (int j = 0; j < 8; ++j) {
                float2 v = make_float2(acc[2 * cls][8 * q + j], acc[2 * cls + 1][8 * q + j]);
                if (scale) {
                    v.x = fmaf(v.x, sc[j], sh[j]);
                    v.y = fmaf(v.y, sc[j], sh[j]);
                }
                if (relu) {
                    v.x = fmaxf(v.x, 0.0f);
                    v.y = fmaxf(v.y, 0.0f);
                }
                if (residual) {
                    v.x = rv[j].x + v.x;
                    v.y = rv[j].y + v.y;
                }
                v0[j] = v.x;
                v1[j] = v.y;
                if (out) *reinterpret_cast<float2*>(out + idx0 + (size_t)j * ovol) = v;
            }
            if (dst.scl || dst.pscl) {
                const int c8o = ob64 * 8 + rg * 4 + 2 * q + hh;
                bf_store_units(dst, v0, n, Cout / 8, c8o, 2 * di + pd, 2 * hi + ph, 2 * wi);
                bf_store_units(dst, v1, n, Cout / 8, c8o, 2 * di + pd, 2 * hi + ph, 2 * wi + 1);
            }
        }
    }
    if constexpr (STATS) {
        __syncthreads();
        if (tid < 128) {
            // channel c of the block = register r = 8 (c >> 4) + (c & 7) of the lanes with hh = (c >> 3) & 1: 12 waves x 32 lanes
            const int c = tid >> 2, part = tid & 3;
            const int r = 8 * (c >> 4) + (c & 7), chh = (c >> 3) & 1;
            const float2* col0 = reinterpret_cast<const float2*>(s_bf) + (size_t)r * NT + chh * 32;
            double a = 0.0, b2 = 0.0;
            constexpr int NCON = NW * 32;
            for (int i = part * (NCON / 4); i < (part + 1) * (NCON / 4); ++i) {
                const float2 e = col0[(i >> 5) * 64 + (i & 31)];
                a += (double)e.x;
                b2 += (double)e.y;
            }
            a += __shfl_xor(a, 1, 64); b2 += __shfl_xor(b2, 1, 64);
            a += __shfl_xor(a, 2, 64); b2 += __shfl_xor(b2, 2, 64);
            if (part == 0) {
                const size_t tiles = (size_t)gridDim.x * gridDim.y;
                const size_t bidx = ((size_t)n * gridDim.y + by) * gridDim.x + bx;
                stats[(size_t)(ob64 * 64 + rg * 32 + c) * ((size_t)(gridDim.z / nob) * tiles) + bidx] = make_double2(a, b2);
            }
        }
    }
}

#include "convt_persist.h"   // the same layer as a persistent kernel: the stores of item i behind the multiplications of item i + 1

// ONE launch for the four (PD, PH) classes: class = blockIdx.x & 3, so the four blocks that read the same input tile are
// dispatched together (the tile's second to fourth reads hit L2) and the grid has one tail instead of four.
template <int TD, int TH, int TW, int CG>
__global__ __launch_bounds__(TD * TH * TW * 2 / CG) void convT3d_k3_s2_bf16x3_kernel(
    const uint4* __restrict__ xs, const uint4* __restrict__ wq, const float* __restrict__ scale, const float* __restrict__ shift,
    const float* __restrict__ residual, BfOut out, int C8, int Cout, int Di, int Hi, int Wi, int Dp, int Hp, int Wp,
    size_t piece_stride, int tiles_w, int relu) {
    const int cls = blockIdx.x & 3, tile_xy = blockIdx.x >> 2;
    if (cls == 0) convT3d_k3_s2_bf16x3_body<1, 1, TD, TH, TW, CG>(xs, wq, scale, shift, residual, out, C8, Cout, Di, Hi, Wi, Dp, Hp, Wp, piece_stride, tiles_w, relu, tile_xy);
    else if (cls == 1) convT3d_k3_s2_bf16x3_body<1, 0, TD, TH, TW, CG>(xs, wq, scale, shift, residual, out, C8, Cout, Di, Hi, Wi, Dp, Hp, Wp, piece_stride, tiles_w, relu, tile_xy);
    else if (cls == 2) convT3d_k3_s2_bf16x3_body<0, 1, TD, TH, TW, CG>(xs, wq, scale, shift, residual, out, C8, Cout, Di, Hi, Wi, Dp, Hp, Wp, piece_stride, tiles_w, relu, tile_xy);
    else convT3d_k3_s2_bf16x3_body<0, 0, TD, TH, TW, CG>(xs, wq, scale, shift, residual, out, C8, Cout, Di, Hi, Wi, Dp, Hp, Wp, piece_stride, tiles_w, relu, tile_xy);
}

}  // namespace mvsdet

using namespace mvsdet;

#include "costreg_mx.h"   // conv0 on one fp16 + two block-scaled FP6 products per fp32-equivalent product (kernels)

namespace {
struct BfPlan {
    int td, th, tiles_d, tiles_h, tiles_w, Dp, Hp, Wp, tw = kBfW;
};
int round_up(int v, int m) { return (v + m - 1) / m * m; }
// Padded extents of the SCL form of a (D,H,W) volume: the one-voxel zero border plus whatever the LARGEST tile overhang of
// any kernel that reads the form needs -- tiles 4 x {8,12} x 16, {3,6} x 16 x 8 (stride 1) and 4 x 8 x 16, 3 x 16 x 8 (+ 1
// halo voxel, transposed) -- so that a buffer does not depend on which tile shape a consumer picks.
void scl_dims(int D, int H, int W, int& Dp, int& Hp, int& Wp) {
    Dp = std::max(round_up(D, 3), std::max(round_up(D, 4), round_up(D, 6))) + 2;
    Hp = std::max(round_up(H, 8), std::max(round_up(H, 12), round_up(H, 16))) + 2;
    Wp = round_up(W, 16) + 2;
}
// One parity class of the PSCL form of a (D,H,W) volume: class coordinates 0 .. ceil(D/2)-1 stored at index + 1, read by the
// stride-2 kernel's 4 x 8 x 16 or 3 x 16 x 8 output tiles (+ 1 slot)
void pscl_dims(int D, int H, int W, int& cDp, int& cHp, int& cWp) {
    const int Dc = (D + 1) / 2, Hc = (H + 1) / 2, Wc = (W + 1) / 2;
    cDp = std::max(round_up(Dc, 3), round_up(Dc, 4)) + 2;
    cHp = std::max(round_up(Hc, 8), round_up(Hc, 16)) + 2;
    cWp = round_up(Wc, 16) + 2;
}
// tile = 4 x TH x 16 with TH = 8 or 12, whichever pads H less (60 rows: 12): the tiling the number of input-channel splits
// is derived from (for every input form and tile shape, so that all of them add up the same partial sums)
BfPlan bf_plan(int D, int H, int W) {
    BfPlan p;
    p.td = 4;
    const int pad8 = (H + 7) / 8 * 8, pad12 = (H + 11) / 12 * 12;
    p.th = pad12 < pad8 ? 12 : 8;
    p.tiles_d = (D + p.td - 1) / p.td;
    p.tiles_h = (H + p.th - 1) / p.th;
    p.tiles_w = (W + kBfW - 1) / kBfW;
    scl_dims(D, H, W, p.Dp, p.Hp, p.Wp);
    return p;
}
// The tile that pads (D, H, W) least.  At equal padding the earlier candidate wins: 12- and 8-wave tiles before the 6-wave
// one (its waves sit 2/2/1/1 on the four SIMDs).  8 x 8 x 8: fp32-input form only (the neck's levels).
BfPlan bf_plan_tile(int D, int H, int W, bool f32in) {
    static const int cand[][3] = {{4, 12, 16}, {4, 8, 16}, {6, 16, 8}, {3, 16, 8}, {8, 8, 8}};
    BfPlan best = bf_plan(D, H, W);
    long long best_vol = -1;
    for (const auto& c : cand) {
        const int td = c[0], th = c[1], tw = c[2];
        if (td == 8 && !f32in) continue;
        const int nd = (D + td - 1) / td, nh = (H + th - 1) / th, nw = (W + tw - 1) / tw;
        const long long v = (long long)nd * td * nh * th * nw * tw;
        if (best_vol < 0 || v < best_vol) {
            best_vol = v;
            best.td = td; best.th = th; best.tw = tw; best.tiles_d = nd; best.tiles_h = nh; best.tiles_w = nw;
        }
    }
    return best;
}
BfOut make_out(float* f32, void* scl, void* pscl, int N, int Cout, int D, int H, int W) {
    BfOut o;
    o.f32 = f32;
    o.scl = static_cast<uint4*>(scl);
    o.pscl = static_cast<uint4*>(pscl);
    o.N = N;
    scl_dims(D, H, W, o.Dp, o.Hp, o.Wp);
    pscl_dims(D, H, W, o.cDp, o.cHp, o.cWp);
    o.piece = (size_t)N * (Cout / 8) * o.Dp * o.Hp * o.Wp;
    o.cpiece = (size_t)8 * N * (Cout / 8) * o.cDp * o.cHp * o.cWp;
    return o;
}
}  // namespace

// Geometry of the split channel-last form of an (N,C,D,H,W) activation: padded extents and total bytes (both pieces).
extern "C" size_t mvsdet_scl_bytes(int N, int C, int D, int H, int W, int* Dp, int* Hp, int* Wp) {
    if (N <= 0 || C <= 0 || D <= 0 || H <= 0 || W <= 0) return 0;
    int dp, hp, wp;
    scl_dims(D, H, W, dp, hp, wp);
    if (Dp) *Dp = dp;
    if (Hp) *Hp = hp;
    if (Wp) *Wp = wp;
    return (size_t)2 * N * ((C + 7) / 8) * dp * hp * wp * 16;
}

// Geometry of the parity-split SCL form: [piece 2][class 8][n][c8][cDp][cHp][cWp][8] bf16; class = 4*(d&1) + 2*(h&1) + (w&1),
// voxel (d,h,w) at index (d/2 + 1, h/2 + 1, w/2 + 1) of its class, zero elsewhere.
extern "C" size_t mvsdet_pscl_bytes(int N, int C, int D, int H, int W, int* cDp, int* cHp, int* cWp) {
    if (N <= 0 || C <= 0 || D <= 0 || H <= 0 || W <= 0) return 0;
    int dp, hp, wp;
    pscl_dims(D, H, W, dp, hp, wp);
    if (cDp) *cDp = dp;
    if (cHp) *cHp = hp;
    if (cWp) *cWp = wp;
    return (size_t)2 * 8 * N * ((C + 7) / 8) * dp * hp * wp * 16;
}

// x (N,C,D,H,W) fp32, element strides xstr = {n, c, d, h} (w stride 1; NULL = contiguous) -> xs (SCL, mvsdet_scl_bytes).  The border of xs must be zero: `zero_border` != 0 clears the whole
// buffer first (one memset; a caller that reuses the buffer for the same shape clears it once and passes 0).
extern "C" int mvsdet_scl_pack_f32(const float* x, const int64_t* xstr, void* xs, int N, int C, int D, int H, int W,
                                   int zero_border, mvsdet_stream_t stream) {
    MVS_REQUIRE(x && xs, "scl_pack: NULL pointer");
    MVS_REQUIRE(N > 0 && C > 0 && D > 0 && H > 0 && W > 0, "scl_pack: bad shape N=%d C=%d D=%d H=%d W=%d", N, C, D, H, W);
    MVS_REQUIRE(((uintptr_t)xs & 15u) == 0, "scl_pack: xs must be 16-byte aligned");
    const BfPlan p = bf_plan(D, H, W);
    const int C8 = (C + 7) / 8;
    MVS_REQUIRE(N <= 65535 && C8 <= 65535, "scl_pack: N or C too large");
    const size_t piece = (size_t)N * C8 * p.Dp * p.Hp * p.Wp;
    hipStream_t st = (hipStream_t)stream;
    if (zero_border == 1 && hipMemsetAsync(xs, 0, 2 * piece * 16, st) != hipSuccess) {
        set_error("scl_pack: hipMemsetAsync failed");
        return MVSDET_ERR_HIP;
    }
    const size_t vol = (size_t)D * H * W;
    dim3 grid((unsigned)((vol + kThreads - 1) / kThreads), (unsigned)C8, (unsigned)N);
    const long long sN = xstr ? xstr[0] : (long long)C * vol, sC = xstr ? xstr[1] : (long long)vol;
    const long long sD = xstr ? xstr[2] : (long long)H * W, sH = xstr ? xstr[3] : (long long)W;
    MVS_REQUIRE(sN >= 0 && sC >= 0 && sD >= 0 && sH >= W, "scl_pack: bad strides");
    if (zero_border == 2) {   // one pass over the padded volume: border voxels get their zeros from the kernel itself
        const size_t volp = (size_t)p.Dp * p.Hp * p.Wp;
        dim3 gridp((unsigned)((volp + kThreads - 1) / kThreads), (unsigned)C8, (unsigned)N);
        hipLaunchKernelGGL(scl_pack_padded_kernel, gridp, dim3(kThreads), 0, st, x, sN, sC, sD, sH, static_cast<uint4*>(xs), C, C8, D,
                           H, W, p.Dp, p.Hp, p.Wp, piece);
        MVS_LAUNCH_CHECK("scl_pack");
        return MVSDET_OK;
    }
    hipLaunchKernelGGL(scl_pack_kernel, grid, dim3(kThreads), 0, st, x, sN, sC, sD, sH, static_cast<uint4*>(xs), C, C8, D, H, W,
                       p.Dp, p.Hp, p.Wp, piece);
    MVS_LAUNCH_CHECK("scl_pack");
    return MVSDET_OK;
}

// weight (Cout = 64*m, Cin, 3,3,3) fp32 -> weight_split, (Cout/64) * ceil(Cin/8) * 14 * 2 * 2 * 64 * 16 bytes
extern "C" size_t mvsdet_split_conv_weight_bytes(int Cout, int Cin) {
    if (Cout <= 0 || Cout % 64 || Cin <= 0) return 0;
    return (size_t)(Cout / 64) * ((Cin + 7) / 8) * kBfPairs * 2 * 2 * 64 * 16;
}
// order: 0 = stride-1 convolution (pair p = taps 2p, 2p+1), 1 = stride-2 convolution (taps grouped by the parity class of the
// input voxel they read: conv3d_k3_s2_bf16x3_kernel), 2 = transposed convolution (ConvTranspose3d weight (Cin,Cout,3,3,3);
// taps grouped by the parity class of the OUTPUT voxel: convT3d_k3_s2_bf16x3_kernel)
static TapTable tap_table(int order) {
    TapTable tt;
    if (order == 0) {
        for (int i = 0; i < 2 * kBfPairs; ++i) tt.t[i] = (signed char)(i < 27 ? i : -1);
    } else {
        // per dimension an odd class has two taps (k = 0, k = 2), an even class one (k = 1); classes in the order
        // pi = 4*pd + 2*ph + pw, inside a class the taps in the order of (jd, jh, jw) over the odd dimensions
        int n = 0;
        for (int pi = 0; pi < 8; ++pi) {
            const int pd = pi >> 2, ph = (pi >> 1) & 1, pw = pi & 1, nt = 1 << (pd + ph + pw);
            for (int j = 0; j < nt; ++j) {
                int bits = j, jw = 0, jh = 0, jd = 0;
                if (pw) { jw = bits & 1; bits >>= 1; }
                if (ph) { jh = bits & 1; bits >>= 1; }
                if (pd) { jd = bits & 1; }
                const int kd = pd ? 2 * jd : 1, kh = ph ? 2 * jh : 1, kw = pw ? 2 * jw : 1;
                tt.t[n++] = (signed char)((kd * 3 + kh) * 3 + kw);
            }
            if (nt == 1) tt.t[n++] = -1;   // the lone tap of the all-even class fills half a pair
        }
    }
    return tt;
}

extern "C" int mvsdet_split_conv_weight_ordered(const float* weight, void* weight_split, int Cout, int Cin, int order,
                                                mvsdet_stream_t stream) {
    MVS_REQUIRE(weight && weight_split, "split_conv_weight: NULL pointer");
    MVS_REQUIRE(Cout > 0 && Cout % 64 == 0 && Cin > 0, "split_conv_weight: Cout=%d must be a positive multiple of 64, Cin=%d > 0", Cout, Cin);
    MVS_REQUIRE(order >= 0 && order <= 3, "split_conv_weight: order %d not in {0,1,2,3}", order);
    if (order == 0 && options().conv_mfma16) order = 3;   // the stride-1 kernels then run their 16x16x32 form: its layout
    MVS_REQUIRE(((uintptr_t)weight_split & 15u) == 0, "split_conv_weight: output must be 16-byte aligned");
    const size_t units = mvsdet_split_conv_weight_bytes(Cout, Cin) / 16;
    const long long so = order == 2 ? 27 : (long long)Cin * 27, sc = order == 2 ? (long long)Cout * 27 : 27;
    hipLaunchKernelGGL(split_conv_weight_kernel, dim3((unsigned)((units + kThreads - 1) / kThreads)), dim3(kThreads), 0,
                       (hipStream_t)stream, weight, static_cast<uint4*>(weight_split), Cin, (Cin + 7) / 8, units,
                       tap_table(order == 3 ? 0 : order), so, sc, order == 3 ? 1 : 0);
    MVS_LAUNCH_CHECK("split_conv_weight");
    return MVSDET_OK;
}

// Up to 8 weight tensors in one launch: weights[i] (Cout[i] = 64*m, Cin[i], 3,3,3) (order 2: (Cin,Cout,3,3,3)) -> splits[i]
// (mvsdet_split_conv_weight_bytes(Cout[i], Cin[i]) bytes each), tap order orders[i] as for mvsdet_split_conv_weight_ordered.
// The pointer arrays are HOST arrays of device pointers.
extern "C" int mvsdet_split_conv_weights_batched(const float* const* weights, void* const* splits, const int* Cout, const int* Cin,
                                                 const int* orders, int count, mvsdet_stream_t stream) {
    MVS_REQUIRE(weights && splits && Cout && Cin && orders, "split_conv_weights_batched: NULL pointer");
    MVS_REQUIRE(count > 0 && count <= kSplitBatchMax, "split_conv_weights_batched: 1..%d tensors per call", kSplitBatchMax);
    SplitBatch b;
    b.n = count;
    b.first[0] = 0;
    for (int k = 0; k < 3; ++k) b.taps[k] = tap_table(k);
    for (int i = 0; i < kSplitBatchMax; ++i) {
        const bool live = i < count;
        if (live) {
            MVS_REQUIRE(weights[i] && splits[i], "split_conv_weights_batched: NULL tensor %d", i);
            MVS_REQUIRE(Cout[i] > 0 && Cout[i] % 64 == 0 && Cin[i] > 0 && orders[i] >= 0 && orders[i] <= 3,
                        "split_conv_weights_batched: tensor %d: Cout=%d (multiple of 64), Cin=%d, order=%d", i, Cout[i], Cin[i], orders[i]);
            MVS_REQUIRE(((uintptr_t)splits[i] & 15u) == 0, "split_conv_weights_batched: outputs must be 16-byte aligned");
        }
        b.w[i] = live ? weights[i] : nullptr;
        b.out[i] = live ? static_cast<uint4*>(splits[i]) : nullptr;
        b.Cin[i] = live ? Cin[i] : 1;
        b.Cout[i] = live ? Cout[i] : 64;
        b.order[i] = live ? ((orders[i] == 0 && options().conv_mfma16) ? 3 : orders[i]) : 0;
        b.first[i + 1] = b.first[i] + (live ? mvsdet_split_conv_weight_bytes(Cout[i], Cin[i]) / 16 : 0);
    }
    const size_t units = b.first[count];
    hipLaunchKernelGGL(split_conv_weight_batched_kernel, dim3((unsigned)((units + kThreads - 1) / kThreads)), dim3(kThreads), 0,
                       (hipStream_t)stream, b);
    MVS_LAUNCH_CHECK("split_conv_weights_batched");
    return MVSDET_OK;
}

extern "C" int mvsdet_split_conv_weight(const float* weight, void* weight_split, int Cout, int Cin, mvsdet_stream_t stream) {
    return mvsdet_split_conv_weight_ordered(weight, weight_split, Cout, Cin, 0, stream);
}

namespace mvsdet {
void launch_splitk_epilogue(const float* partial, int nsplit, size_t total, const float* scale, const float* shift,
                            const float* residual, float* out, int Cout, size_t vol, int relu, hipStream_t st);   // costreg_conv0.hip
}

// splits of the channel groups for a grid of `blocks` blocks (one 8-12 wave block per CU): up to ~3 rounds of the chip, at
// least 2 channel groups per split (options "conv_split_blocks", "conv_split_min_groups")
static int bf_nsplit(long long blocks, int C8) {
    const int forced = options().conv_nsplit;   // tuning knob: the size query and the launch both come through here
    if (forced > 0) return (int)std::max(1, std::min(forced, C8 / 4));
    if (blocks >= 192) return 1;
    // three rounds of the chip (option "conv_split_blocks", default 768): the 3-D neck of one scene 2.54 -> 2.32 ms against two
    // rounds (512); four (1024) 2.43 -- shorter chains of stages per block against more partial sums to write and add
    const int target = std::max(1, options().conv_split_blocks);
    const int ming = std::max(1, options().conv_split_min_groups);
    return (int)std::max(1LL, std::min<long long>((target + blocks - 1) / blocks, C8 / ming));
}

extern "C" size_t mvsdet_conv3d_k3_bf16x3_workspace_bytes(int N, int Cin, int Cout, int D, int H, int W) {
    if (N <= 0 || Cin <= 0 || Cout <= 0 || Cout % 64 || D <= 0 || H <= 0 || W <= 0) return 0;
    const BfPlan p = bf_plan(D, H, W);
    const int ns = bf_nsplit((long long)p.tiles_w * p.tiles_h * p.tiles_d * N * (Cout / 64), (Cin + 7) / 8);
    return ns > 1 ? (size_t)ns * N * Cout * D * H * W * sizeof(float) : 0;
}

static int launch_bf16x3(const char* name, const void* xs, const float* xf, const int64_t* xstr, const void* weight_split,
                         const float* scale, const float* shift, const float* residual, float* out, void* out_scl, void* out_pscl,
                         int N, int Cin, int Cout, int D, int H, int W, int relu, mvsdet_stream_t stream, void* workspace = nullptr,
                         size_t workspace_bytes = 0, void* stats = nullptr, size_t stats_bytes = 0, const float* stats_pivot = nullptr) {
    MVS_REQUIRE((xs || xf) && weight_split && (out || out_scl || out_pscl), "%s: NULL pointer", name);
    MVS_REQUIRE((scale == nullptr) == (shift == nullptr), "%s: scale and shift come together", name);
    MVS_REQUIRE(N > 0 && Cin > 0 && D > 0 && H > 0 && W > 0, "%s: bad shape N=%d Cin=%d D=%d H=%d W=%d", name, N, Cin, D, H, W);
    MVS_REQUIRE(Cout > 0 && Cout % 64 == 0, "%s: Cout=%d must be a multiple of 64", name, Cout);
    MVS_REQUIRE((((uintptr_t)xs | (uintptr_t)weight_split | (uintptr_t)out_scl | (uintptr_t)out_pscl) & 15u) == 0,
                "%s: SCL buffers and weights must be 16-byte aligned", name);
    const BfPlan p = bf_plan_tile(D, H, W, xf != nullptr);
    const int C8 = (Cin + 7) / 8;
    MVS_REQUIRE((size_t)p.Dp * p.Hp * p.Wp < ((size_t)1 << 31), "%s: one padded channel-group volume exceeds 2^31 voxels", name);
    const size_t vol = (size_t)D * H * W, total = (size_t)N * Cout * vol;
    // without (enough) workspace the convolution runs unsplit.  The number of splits follows from the 4 x TH x 16 tiling for
    // EVERY input form and tile shape, so that all of them add up the same partial sums.  The split form writes fp32 only.
    const BfPlan ps = bf_plan(D, H, W);
    int nsplit = bf_nsplit((long long)ps.tiles_w * ps.tiles_h * ps.tiles_d * N * (Cout / 64), C8);
    if (!workspace || workspace_bytes < (size_t)nsplit * total * sizeof(float) || out_scl || out_pscl || stats) nsplit = 1;
    MVS_REQUIRE(nsplit == 1 || out, "%s: the split form needs the fp32 output", name);
    if (stats) {
        MVS_REQUIRE(options().conv_mfma16 != 0 && options().conv_subpairs != 2, "%s: the statistics epilogue exists in the 16x16x32 form only", name);
        MVS_REQUIRE(!scale && !residual && !relu && out && !out_scl && !out_pscl, "%s: statistics are those of the raw fp32 output", name);
        MVS_REQUIRE(stats_bytes >= (size_t)Cout * N * p.tiles_w * p.tiles_h * p.tiles_d * sizeof(double2) && ((uintptr_t)stats & 15u) == 0,
                    "%s: the statistics buffer holds Cout x mvsdet_conv3d_k3_bf16x3_stats_parts double2", name);
    }
    MVS_REQUIRE((long long)N * (Cout / 64) * nsplit <= 65535 && p.tiles_d <= 65535, "%s: N*Cout/64 or D too large", name);
    const long long sN = xstr ? xstr[0] : (long long)Cin * vol, sC = xstr ? xstr[1] : (long long)vol;
    const long long sD = xstr ? xstr[2] : (long long)H * W, sH = xstr ? xstr[3] : (long long)W;
    if (xf) {
        MVS_REQUIRE(sN >= 0 && sC >= 0 && sD >= 0 && sH >= W, "%s: bad strides", name);
        MVS_REQUIRE((long long)(D - 1) * sD + (long long)(H - 1) * sH + W < (1LL << 31), "%s: one channel volume spans more than 2^31 elements", name);
    }
    const size_t piece = (size_t)N * C8 * p.Dp * p.Hp * p.Wp;
    const BfOut dst = make_out(out, out_scl, out_pscl, N, Cout, D, H, W);
    dim3 grid((unsigned)(p.tiles_w * p.tiles_h), (unsigned)p.tiles_d, (unsigned)(N * (Cout / 64) * nsplit));
    const int xcd_map = options().conv_xcd != 0 && ((long long)grid.x * grid.y * grid.z) % 8 == 0 && (long long)grid.x * grid.y * grid.z >= 64;
    hipStream_t st = (hipStream_t)stream;
#define MVS_BF_CASE(TD_, TH_, F32_, TW_, SUBP_, M16_, ...)                                                                  \
    {                                                                                                                       \
        constexpr int CGN_ = (0, ##__VA_ARGS__) ? (0, ##__VA_ARGS__) : 4;                                                   \
        auto* k = conv3d_k3_bf16x3_kernel<TD_, TH_, F32_, TW_, SUBP_, M16_, CGN_>;                                          \
        const size_t lds = bf_lds_bytes(TD_, TH_, TW_, SUBP_);                                                              \
        if (hipFuncSetAttribute(reinterpret_cast<const void*>(k), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds) !=  \
            hipSuccess) {                                                                                                   \
            set_error("%s: hipFuncSetAttribute(MaxDynamicSharedMemorySize) failed", name);                                  \
            return MVSDET_ERR_HIP;                                                                                          \
        }                                                                                                                   \
        hipLaunchKernelGGL(k, grid, dim3(TD_ * TH_ * TW_ * 4 / CGN_), lds, st, static_cast<const uint4*>(xs), xf, sN, sC, sD, sH, \
                           Cin, static_cast<const uint4*>(weight_split), scale, shift, residual, dst, C8, Cout, D, H, W,    \
                           p.Dp, p.Hp, p.Wp, piece, p.tiles_w, relu, nsplit, static_cast<float*>(workspace), total, xcd_map,         \
                           static_cast<double2*>(stats), stats_pivot);                                                      \
    }
#define MVS_BF_TILE(TD_, TH_, F32_, TW_) { if (m16) MVS_BF_CASE(TD_, TH_, F32_, TW_, 6, true) else MVS_BF_CASE(TD_, TH_, F32_, TW_, 5, false) }
    // option "conv_subpairs" = 2: weight sub-stages of 2 tap pairs (instead of 5) bring the 3 x 16 x 8 tile's LDS to 77 KiB, two
    // 6-wave blocks per CU.  Measured at the cost network's 6 x 30 x 40 layer: 1.00 against 0.97 ms -- seven barriers per
    // channel group cost what the second block wins; kept as a knob, off by default.
    // option "conv_mfma16": the 16x16x32 form of the kernel (the weights must have been split while the option was set).
    const bool sub2 = options().conv_subpairs == 2, m16 = options().conv_mfma16 != 0;
    if (xf) {
        // the 3 x 16 x 8 tile on 12 waves of 32 voxels (16x16x32 form; option "conv_cgn" = 4: 6 waves of 64)
        if (p.tw == 8 && p.td == 3) { if (sub2 && !m16) MVS_BF_CASE(3, 16, true, 8, 2, false) else if (m16 && options().conv_cgn != 4) MVS_BF_CASE(3, 16, true, 8, 6, true, 2) else MVS_BF_TILE(3, 16, true, 8) }
        else if (p.tw == 8 && p.td == 6) MVS_BF_TILE(6, 16, true, 8)
        else if (p.tw == 8) MVS_BF_TILE(8, 8, true, 8)
        // (the cost network's first layer: two weight sub-stages of 4 + 3 k-steps instead of 3 + 3 + 1 -- one barrier less per
        // channel group, all 160 KiB of LDS; "conv_subpairs" = 6 keeps three)
        else if (p.th == 12) { if (m16 && options().conv_subpairs != 6) MVS_BF_CASE(4, 12, true, kBfW, 8, true) else MVS_BF_TILE(4, 12, true, kBfW) }
        else MVS_BF_TILE(4, 8, true, kBfW)
    } else {
        if (p.tw == 8 && p.td == 3) { if (m16 && options().conv_cgn != 4) MVS_BF_CASE(3, 16, false, 8, 6, true, 2) else MVS_BF_TILE(3, 16, false, 8) }
        else if (p.tw == 8) MVS_BF_TILE(6, 16, false, 8)
        else if (p.th == 12) MVS_BF_TILE(4, 12, false, kBfW)
        else MVS_BF_TILE(4, 8, false, kBfW)
    }
#undef MVS_BF_TILE
#undef MVS_BF_CASE
    MVS_LAUNCH_CHECK(name);
    if (nsplit > 1) {
        launch_splitk_epilogue(static_cast<const float*>(workspace), nsplit, total, scale, shift, residual, out, Cout, vol, relu, st);
        MVS_LAUNCH_CHECK(name);
    }
    return MVSDET_OK;
}

// Conv3d(Cin -> Cout = 64*m, kernel 3, stride 1, padding 1, no bias) [+ per-channel affine] [+ residual] [+ ReLU] on the
// bf16 matrix cores, three-term split (file header).  Input: xs = SCL form (mvsdet_scl_pack_f32 or a producing layer's
// out_scl), or x = the fp32 tensor itself (x_strides = element strides of n, c, d, h; w stride 1; NULL = contiguous), cut
// inside the kernel -- same results bit for bit.  weight_split: mvsdet_split_conv_weight.  Outputs, any combination (NULL =
// not wanted): out_f32 (N,Cout,D,H,W); out_scl = its SCL form; out_pscl = its parity-split SCL form (their buffers'
// borders must be zero: the kernel writes interior voxels only).  workspace (mvsdet_conv3d_k3_bf16x3_workspace_bytes; NULL
// or too small: unsplit) is only used when out_f32 is the sole output.
extern "C" int mvsdet_conv3d_k3_bf16x3_io(const void* xs, const float* x, const int64_t* x_strides, const void* weight_split,
                                          const float* scale, const float* shift, const float* residual, float* out_f32,
                                          void* out_scl, void* out_pscl, void* workspace, size_t workspace_bytes, int N, int Cin,
                                          int Cout, int D, int H, int W, int relu, mvsdet_stream_t stream) {
    MVS_REQUIRE((xs == nullptr) != (x == nullptr), "conv3d_k3_bf16x3_io: exactly one of xs (SCL) and x (fp32)");
    return launch_bf16x3("conv3d_k3_bf16x3_io", xs, x, x_strides, weight_split, scale, shift, residual, out_f32, out_scl, out_pscl, N,
                         Cin, Cout, D, H, W, relu, stream, workspace, workspace_bytes);
}

extern "C" int mvsdet_conv3d_k3_bf16x3(const void* xs, const void* weight_split, const float* scale, const float* shift,
                                       const float* residual, float* out, int N, int Cin, int Cout, int D, int H, int W, int relu,
                                       mvsdet_stream_t stream) {
    return launch_bf16x3("conv3d_k3_bf16x3", xs, nullptr, nullptr, weight_split, scale, shift, residual, out, nullptr, nullptr, N, Cin,
                         Cout, D, H, W, relu, stream);
}

extern "C" int mvsdet_conv3d_k3_bf16x3_f32in(const float* x, const int64_t* x_strides, const void* weight_split,
                                             const float* scale, const float* shift, const float* residual, float* out, int N,
                                             int Cin, int Cout, int D, int H, int W, int relu, mvsdet_stream_t stream) {
    return launch_bf16x3("conv3d_k3_bf16x3_f32in", nullptr, x, x_strides, weight_split, scale, shift, residual, out, nullptr, nullptr,
                         N, Cin, Cout, D, H, W, relu, stream);
}

// The convolution in front of a training-mode BatchNorm (module.py:26-37): raw fp32 output + per-channel partial sums of the
// outputs and of their squares, one double2 per (channel, block of the grid), stats[c * parts + i] -- what
// mvsdet_bn3d_relu_train_fwd_parts_f32 finishes, so that the BatchNorm reads the tensor once instead of twice.
// parts = mvsdet_conv3d_k3_bf16x3_stats_parts(N, D, H, W, fp32 input form).  x / xs as for mvsdet_conv3d_k3_bf16x3_io.
// pivot (Cout floats or NULL = zeros): the sums are those of (value - pivot_c) -- any value near the channel's mean (the
// BatchNorm's running mean) keeps the fp32 lane sums from cancelling; the finishing call gets the same vector.
extern "C" size_t mvsdet_conv3d_k3_bf16x3_stats_parts(int N, int D, int H, int W, int f32_input) {
    if (N <= 0 || D <= 0 || H <= 0 || W <= 0) return 0;
    const BfPlan p = bf_plan_tile(D, H, W, f32_input != 0);
    return (size_t)N * p.tiles_w * p.tiles_h * p.tiles_d;
}

extern "C" int mvsdet_conv3d_k3_bf16x3_stats(const void* xs, const float* x, const int64_t* x_strides, const void* weight_split,
                                             float* out_f32, void* stats, size_t stats_bytes, const float* pivot, int N, int Cin,
                                             int Cout, int D, int H, int W, mvsdet_stream_t stream) {
    MVS_REQUIRE((xs == nullptr) != (x == nullptr), "conv3d_k3_bf16x3_stats: exactly one of xs (SCL) and x (fp32)");
    MVS_REQUIRE(stats != nullptr, "conv3d_k3_bf16x3_stats: NULL statistics buffer");
    return launch_bf16x3("conv3d_k3_bf16x3_stats", xs, x, x_strides, weight_split, nullptr, nullptr, nullptr, out_f32, nullptr, nullptr, N,
                         Cin, Cout, D, H, W, 0, stream, nullptr, 0, stats, stats_bytes, pivot);
}

// The two forms with a workspace (mvsdet_conv3d_k3_bf16x3_workspace_bytes; NULL or too small: unsplit) for small volumes.
extern "C" int mvsdet_conv3d_k3_bf16x3_ws(const void* xs, const void* weight_split, const float* scale, const float* shift,
                                          const float* residual, float* out, void* workspace, size_t workspace_bytes, int N, int Cin,
                                          int Cout, int D, int H, int W, int relu, mvsdet_stream_t stream) {
    return launch_bf16x3("conv3d_k3_bf16x3", xs, nullptr, nullptr, weight_split, scale, shift, residual, out, nullptr, nullptr, N, Cin,
                         Cout, D, H, W, relu, stream, workspace, workspace_bytes);
}

extern "C" int mvsdet_conv3d_k3_bf16x3_f32in_ws(const float* x, const int64_t* x_strides, const void* weight_split,
                                                const float* scale, const float* shift, const float* residual, float* out,
                                                void* workspace, size_t workspace_bytes, int N, int Cin, int Cout, int D, int H,
                                                int W, int relu, mvsdet_stream_t stream) {
    return launch_bf16x3("conv3d_k3_bf16x3_f32in", nullptr, x, x_strides, weight_split, scale, shift, residual, out, nullptr, nullptr,
                         N, Cin, Cout, D, H, W, relu, stream, workspace, workspace_bytes);
}

// Conv3d(Cin -> Cout = 64*m, kernel 3, stride 2, padding 1, no bias) [+ affine] [+ ReLU] of mvsnet.py:77,79 on the bf16 matrix
// cores, three-term split; x (N,Cin,D,H,W) fp32 (x_strides as above) -> out (N,Cout,(D-1)/2+1,(H-1)/2+1,(W-1)/2+1);
// weight_split from mvsdet_split_conv_weight_ordered(order = 1).
extern "C" size_t mvsdet_conv3d_k3_s2_bf16x3_workspace_bytes(int N, int Cin, int Cout, int Di, int Hi, int Wi) {
    if (N <= 0 || Cin <= 0 || Cout <= 0 || Cout % 64 || Di <= 0 || Hi <= 0 || Wi <= 0) return 0;
    const int D = (Di - 1) / 2 + 1, H = (Hi - 1) / 2 + 1, W = (Wi - 1) / 2 + 1;
    const long long tiles = (long long)((W + kBfW - 1) / kBfW) * ((H + kS2TH - 1) / kS2TH) * ((D + kS2TD - 1) / kS2TD);
    const int ns = bf_nsplit(tiles * N * (Cout / 64), (Cin + 7) / 8);
    return ns > 1 ? (size_t)ns * N * Cout * D * H * W * sizeof(float) : 0;
}

static int launch_s2_bf16x3(const float* x, const int64_t* x_strides, const void* x_pscl, const void* weight_split, const float* scale,
                            const float* shift, float* out, void* out_scl, void* workspace, size_t workspace_bytes, int N, int Cin,
                            int Cout, int Di, int Hi, int Wi, int relu, mvsdet_stream_t stream) {
    const char* name = x_pscl ? "conv3d_k3_s2_bf16x3 (PSCL input)" : "conv3d_k3_s2_bf16x3_f32in";
    MVS_REQUIRE((x == nullptr) != (x_pscl == nullptr), "conv3d_k3_s2_bf16x3: exactly one of x (fp32) and x_pscl");
    MVS_REQUIRE(weight_split && (out || out_scl), "%s: NULL pointer", name);
    MVS_REQUIRE((scale == nullptr) == (shift == nullptr), "%s: scale and shift come together", name);
    MVS_REQUIRE(N > 0 && Cin > 0 && Di > 0 && Hi > 0 && Wi > 0, "%s: bad shape N=%d Cin=%d D=%d H=%d W=%d", name, N, Cin, Di, Hi, Wi);
    MVS_REQUIRE(Cout > 0 && Cout % 64 == 0, "%s: Cout=%d must be a multiple of 64", name, Cout);
    MVS_REQUIRE((((uintptr_t)weight_split | (uintptr_t)x_pscl | (uintptr_t)out_scl) & 15u) == 0, "%s: SCL buffers and weights must be 16-byte aligned", name);
    const int D = (Di - 1) / 2 + 1, H = (Hi - 1) / 2 + 1, W = (Wi - 1) / 2 + 1;
    const size_t ivol = (size_t)Di * Hi * Wi;
    const long long sN = x_strides ? x_strides[0] : (long long)Cin * ivol, sC = x_strides ? x_strides[1] : (long long)ivol;
    const long long sD = x_strides ? x_strides[2] : (long long)Hi * Wi, sH = x_strides ? x_strides[3] : (long long)Wi;
    if (x) {
        MVS_REQUIRE(sN >= 0 && sC >= 0 && sD >= 0 && sH >= Wi, "%s: bad strides", name);
        MVS_REQUIRE((long long)(Di - 1) * sD + (long long)(Hi - 1) * sH + Wi < (1LL << 31), "%s: one channel volume spans more than 2^31 elements", name);
    }
    // output tile 4 x 8 x 16 or 3 x 16 x 8, whichever pads (D, H, W) less; the number of splits follows from the former
    const long long tiles416 = (long long)((W + kBfW - 1) / kBfW) * ((H + kS2TH - 1) / kS2TH) * ((D + kS2TD - 1) / kS2TD);
    const long long pad416 = tiles416 * (kS2TD * kS2TH * kBfW);
    const long long tiles38 = (long long)((W + 7) / 8) * ((H + 15) / 16) * ((D + 2) / 3), pad38 = tiles38 * (3 * 16 * 8);
    const bool t38 = pad38 < pad416;
    const int ttd = t38 ? 3 : kS2TD, tth = t38 ? 16 : kS2TH, ttw = t38 ? 8 : kBfW;
    const int tiles_w = (W + ttw - 1) / ttw, tiles_h = (H + tth - 1) / tth, tiles_d = (D + ttd - 1) / ttd;
    const size_t total = (size_t)N * Cout * D * H * W;
    int nsplit = bf_nsplit(tiles416 * N * (Cout / 64), (Cin + 7) / 8);
    if (!workspace || workspace_bytes < (size_t)nsplit * total * sizeof(float) || out_scl) nsplit = 1;   // unsplit without (enough) workspace
    MVS_REQUIRE(nsplit == 1 || out, "%s: the split form needs the fp32 output", name);
    MVS_REQUIRE((long long)N * (Cout / 64) * nsplit <= 65535 && tiles_d <= 65535, "%s: N*Cout/64 or D too large", name);
    int cDp = 0, cHp = 0, cWp = 0;
    pscl_dims(Di, Hi, Wi, cDp, cHp, cWp);
    MVS_REQUIRE(!x_pscl || (tiles_d * ttd + 2 <= cDp && tiles_h * tth + 2 <= cHp && tiles_w * ttw + 2 <= cWp),
                "%s: the PSCL padding does not cover the tiles", name);
    const int C8 = (Cin + 7) / 8;
    const size_t cpiece = (size_t)8 * N * C8 * cDp * cHp * cWp;
    const BfOut dst = make_out(out, out_scl, nullptr, N, Cout, D, H, W);
    dim3 grid((unsigned)(tiles_w * tiles_h), (unsigned)tiles_d, (unsigned)(N * (Cout / 64) * nsplit));
#define MVS_S2_CASE(TD_, TH_, TW_, PIN_, CG_, ...)                                                                           \
    {                                                                                                                        \
        constexpr int OB_ = (0, ##__VA_ARGS__) ? (0, ##__VA_ARGS__) : 1;                                                     \
        auto* k = conv3d_k3_s2_bf16x3_kernel<TD_, TH_, TW_, PIN_, CG_, OB_>;                                                 \
        const size_t lds = s2_lds_bytes(TD_, TH_, TW_, OB_);                                                                 \
        grid.z = (unsigned)(N * (Cout / (64 * OB_)) * nsplit);                                                               \
        if (hipFuncSetAttribute(reinterpret_cast<const void*>(k), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds) !=   \
            hipSuccess) {                                                                                                    \
            set_error("%s: hipFuncSetAttribute(MaxDynamicSharedMemorySize) failed", name);                                   \
            return MVSDET_ERR_HIP;                                                                                           \
        }                                                                                                                    \
        hipLaunchKernelGGL(k, grid, dim3(TD_ * TH_ * TW_ * 2 / CG_), lds, (hipStream_t)stream, x, sN, sC, sD, sH, Cin,       \
                           static_cast<const uint4*>(x_pscl), cDp, cHp, cWp, cpiece, N,                                      \
                           static_cast<const uint4*>(weight_split), scale, shift, dst, C8, Cout, Di, Hi, Wi, D, H, W,        \
                           tiles_w, relu, nsplit, static_cast<float*>(workspace), total,                                     \
                           (int)(options().conv_xcd != 0 && ((long long)grid.x * grid.y * grid.z) % 8 == 0 &&                \
                                 (long long)grid.x * grid.y * grid.z >= 64));                                                \
    }
    // the PSCL-fed 3 x 16 x 8 tile on 12 waves of one column group (conv1 0.573 -> 0.563, conv3 0.304 -> 0.291 ms: two 6-wave
    // blocks already share a CU here; option "conv_s2_cg" = 2: the 6-wave form)
    // ("conv_s2_ob" = 1: 64 output channels per block; default: 128 where the layer has them)
    if (x_pscl) {
        if (t38) {
            if (options().conv_s2_cg == 2) MVS_S2_CASE(3, 16, 8, true, 2)
            else if (Cout % 128 == 0 && nsplit == 1 && options().conv_s2_ob != 1) MVS_S2_CASE(3, 16, 8, true, 1, 2)
            else MVS_S2_CASE(3, 16, 8, true, 1)
        } else MVS_S2_CASE(kS2TD, kS2TH, kBfW, true, 2)
    }
    else if (t38) {   // fp32 input (the training route, the neck): the same 12 waves of one column group
        if (options().conv_s2_cg == 2) MVS_S2_CASE(3, 16, 8, false, 2)
        else if (Cout % 128 == 0 && nsplit == 1 && options().conv_s2_ob != 1) MVS_S2_CASE(3, 16, 8, false, 1, 2)
        else MVS_S2_CASE(3, 16, 8, false, 1)
    } else MVS_S2_CASE(kS2TD, kS2TH, kBfW, false, 2)
#undef MVS_S2_CASE
    MVS_LAUNCH_CHECK(name);
    if (nsplit > 1) {
        launch_splitk_epilogue(static_cast<const float*>(workspace), nsplit, total, scale, shift, nullptr, out, Cout,
                               (size_t)D * H * W, relu, (hipStream_t)stream);
        MVS_LAUNCH_CHECK(name);
    }
    return MVSDET_OK;
}

extern "C" int mvsdet_conv3d_k3_s2_bf16x3_f32in(const float* x, const int64_t* x_strides, const void* weight_split,
                                                const float* scale, const float* shift, float* out, int N, int Cin, int Cout,
                                                int Di, int Hi, int Wi, int relu, mvsdet_stream_t stream) {
    return launch_s2_bf16x3(x, x_strides, nullptr, weight_split, scale, shift, out, nullptr, nullptr, 0, N, Cin, Cout, Di, Hi, Wi, relu,
                            stream);
}

// with a workspace (mvsdet_conv3d_k3_s2_bf16x3_workspace_bytes; NULL or too small: unsplit) for small volumes
extern "C" int mvsdet_conv3d_k3_s2_bf16x3_f32in_ws(const float* x, const int64_t* x_strides, const void* weight_split,
                                                   const float* scale, const float* shift, float* out, void* workspace,
                                                   size_t workspace_bytes, int N, int Cin, int Cout, int Di, int Hi, int Wi, int relu,
                                                   mvsdet_stream_t stream) {
    return launch_s2_bf16x3(x, x_strides, nullptr, weight_split, scale, shift, out, nullptr, workspace, workspace_bytes, N, Cin, Cout, Di,
                            Hi, Wi, relu, stream);
}

// The general form: input = the fp32 tensor x (x_strides) or its parity-split SCL form x_pscl (a producing layer's out_pscl:
// the class tiles then arrive by LDS-DMA); outputs out_f32 and / or out_scl (SCL form of the (N,Cout,D,H,W) result, zero border).
extern "C" int mvsdet_conv3d_k3_s2_bf16x3_io(const float* x, const int64_t* x_strides, const void* x_pscl, const void* weight_split,
                                             const float* scale, const float* shift, float* out_f32, void* out_scl, void* workspace,
                                             size_t workspace_bytes, int N, int Cin, int Cout, int Di, int Hi, int Wi, int relu,
                                             mvsdet_stream_t stream) {
    return launch_s2_bf16x3(x, x_strides, x_pscl, weight_split, scale, shift, out_f32, out_scl, workspace, workspace_bytes, N, Cin, Cout,
                            Di, Hi, Wi, relu, stream);
}

// ConvTranspose3d(Cin -> Cout = 64*m, kernel 3, stride 2, padding 1, output_padding 1, no bias) [+ affine] [+ ReLU] [+ residual,
// added last] of mvsnet.py:92-100,110-111 on the bf16 matrix cores, three-term split.  xs: SCL form of the input (N,Cin,D,H,W)
// (mvsdet_scl_pack_f32 or a producing layer's out_scl); weight_split: mvsdet_split_conv_weight_ordered(order = 2) of the
// (Cin,Cout,3,3,3) weight; out_f32 / residual (N,Cout,2D,2H,2W) fp32, 8-byte aligned; out_scl: the SCL form of the result.
static bool convT_tile38(int D, int H, int W) {
    const long long pad416 = (long long)((W + kBfW - 1) / kBfW) * kBfW * ((H + kS2TH - 1) / kS2TH) * kS2TH * ((D + kS2TD - 1) / kS2TD) * kS2TD;
    const long long pad38 = (long long)((W + 7) / 8) * 8 * ((H + 15) / 16) * 16 * ((D + 2) / 3) * 3;
    return pad38 < pad416;
}

static int launch_convT(const void* xs, const void* weight_split, const float* scale, const float* shift, const float* residual,
                        float* out, void* out_scl, int N, int Cin, int Cout, int D, int H, int W, int relu, mvsdet_stream_t stream,
                        void* stats = nullptr, size_t stats_bytes = 0, const float* stats_pivot = nullptr) {
    const char* name = "convT3d_k3_s2_bf16x3";
    MVS_REQUIRE(xs && weight_split && (out || out_scl), "%s: NULL pointer", name);
    MVS_REQUIRE((scale == nullptr) == (shift == nullptr), "%s: scale and shift come together", name);
    MVS_REQUIRE(N > 0 && Cin > 0 && D > 0 && H > 0 && W > 0, "%s: bad shape N=%d Cin=%d D=%d H=%d W=%d", name, N, Cin, D, H, W);
    MVS_REQUIRE(Cout > 0 && Cout % 64 == 0, "%s: Cout=%d must be a multiple of 64", name, Cout);
    MVS_REQUIRE((((uintptr_t)xs | (uintptr_t)weight_split | (uintptr_t)out_scl) & 15u) == 0, "%s: SCL buffers and weights must be 16-byte aligned", name);
    MVS_REQUIRE(((uintptr_t)out & 7u) == 0 && (residual == nullptr || ((uintptr_t)residual & 7u) == 0),
                "%s: out and residual must be 8-byte aligned", name);
    const BfPlan p = bf_plan(D, H, W);   // the padded extents of the SCL input
    const int C8 = (Cin + 7) / 8;
    // input tile 4 x 8 x 16, or 3 x 16 x 8 where that pads (D, H, W) less
    const bool t38 = convT_tile38(D, H, W);
    const int ttd = t38 ? 3 : kS2TD, tth = t38 ? 16 : kS2TH, ttw = t38 ? 8 : kBfW;
    const int tiles_w = (W + ttw - 1) / ttw, tiles_h = (H + tth - 1) / tth, tiles_d = (D + ttd - 1) / ttd;
    // the halo tile reaches one voxel past the last tile: padded index tiles*T + 1 must exist
    MVS_REQUIRE(tiles_d * ttd + 2 <= p.Dp && tiles_h * tth + 2 <= p.Hp && tiles_w * ttw + 2 <= p.Wp,
                "%s: the SCL padding does not cover the tiles", name);
    MVS_REQUIRE((long long)N * (Cout / 64) <= 65535 && tiles_d <= 65535, "%s: N*Cout/64 or D too large", name);
    const size_t piece = (size_t)N * C8 * p.Dp * p.Hp * p.Wp;
    const BfOut dst = make_out(out, out_scl, nullptr, N, Cout, 2 * D, 2 * H, 2 * W);
    dim3 grid((unsigned)(tiles_w * tiles_h * 4), (unsigned)tiles_d, (unsigned)(N * (Cout / 64)));
    hipStream_t st = (hipStream_t)stream;
#define MVS_CT_CASE(TD_, TH_, TW_, CG_)                                                                                      \
    {                                                                                                                        \
        const size_t lds = ct_lds_bytes(TD_, TH_, TW_, CG_);                                                                 \
        auto* k = convT3d_k3_s2_bf16x3_kernel<TD_, TH_, TW_, CG_>;                                                           \
        if (hipFuncSetAttribute(reinterpret_cast<const void*>(k), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds) !=   \
            hipSuccess) {                                                                                                    \
            set_error("%s: hipFuncSetAttribute(MaxDynamicSharedMemorySize) failed", name);                                   \
            return MVSDET_ERR_HIP;                                                                                           \
        }                                                                                                                    \
        hipLaunchKernelGGL(k, grid, dim3(TD_ * TH_ * TW_ * 2 / CG_), lds, st, static_cast<const uint4*>(xs),                 \
                           static_cast<const uint4*>(weight_split), scale, shift, residual, dst, C8, Cout, D, H, W, p.Dp, p.Hp, \
                           p.Wp, piece, tiles_w, relu);                                                                      \
    }
    // the 3 x 16 x 8 tile: all eight output parity classes in one block of 32 output channels (conv9 0.40 -> 0.30, conv11 0.66 ->
    // 0.48 ms, the same bits); option "convT_cg" = 1: one block per (PD, PH) on 12 waves of one column group, 2: on 6 waves of two
    if (stats) {
        MVS_REQUIRE(t38 && options().convT_cg != 1 && options().convT_cg != 2,
                    "%s: the statistics epilogue exists in the all-classes kernel on 3 x 16 x 8 tiles only (mvsdet_convT3d_k3_s2_bf16x3_stats_parts = 0 otherwise)", name);
        MVS_REQUIRE(!scale && !residual && !relu && out && !out_scl, "%s: statistics are those of the raw fp32 output", name);
        MVS_REQUIRE(stats_bytes >= (size_t)Cout * N * tiles_w * tiles_h * tiles_d * sizeof(double2) && ((uintptr_t)stats & 15u) == 0,
                    "%s: the statistics buffer holds Cout x mvsdet_convT3d_k3_s2_bf16x3_stats_parts double2", name);
        const size_t lds = ctf_lds_bytes(3, 16, 8);
        auto* k = convT3d_k3_s2_bf16x3_fused_kernel<3, 16, 8, true>;
        if (hipFuncSetAttribute(reinterpret_cast<const void*>(k), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds) != hipSuccess) {
            set_error("%s: hipFuncSetAttribute(MaxDynamicSharedMemorySize) failed", name);
            return MVSDET_ERR_HIP;
        }
        MVS_REQUIRE((long long)N * (Cout / 32) <= 65535, "%s: N*Cout/32 too large", name);
        dim3 fgrid((unsigned)(tiles_w * tiles_h), (unsigned)tiles_d, (unsigned)(N * (Cout / 32)));
        hipLaunchKernelGGL(k, fgrid, dim3(3 * 16 * 8 * 2), lds, st, static_cast<const uint4*>(xs), static_cast<const uint4*>(weight_split),
                           scale, shift, residual, dst, C8, Cout, D, H, W, p.Dp, p.Hp, p.Wp, piece, tiles_w, relu,
                           (int)(options().conv_xcd != 0 && ((long long)fgrid.x * fgrid.y * fgrid.z) % 8 == 0 &&
                                 (long long)fgrid.x * fgrid.y * fgrid.z >= 64),
                           static_cast<double2*>(stats), stats_pivot);
    } else
    if (t38 && residual && options().convT_cg == 0 && options().convT_persist != 0 && C8 % 4 == 0 && C8 >= 12 && Cout <= kCtpMaxCout && !(out && out_scl) &&
        (unsigned long long)N * Cout * 8ull * D * H * W * 4ull < 0x7ffffff0ull && (!out_scl || 2ull * dst.piece * 16ull < 0x7ffffff0ull)) {
        // persistent form (convt_persist.h): one block per CU, the epilogue of an item behind the loop of the next
        int dev = 0, cus = 0;
        if (hipGetDevice(&dev) != hipSuccess || hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, dev) != hipSuccess || cus < 8) {
            set_error("%s: cannot read the device's CU count", name);
            return MVSDET_ERR_HIP;
        }
        const unsigned G = (unsigned)(options().convT_persist > 1 ? options().convT_persist : cus) & ~7u;
        const size_t lds = ctp_lds_bytes();
        const unsigned out_bytes = out_scl ? (unsigned)(2ull * dst.piece * 16ull) : (unsigned)((unsigned long long)N * Cout * 8ull * D * H * W * 4ull);
        const unsigned res_bytes = (unsigned)((unsigned long long)N * Cout * 8ull * D * H * W * 4ull);
#ifndef MVS_CONVT_WHATIF
#define MVS_CONVT_WHATIF 0   // what-if BUILDS only (-DMVS_CONVT_WHATIF=n, wrong results: convt_persist.h, profiles/r06_convt_persist.txt)
#endif
#define MVS_CTP_CASE(SCL_, RES_)                                                                                             \
    {                                                                                                                        \
        auto* k = convT3d_k3_s2_bf16x3_persist_kernel<SCL_, RES_>;                                                           \
        if (hipFuncSetAttribute(reinterpret_cast<const void*>(k), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds) !=   \
            hipSuccess) {                                                                                                    \
            set_error("%s: hipFuncSetAttribute(MaxDynamicSharedMemorySize) failed", name);                                   \
            return MVSDET_ERR_HIP;                                                                                           \
        }                                                                                                                    \
        hipLaunchKernelGGL(k, dim3(G), dim3(768), lds, st, static_cast<const uint4*>(xs), static_cast<const uint4*>(weight_split), \
                           scale, shift, residual, dst, C8, Cout, D, H, W, p.Dp, p.Hp, p.Wp, piece, tiles_w, tiles_h, tiles_d, N, relu, \
                           out_bytes, res_bytes, (int)(options().conv_xcd != 0), MVS_CONVT_WHATIF);                                            \
    }
        if (out_scl) MVS_CTP_CASE(true, true)
        else MVS_CTP_CASE(false, true)
#undef MVS_CTP_CASE
    } else
    if (t38 && options().convT_cg != 1 && options().convT_cg != 2) {
        const size_t lds = ctf_lds_bytes(3, 16, 8);
        auto* k = convT3d_k3_s2_bf16x3_fused_kernel<3, 16, 8>;
        if (hipFuncSetAttribute(reinterpret_cast<const void*>(k), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds) != hipSuccess) {
            set_error("%s: hipFuncSetAttribute(MaxDynamicSharedMemorySize) failed", name);
            return MVSDET_ERR_HIP;
        }
        MVS_REQUIRE((long long)N * (Cout / 32) <= 65535, "%s: N*Cout/32 too large", name);
        dim3 fgrid((unsigned)(tiles_w * tiles_h), (unsigned)tiles_d, (unsigned)(N * (Cout / 32)));
        hipLaunchKernelGGL(k, fgrid, dim3(3 * 16 * 8 * 2), lds, st, static_cast<const uint4*>(xs), static_cast<const uint4*>(weight_split),
                           scale, shift, residual, dst, C8, Cout, D, H, W, p.Dp, p.Hp, p.Wp, piece, tiles_w, relu,
                           (int)(options().conv_xcd != 0 && ((long long)fgrid.x * fgrid.y * fgrid.z) % 8 == 0 &&
                                 (long long)fgrid.x * fgrid.y * fgrid.z >= 64),
                           static_cast<double2*>(nullptr), static_cast<const float*>(nullptr));
    } else
    if (t38) { if (options().convT_cg == 2) MVS_CT_CASE(3, 16, 8, 2) else MVS_CT_CASE(3, 16, 8, 1) }
    else MVS_CT_CASE(kS2TD, kS2TH, kBfW, 2)
#undef MVS_CT_CASE
    MVS_LAUNCH_CHECK(name);
    return MVSDET_OK;
}

extern "C" int mvsdet_convT3d_k3_s2_bf16x3(const void* xs, const void* weight_split, const float* scale, const float* shift,
                                           const float* residual, float* out, int N, int Cin, int Cout, int D, int H, int W,
                                           int relu, mvsdet_stream_t stream) {
    return launch_convT(xs, weight_split, scale, shift, residual, out, nullptr, N, Cin, Cout, D, H, W, relu, stream);
}

// The transposed layer in front of a training-mode BatchNorm (mvsnet.py:92-100 under model.train()): raw fp32 output + per-channel
// partial sums as mvsdet_conv3d_k3_bf16x3_stats leaves them (sums of value - pivot_c and of its square, one double2 per channel and
// block; parts = mvsdet_convT3d_k3_s2_bf16x3_stats_parts(N, D, H, W) of the COARSE input extents, 0 = this shape has no such form).
extern "C" size_t mvsdet_convT3d_k3_s2_bf16x3_stats_parts(int N, int D, int H, int W) {
    if (N <= 0 || D <= 0 || H <= 0 || W <= 0) return 0;
    if (!convT_tile38(D, H, W) || options().convT_cg == 1 || options().convT_cg == 2) return 0;
    return (size_t)N * ((W + 7) / 8) * ((H + 15) / 16) * ((D + 2) / 3);
}

extern "C" int mvsdet_convT3d_k3_s2_bf16x3_stats(const void* xs, const void* weight_split, float* out_f32, void* stats, size_t stats_bytes,
                                                 const float* pivot, int N, int Cin, int Cout, int D, int H, int W, mvsdet_stream_t stream) {
    MVS_REQUIRE(stats != nullptr, "convT3d_k3_s2_bf16x3_stats: NULL statistics buffer");
    return launch_convT(xs, weight_split, nullptr, nullptr, nullptr, out_f32, nullptr, N, Cin, Cout, D, H, W, 0, stream, stats, stats_bytes, pivot);
}

extern "C" int mvsdet_convT3d_k3_s2_bf16x3_io(const void* xs, const void* weight_split, const float* scale, const float* shift,
                                              const float* residual, float* out_f32, void* out_scl, int N, int Cin, int Cout, int D,
                                              int H, int W, int relu, mvsdet_stream_t stream) {
    return launch_convT(xs, weight_split, scale, shift, residual, out_f32, out_scl, N, Cin, Cout, D, H, W, relu, stream);
}

// ---------------------------------------------------------------------------------------------------------------
// conv0 on ONE fp16 and TWO block-scaled FP6 (OCP MX e2m3) products per fp32-equivalent product (costreg_mx.h): the fp32
// (N,Cin,D,H,W) tensor read in place (x_strides as for mvsdet_conv3d_k3_bf16x3_f32in), weights from mvsdet_split_conv_weight_mx,
// outputs as for mvsdet_conv3d_k3_bf16x3_io.  Values within ~2^-15 relative of the exact convolution (bf16x3: 2^-16).
// ---------------------------------------------------------------------------------------------------------------
extern "C" size_t mvsdet_split_conv_weight_mx_bytes(int Cout, int Cin) {
    if (Cout <= 0 || Cout % 64 || Cin <= 0) return 0;
    return (size_t)(Cout / 64) * ((Cin + 7) / 8) * 2 * kMxSlots * 64 * 16;
}

extern "C" int mvsdet_split_conv_weight_mx(const float* weight, void* weight_split, int Cout, int Cin, mvsdet_stream_t stream) {
    MVS_REQUIRE(weight && weight_split, "split_conv_weight_mx: NULL pointer");
    MVS_REQUIRE(Cout > 0 && Cout % 64 == 0 && Cin > 0, "split_conv_weight_mx: Cout=%d must be a positive multiple of 64, Cin=%d > 0", Cout, Cin);
    MVS_REQUIRE(((uintptr_t)weight_split & 15u) == 0, "split_conv_weight_mx: output must be 16-byte aligned");
    const size_t units = mvsdet_split_conv_weight_mx_bytes(Cout, Cin) / 16;
    hipLaunchKernelGGL(split_conv_weight_mx_kernel, dim3((unsigned)((units + kThreads - 1) / kThreads)), dim3(kThreads), 0,
                       (hipStream_t)stream, weight, static_cast<uint4*>(weight_split), Cin, (Cin + 7) / 8, units);
    MVS_LAUNCH_CHECK("split_conv_weight_mx");
    return MVSDET_OK;
}

extern "C" int mvsdet_conv3d_k3_fp16mx_f32in(const float* x, const int64_t* x_strides, const void* weight_split_mx, const float* scale,
                                             const float* shift, float* out_f32, void* out_scl, void* out_pscl, int N, int Cin, int Cout,
                                             int D, int H, int W, int relu, mvsdet_stream_t stream) {
    const char* name = "conv3d_k3_fp16mx_f32in";
    MVS_REQUIRE(x && weight_split_mx && (out_f32 || out_scl || out_pscl), "%s: NULL pointer", name);
    MVS_REQUIRE((scale == nullptr) == (shift == nullptr), "%s: scale and shift come together", name);
    MVS_REQUIRE(N > 0 && Cin > 0 && D > 0 && H > 0 && W > 0, "%s: bad shape N=%d Cin=%d D=%d H=%d W=%d", name, N, Cin, D, H, W);
    MVS_REQUIRE(Cout > 0 && Cout % 64 == 0, "%s: Cout=%d must be a multiple of 64", name, Cout);
    MVS_REQUIRE((((uintptr_t)weight_split_mx | (uintptr_t)out_scl | (uintptr_t)out_pscl) & 15u) == 0, "%s: SCL buffers and weights must be 16-byte aligned", name);
    // option "conv_mx_th": 0 (default) = the wave-specialised kernel (4 x 8 x 16 tiles: 8 multiplying + 4 staging waves: 3.89 ms at conv0);
    // 8 = every wave does everything on 4 x 8 x 16 tiles (8 waves: the kernel wants ~190 registers per lane, which two waves per SIMD
    // have: 4.02); 12 = the same on 4 x 12 x 16 tiles (12 waves, 168 registers: spills, 4.1-4.7).  Same values bit for bit.
    BfPlan p = bf_plan(D, H, W);
    const int mx_form = options().conv_mx_th;
    if (mx_form != 12) { p.th = 8; p.tiles_h = (H + 7) / 8; }
    const int C8 = (Cin + 7) / 8;
    const size_t vol = (size_t)D * H * W;
    MVS_REQUIRE((long long)N * (Cout / 64) <= 65535 && p.tiles_d <= 65535, "%s: N*Cout/64 or D too large", name);
    const long long sN = x_strides ? x_strides[0] : (long long)Cin * vol, sC = x_strides ? x_strides[1] : (long long)vol;
    const long long sD = x_strides ? x_strides[2] : (long long)H * W, sH = x_strides ? x_strides[3] : (long long)W;
    MVS_REQUIRE(sN >= 0 && sC >= 0 && sD >= 0 && sH >= W, "%s: bad strides", name);
    MVS_REQUIRE((long long)(D - 1) * sD + (long long)(H - 1) * sH + W < (1LL << 31), "%s: one channel volume spans more than 2^31 elements", name);
    const BfOut dst = make_out(out_f32, out_scl, out_pscl, N, Cout, D, H, W);
    dim3 grid((unsigned)(p.tiles_w * p.tiles_h), (unsigned)p.tiles_d, (unsigned)(N * (Cout / 64)));
    const int xcd_map = options().conv_xcd != 0 && ((long long)grid.x * grid.y * grid.z) % 8 == 0 && (long long)grid.x * grid.y * grid.z >= 64;
    hipStream_t st = (hipStream_t)stream;
#define MVS_MX_CASE(TH_)                                                                                                     \
    {                                                                                                                        \
        auto* k = conv3d_k3_fp16mx_kernel<4, TH_>;                                                                      \
        const size_t lds = ((size_t)2 * 2 * bf_in_slots(4, TH_, kBfW) + 2 * kMxSlots * 64) * 16;                             \
        if (hipFuncSetAttribute(reinterpret_cast<const void*>(k), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds) != hipSuccess) { \
            set_error("%s: hipFuncSetAttribute(MaxDynamicSharedMemorySize) failed", name);                                  \
            return MVSDET_ERR_HIP;                                                                                           \
        }                                                                                                                    \
        hipLaunchKernelGGL(k, grid, dim3(4 * TH_ * 16), lds, st, x, sN, sC, sD, sH, Cin, static_cast<const uint4*>(weight_split_mx), \
                           scale, shift, dst, C8, Cout, D, H, W, p.tiles_w, relu, xcd_map);                                  \
    }
    if (mx_form == 0) {
        auto* k = conv3d_k3_fp16mx_ws_kernel<4, 8>;
        const size_t lds = ((size_t)2 * 2 * bf_in_slots(4, 8, kBfW) + 2 * kMxSlots * 64) * 16;
        if (hipFuncSetAttribute(reinterpret_cast<const void*>(k), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds) != hipSuccess) {
            set_error("%s: hipFuncSetAttribute(MaxDynamicSharedMemorySize) failed", name);
            return MVSDET_ERR_HIP;
        }
        hipLaunchKernelGGL(k, grid, dim3(4 * 8 * 16 + 256), lds, st, x, sN, sC, sD, sH, Cin, static_cast<const uint4*>(weight_split_mx), scale,
                           shift, dst, C8, Cout, D, H, W, p.tiles_w, relu, xcd_map);
    } else if (p.th == 12) MVS_MX_CASE(12) else MVS_MX_CASE(8)
#undef MVS_MX_CASE
    MVS_LAUNCH_CHECK(name);
    return MVSDET_OK;
}

