// Backward of the fused plane-sweep variance (a3+a4) with respect to the 2-D features -- the only input
// with a gradient: the sampling grid is built under torch.no_grad() (mvs_models/module.py:115).
//
//   var = Q/(K+1) - (S/(K+1))^2,  S = f + sum_j w_j,  Q = f^2 + sum_j w_j^2
//   dvar/dv = 2 v r - 2 S r^2   (r = 1/(K+1)) for each contributing value v in {f, w_1..w_K}
//   w_j = sum_t weight_t * tap_t  ->  dL/dtap_t += weight_t * dL/dw_j   (bilinear scatter)
//
// Same tiles, decode duty and sweep geometry as the forward slab kernel (sweep_kernel.h): block = (reference view,
// 128-pixel tile, 32-channel slab); the sampling positions are recomputed per plane from the pixel rays (one position
// per lane and pass, DPP broadcasts).  Where the forward keeps a neighbour's footprint box of SOURCE texels resident
// in LDS for a run of planes, the backward keeps a GRADIENT image of the same box there: tap gradients are
// accumulated with LDS atomics over the whole run and flushed to the packed gradient map with global atomics only when
// the run's box changes (a run is ~9-35 planes long).
//
// The gradient image is kept in DOUBLE.  tools/micro/lds_atomic.hip: on gfx950 ds_add_f32 retires 0.33 lane-op per
// clock and CU whatever the address pattern (one lane every three clocks -- 2.8 G tap gradients = 11.6 of round 2's
// 13.5 ms), ds_add_f64 4-8 and ds_add_u32 15.  So one kernel instance owns HALF a slab -- the packed floats
// [16*HALF, 16*HALF + 16) of every texel, lane (ps, g) the two floats 2g, 2g+1 of its 4 consecutive pixels -- which
// makes a texel of the image 16 doubles = the 128 bytes the fp32 image had, and the two instances (two launches)
// together repeat the position decode and the tap gathers (8 bytes per lane, from L2) but not the arithmetic or the
// atomics.  The sums are also more accurate than fp32 ones and lose nothing when a box collects thousands of taps.
// A footprint larger than the box falls back to one global fp32 atomic per tap.  The packed gradient map is unpacked
// to (N,C,H,W) afterwards.
//
// What bounds it (round 4, rocprofv3 --pmc at the reference-true shape, tools/pmc_bwd.sh): NOT the LDS atomics -- the same
// kernel with the image in 64-bit fixed point (ds_add_u64 costs half a ds_add_f64: tools/micro/lds_atomic_mask.hip) waits a
// quarter as long on the LDS (SQ_WAIT_INST_LDS 1.8e8 -> 4.5e7) and is 3 % faster, less than the pass over dL/dvar that finds
// its scale costs.  With two waves per SIMD the waves spend 54 % of their cycles waiting (SQ_WAIT_ANY / SQ_WAVE_CYCLES) on
// dL/dvar from HBM and the taps from L2 while the vector pipe is 63 % busy.  GROUPS = 2 (D >= 32) doubles the waves on the
// same LDS: wave group q of a block takes the planes d with d % GROUPS == q, both add into the SAME gradient images (sums
// commute) and meet at the refills: 55.3 -> 44.4 ms at 64 planes of 120 x 160; at 12 planes it loses (2.8 -> 3.1 ms: six
// planes per group between a block's zeroing and flush), so short sweeps keep one group.
#include "common.h"

#include <algorithm>

#include "pack.h"
#include "sweep_kernel.h"

namespace mvsdet {

// LDS bank swizzle of the gradient images: double 2*g + i of the texel at double index `base` (a multiple of 16) is kept
// at 2*g + (i ^ ((base >> 6) & 1)).  One LDS atomic of a wave touches a fixed i of 8 texels x 8 lanes, 16 bytes apart:
// half of the banks; the texels of neighbouring pixel slots lie ~4 slots apart, so flipping i with bit 2 of the slot
// index spreads them over all of them.
__device__ __forceinline__ int grad_swizzle(int base) { return (base >> 6) & 1; }

constexpr int kHalfSlab = kSlab / 2;  // packed floats of a texel one kernel instance owns

// byte offset, inside the LDS images, of the FIRST double of a lane's pair (add 16 * g) of the texel at double index `base`, swizzle
// applied: doubles (2g, 2g+1) of the texel live at 2g + (0 ^ sw), 2g + (1 ^ sw), i.e. at this address and at it with bit 3 flipped
__device__ __forceinline__ int grad_byte(int base) { return base * 8 + 8 * grad_swizzle(base); }
__device__ __forceinline__ void lds_add_f64_at(int addr, double v) {
    __hip_atomic_fetch_add((__attribute__((address_space(3))) double*)(size_t)(unsigned)addr, v, __ATOMIC_RELAXED,
                           __HIP_MEMORY_SCOPE_WORKGROUP);
}

template <int K, int TW, int HALF, int GROUPS>
__global__ __launch_bounds__(kThreads * GROUPS, 2 * GROUPS) void plane_sweep_variance_bwd_kernel(
    const float* __restrict__ packed, const int64_t* __restrict__ nbr, const int4* __restrict__ header,
    const float* __restrict__ proj, const float* __restrict__ depth, const int4* __restrict__ boxes,
    const unsigned* __restrict__ flags, const float* __restrict__ gvar, float* __restrict__ gpacked, int N, int C, int S, int D,
    int H, int W, int tiles_x, int tiles, int box_cap) {
    constexpr int KK = K > 0 ? K : 1;
    constexpr int NP = (K + 1) / 2;
    constexpr int NPP = NP > 0 ? NP : 1;
    constexpr int TH = kTilePix / TW;
    extern __shared__ double s_grad[];  // K slots of (box_cap + kBoxPad) texels x 16 doubles: gradient images of the resident boxes

    const int HW = H * W;
    if constexpr (K > 0) {
        // The geometry may come from the FORWARD pass (mvsdet_plane_sweep_variance_bwd_packed_f32): its layout hangs on the tile
        // shape of the launch that built it, and its boxes must fit this launch's LDS slots (sized for the largest capacity a
        // geometry of this shape can have, so "sweep_boxcap" may move in between).  When `mvsdet_set_option` moved the tile
        // shape in between (or the buffer is not a geometry at all) nothing of it is read: the whole gradient becomes NaN (block-uniform exit before any
        // barrier), as the forward kernel does with its output.
        const int4 hd = *header;
        if (hd.x != kGeoMagic || (hd.y & 0xff) != TW || (hd.y >> 8) > box_cap || hd.z != W || hd.w != ((D << 8) | K)) {
            const size_t total = (size_t)N * S * HW * kSlab;
            const size_t step = (size_t)gridDim.x * blockDim.x;
            for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += step) gpacked[i] = __builtin_nanf("");
            return;
        }
    }
    // HALF == 2: both halves in one launch; blocks 16k .. 16k+7 are the lower halves of eight (view, tile, slab) units and
    // blocks 16k+8 .. 16k+15 the upper halves of the same eight, so a unit's two halves run on the same XCD (block id modulo
    // 8) at about the same time and the second finds the 128-byte texel lines of the first in its L2
    const int half = HALF == 2 ? (int)((blockIdx.x >> 3) & 1) : HALF;
    const int id = HALF == 2 ? (int)((blockIdx.x >> 4) * 8 + (blockIdx.x & 7)) : (int)blockIdx.x;
    if (HALF == 2 && id >= N * tiles * S) return;
    const int slab = id % S;
    const int bt = id / S;
    const int tile = bt % tiles, n = bt / tiles;
    const int tx0 = (tile % tiles_x) * TW, ty0 = (tile / tiles_x) * TH;
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave_of_block = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wave = wave_of_block & 3, plane_group = wave_of_block >> 2;
    const int g = lane & 7, ps = lane >> 3;
    const size_t slab_stride = (size_t)HW * kSlab;
    // float2 views of the slab images, already at the lane's two floats: texel t is element 16 * t
    const float2* ref_img = reinterpret_cast<const float2*>(packed + ((size_t)n * S + slab) * slab_stride) + 8 * half + g;
    // Tap offsets travel in BYTES (as in the forward kernel): the slab bases are block-uniform (scalar registers), the decoding lane
    // sends the byte offset of the texel, and the receiving lane's address is ONE v_add_u32_dpp (quad broadcast + its own bytes of the
    // texel) -- a gather is `global_load_dwordx2 v, v_off, s[base]`, an LDS add takes its address ready-made, the second double of
    // a lane's pair is that address with bit 3 flipped, the two gradients of a tap are one packed product: 56 -> 29 vector
    // instructions per pixel step of pass 2, 17 -> 13 of pass 1.  Round 4 measured this form at no gain (the waves waited on memory
    // 54 % of their cycles); round 5, with two planes of dL/dvar in flight: 2.39 -> 2.31 ms at 12 planes of 60 x 80, 6.27 -> 6.10 at
    // 100 views, 44.7 -> 39.2 ms at 64 planes of 120 x 160 (same box, tools/ab_libs.sh) -- PROVIDED the 16 gathers of a neighbour are
    // requested ahead of the first use (below): left to the scheduler, the first neighbour's became load - wait - 3 loads - wait per
    // pixel step under the register budget (eight exposed L2 round trips instead of one: 2.56 ms).
    const char* nb_img[KK];
    float* nb_grad[KK];
#pragma unroll
    for (int j = 0; j < K; ++j) {
        int64_t v = nbr[(size_t)n * K + j];
        v = v < 0 ? 0 : (v >= N ? N - 1 : v);
        nb_img[j] = reinterpret_cast<const char*>(packed + ((size_t)v * S + slab) * slab_stride);
        nb_grad[j] = gpacked + ((size_t)v * S + slab) * slab_stride + kHalfSlab * half;
    }
    const int lane_b = (8 * half + g) * 8;   // the lane's float2 of a texel of the slab image, in bytes
    const int g8 = g * 8;                    // ... of a texel's half of the gradient map (nb_grad is already at the half)
    const int g16_lds = g * 16 + (int)(unsigned)(size_t)(__attribute__((address_space(3))) char*)s_grad;   // the lane's two doubles of a texel of the LDS image
    const float r = 1.0f / (float)(K + 1);
    const float two_r = 2.0f * r, two_r2 = 2.0f * r * r;
    const int slot_el = (box_cap + kBoxPad) * kHalfSlab;  // doubles per LDS slot

    // the lane's 4 consecutive pixels, their reference features and the running reference-term gradient
    const int p0 = 32 * wave + 4 * ps;
    const int px0 = tx0 + p0 % TW, py = ty0 + p0 / TW;
    float f[4][2], gref[4][2];
    bool pok[4];
#pragma unroll
    for (int s = 0; s < 4; ++s) {
        pok[s] = (px0 + s < W) && (py < H);
        const int pix = min(py, H - 1) * W + min(px0 + s, W - 1);
        const float2 v = ref_img[(size_t)pix * kHalfSlab];
        f[s][0] = v.x; f[s][1] = v.y;
        gref[s][0] = gref[s][1] = 0.0f;
    }
    const int st_n = (py < H) ? max(0, min(4, W - px0)) : 0;
    const bool st_vec = (st_n == 4) && ((W & 3) == 0) && ((HW & 3) == 0);
    // packed float q = 16*HALF + 2g + i is channel 8*(q & 3) + (q >> 2) of the slab (pack.h)
    size_t g_off[2];
    bool g_ok[2];
#pragma unroll
    for (int i = 0; i < 2; ++i) {
        const int c = slab * kSlab + 16 * (g & 1) + 8 * i + 4 * half + (g >> 1);
        g_ok[i] = (c < C) && (st_n > 0);
        g_off[i] = ((size_t)n * C + min(c, C - 1)) * D * HW + (size_t)py * W + px0;
    }

    // decode duty of the lane (sweep_kernel.h): pixel p0 + (g & 3), neighbour 2*pass + (g >> 2)
    const int sd = g & 3, qd = g >> 2;
    const int dx = px0 + sd;
    const bool d_inside = (dx < W) && (py < H);
    SampleRay ray[NPP];
    float tr0[NPP], tr1[NPP], tr2[NPP];
#pragma unroll
    for (int p = 0; p < NP; ++p) {
        const int jd = min(2 * p + qd, K - 1);
        const float* P = proj + ((size_t)n * K + jd) * 16;
        ray[p] = sample_ray(P, (float)dx, (float)py);
        tr0[p] = P[3]; tr1[p] = P[7]; tr2[p] = P[11];
    }
    int lx0[NPP], lx1[NPP], ly0[NPP], ly1[NPP];   // resident box of the lane's neighbour
    int rx0[KK], ry0[KK], rx1[KK], ry1[KK];       // resident box per neighbour (block-uniform)
    bool have[KK];
#pragma unroll
    for (int p = 0; p < NPP; ++p) { lx0[p] = 0; lx1[p] = 0; ly0[p] = 0; ly1[p] = 0; }
#pragma unroll
    for (int j = 0; j < KK; ++j) { rx0[j] = 0; ry0[j] = 0; rx1[j] = -1; ry1[j] = -1; have[j] = false; }

    // the gradient image of neighbour j's resident box -> packed gradient map, then zero again.  Four box texels per
    // wave-instruction, 64 contiguous bytes each.
    auto flush_box = [&](int j) {
        const int nc = rx1[j] - rx0[j] + 1, ntex = nc * (ry1[j] - ry0[j] + 1);
        const float inv_nc = 1.0f / (float)nc;
        for (int t0 = wave_of_block * 4; t0 < ntex; t0 += 16 * GROUPS) {
            const int t = t0 + (lane >> 4), q = lane & 15;
            if (t < ntex) {
                const int row = (int)(((float)t + 0.5f) * inv_nc), col = t - row * nc;
                const int base = j * slot_el + box_slot(t) * kHalfSlab;
                double* cell = s_grad + base + (q ^ grad_swizzle(base));
                const float v = (float)*cell;
                if (v != 0.0f) atomicAdd(nb_grad[j] + ((size_t)(ry0[j] + row) * W + (rx0[j] + col)) * kSlab + q, v);
                *cell = 0.0;
            }
        }
    };

    for (int e = tid; e < K * slot_el / 2; e += kThreads * GROUPS) reinterpret_cast<double2*>(s_grad)[e] = make_double2(0.0, 0.0);
    __syncthreads();

    const unsigned* fl_bt = flags + (size_t)bt * D;
    const float* depth_n = depth + (size_t)n * D;
    // dL/dvar of one plane, the lane's 4 pixels of its 2 channels.  One wave group keeps TWO planes in flight (gnext = the next
    // plane's, gahead = the one after it), see the request below
    float4 gnext[2], gahead[2];
    auto load_go = [&](int d) {
#pragma unroll
        for (int i = 0; i < 2; ++i) {
            float4 v = make_float4(0.f, 0.f, 0.f, 0.f);
            if (g_ok[i]) {
                const float* src = gvar + g_off[i] + (size_t)d * HW;
                if (st_vec) {
                    v = *reinterpret_cast<const float4*>(src);
                } else {
                    v.x = src[0];
                    if (st_n > 1) v.y = src[1];
                    if (st_n > 2) v.z = src[2];
                    if (st_n > 3) v.w = src[3];
                }
            }
            gnext[i] = v;
        }
    };
    // two planes in flight where the registers allow it without costing a block per CU (K <= 2: 156 VGPRs; K = 3 would go from 169 to 176)
    constexpr bool kAhead2 = GROUPS == 1 && K <= 2;
    if constexpr (kAhead2) {   // planes 0 (-> gnext) and 1 (-> gahead)
        if (D > 1) load_go(1);
        gahead[0] = gnext[0]; gahead[1] = gnext[1];
        load_go(0);
    } else if constexpr (GROUPS == 1) {
        load_go(0);
    }
    // flags and depth of a plane are requested a plane ahead (scalars: two SGPRs), as in the forward kernel
    unsigned fl_next = K > 0 ? fl_bt[0] : 0u;
    float dv_next = K > 0 ? depth_n[0] : 0.0f;
    for (int d = 0; d < D; ++d) {
        const unsigned fl = __builtin_amdgcn_readfirstlane(fl_next);
        const float dval = __int_as_float(__builtin_amdgcn_readfirstlane(__float_as_int(dv_next)));
        if (K > 0 && d + 1 < D) { fl_next = fl_bt[d + 1]; dv_next = depth_n[d + 1]; }
        bool refill = false;
#pragma unroll
        for (int j = 0; j < K; ++j)
            if ((fl >> (4 * j)) & kFlagStaged)
                if (((fl >> (4 * j)) & kFlagRefill) || !have[j]) refill = true;
        if (refill) {
            __syncthreads();  // every wave has added its taps of the planes that used the old boxes
#pragma unroll
            for (int j = 0; j < K; ++j) {
                if (((fl >> (4 * j)) & kFlagStaged) && (((fl >> (4 * j)) & kFlagRefill) || !have[j])) {
                    if (have[j]) flush_box(j);
                    const int4 b = boxes[((size_t)bt * D + d) * K + j];
                    rx0[j] = __builtin_amdgcn_readfirstlane(b.x);
                    rx1[j] = __builtin_amdgcn_readfirstlane(b.y);
                    ry0[j] = __builtin_amdgcn_readfirstlane(b.z);
                    ry1[j] = __builtin_amdgcn_readfirstlane(b.w);
                    have[j] = true;
                    if (min(2 * (j / 2) + qd, K - 1) == j) { lx0[j / 2] = rx0[j]; lx1[j / 2] = rx1[j]; ly0[j / 2] = ry0[j]; ly1[j / 2] = ry1[j]; }
                }
            }
            __syncthreads();  // slots zeroed again before anyone adds into them
        }
        if (GROUPS > 1 && d % GROUPS != plane_group) continue;   // the refills above are everybody's, the plane is one group's
        // ---- dL/dvar of the lane's pixels and channels (8 channel rows x 64 or 128 contiguous bytes per wave-instruction).
        //      Loads return IN ORDER (vmcnt is one in-order counter): a request for a later plane's dL/dvar issued just ahead of
        //      this plane's tap gathers turns the wait for the gathers -- L2 hits -- into a wait for that HBM read as well.  So one
        //      wave group requests plane d+2 BEHIND this plane's gathers (below, after pass 1) and consumes it two planes later:
        //      it has pass 2 of this plane and the decode and gather latency of the next one to arrive, and the next plane's
        //      gather wait finds it done.  Round 5: 2.70 -> 2.34 ms at 12 planes (one plane ahead, requested here: the 2.70; one
        //      plane ahead behind the gathers 2.85; three in flight 2.39; 156 VGPRs, still three blocks per CU).  Two groups have
        //      128 VGPRs and spill with it: they request their plane here.
        float go[4][2];
        if constexpr (GROUPS > 1) load_go(d);
#pragma unroll
        for (int i = 0; i < 2; ++i) { go[0][i] = gnext[i].x; go[1][i] = gnext[i].y; go[2][i] = gnext[i].z; go[3][i] = gnext[i].w; }
        if constexpr (GROUPS == 1 && !kAhead2)
            if (d + 1 < D) load_go(d + 1);   // K > 2: one plane ahead, requested here (round 4's form)
        // ---- pass 1 over the neighbours: warped values (taps gathered from the slab images), S.  The lane keeps what it
        //      decoded (tap offsets in the gradient slot / gradient image, weights) for pass 2, which fetches it again
        //      by DPP instead of holding the broadcast copies of every neighbour.
        float S_[4][2], wv[KK][4][2];
        int dox[NPP][4];
        float dwx[NPP][4];
#pragma unroll
        for (int s = 0; s < 4; ++s) { S_[s][0] = f[s][0]; S_[s][1] = f[s][1]; }
#define MVS_BWD_TAP(O) (*reinterpret_cast<const float2*>(nb_img[j] + (size_t)(unsigned)(O)))
#define MVS_BWD_LOAD(SS)                                                                                              \
        {                                                                                                             \
            int i0, i1, i2, i3;                                                                                       \
            quad_bcast_add4<SS>(ri0, ri1, ri2, ri3, lane_b, i0, i1, i2, i3);                                          \
            tq[SS][0] = MVS_BWD_TAP(i0); tq[SS][1] = MVS_BWD_TAP(i1); tq[SS][2] = MVS_BWD_TAP(i2); tq[SS][3] = MVS_BWD_TAP(i3); \
        }
#define MVS_BWD_MATH(SS)                                                                                              \
        {                                                                                                             \
            const float w0 = __int_as_float(quad_bcast<SS>(rw0)), w1 = __int_as_float(quad_bcast<SS>(rw1));           \
            const float w2 = __int_as_float(quad_bcast<SS>(rw2)), w3 = __int_as_float(quad_bcast<SS>(rw3));           \
            const float2 t0 = tq[SS][0], t1 = tq[SS][1], t2 = tq[SS][2], t3 = tq[SS][3];                              \
            const float a0[2] = {t0.x, t0.y}, a1[2] = {t1.x, t1.y}, a2[2] = {t2.x, t2.y}, a3[2] = {t3.x, t3.y};       \
            _Pragma("unroll") for (int i = 0; i < 2; ++i) {                                                           \
                float v = a0[i] * w0;                                                                                 \
                v = fmaf(a1[i], w1, v);                                                                               \
                v = fmaf(a2[i], w2, v);                                                                               \
                v = fmaf(a3[i], w3, v);                                                                               \
                wv[j][SS][i] = v;                                                                                     \
                S_[SS][i] += v;                                                                                       \
            }                                                                                                         \
        }
#define MVS_BWD_OF(QQ)                                                                                                \
        if constexpr (2 * p + QQ < K) {                                                                               \
            constexpr int j = 2 * p + QQ;                                                                             \
            if ((fl >> (4 * j)) & kFlagLive) {                                                                        \
                const int ri0 = from_quad<QQ>(di0), ri1 = from_quad<QQ>(di1), ri2 = from_quad<QQ>(di2), ri3 = from_quad<QQ>(di3); \
                const int rw0 = from_quad<QQ>(__float_as_int(dw.x)), rw1 = from_quad<QQ>(__float_as_int(dw.y));       \
                const int rw2 = from_quad<QQ>(__float_as_int(dw.z)), rw3 = from_quad<QQ>(__float_as_int(dw.w));       \
                /* the 16 gathers of the neighbour are requested before the first of them is used: one exposed L2 round */ \
                /* trip per neighbour, not four (the scheduler sinks the requests back to their uses otherwise)         */ \
                float2 tq[4][4];                                                                                      \
                MVS_BWD_LOAD(0) MVS_BWD_LOAD(1) MVS_BWD_LOAD(2) MVS_BWD_LOAD(3)                                       \
                __builtin_amdgcn_sched_barrier(0);                                                                    \
                MVS_BWD_MATH(0) MVS_BWD_MATH(1) MVS_BWD_MATH(2) MVS_BWD_MATH(3)                                       \
            } else {                                                                                                  \
                _Pragma("unroll") for (int s = 0; s < 4; ++s) wv[j][s][0] = wv[j][s][1] = 0.0f;                       \
            }                                                                                                         \
        }
        auto pass = [&](auto pc) {
            constexpr int p = decltype(pc)::value;
            const unsigned fp = fl >> (8 * p);
            const unsigned fq = qd ? ((2 * p + 1 < K) ? (fp >> 4) : fp) : fp;
            const bool l_staged = (fq & kFlagStaged) != 0;
            float2 e = make_float2(kNoSample, kNoSample);
            if (d_inside) e = sample_at(ray[p], tr0[p], tr1[p], tr2[p], dval, H, W);
            const SampleTaps tp = decode_sample(e.x, e.y, H, W);
            const float4 dw = tap_weights(tp);
            // taps clamped into the resident box (staged) or the image: an invalid tap carries weight 0, so reading a
            // clamped texel and adding 0 to it is harmless; a valid one lies inside the box by construction
            const int lox = l_staged ? lx0[p] : 0, hix = l_staged ? lx1[p] : W - 1;
            const int loy = l_staged ? ly0[p] : 0, hiy = l_staged ? ly1[p] : H - 1;
            const int xa = clampi(tp.x0, lox, hix), xb = clampi(tp.x0 + 1, lox, hix);
            const int ya = clampi(tp.y0, loy, hiy), yb = clampi(tp.y0 + 1, loy, hiy);
            // byte offsets of the texels in the slab image (and in the gradient map, which has its layout)
            const int di0 = (ya * W + xa) * (kSlab * 4), di1 = (ya * W + xb) * (kSlab * 4);
            const int di2 = (yb * W + xa) * (kSlab * 4), di3 = (yb * W + xb) * (kSlab * 4);
            const int pitch = hix - lox + 1;
            const int sbase = min(2 * p + qd, K - 1) * slot_el;
            const int ta = (ya - loy) * pitch - lox, tb = (yb - loy) * pitch - lox;
            // staged: byte offset of the texel in the LDS image with the bank swizzle already applied to the lane's FIRST double
            // (grad_byte: the second one is that address with bit 3 flipped); else the byte offset in the gradient map
            dox[p][0] = l_staged ? grad_byte(box_slot(ta + xa) * kHalfSlab + sbase) : di0;
            dox[p][1] = l_staged ? grad_byte(box_slot(ta + xb) * kHalfSlab + sbase) : di1;
            dox[p][2] = l_staged ? grad_byte(box_slot(tb + xa) * kHalfSlab + sbase) : di2;
            dox[p][3] = l_staged ? grad_byte(box_slot(tb + xb) * kHalfSlab + sbase) : di3;
            dwx[p][0] = dw.x; dwx[p][1] = dw.y; dwx[p][2] = dw.z; dwx[p][3] = dw.w;
            MVS_BWD_OF(0)
            MVS_BWD_OF(1)
        };
        if constexpr (NP > 0) pass(std::integral_constant<int, 0>{});
        if constexpr (NP > 1) pass(std::integral_constant<int, 1>{});
#undef MVS_BWD_OF
#undef MVS_BWD_LOAD
#undef MVS_BWD_MATH
#undef MVS_BWD_TAP
        // ---- dL/dvar two planes ahead, behind this plane's gathers (see above): gnext <- plane d+1's, gahead <- the new request
        if constexpr (kAhead2) {
            gnext[0] = gahead[0]; gnext[1] = gahead[1];
            if (d + 2 < D) {
                const float4 k0 = gnext[0], k1 = gnext[1];
                load_go(d + 2);
                gahead[0] = gnext[0]; gahead[1] = gnext[1];
                gnext[0] = k0; gnext[1] = k1;
            }
        }
        // ---- reference term, kept in registers across the planes
#pragma unroll
        for (int s = 0; s < 4; ++s)
#pragma unroll
            for (int i = 0; i < 2; ++i)
                if (pok[s]) gref[s][i] = fmaf(go[s][i], two_r * f[s][i] - two_r2 * S_[s][i], gref[s][i]);
        // ---- pass 2: tap gradients of every live neighbour
        // A step = the 4 taps of the lane's pixel SS of neighbour j, both channels of the lane: the 4 addresses are one v_add_u32_dpp each
        // (quad broadcast of the decoding lane's byte offset + the lane's own bytes), the two gradients of a tap one packed product.
        // LDS adds are unconditional: an invalid tap (weight 0) and a pixel outside the map (gradient 0) add 0.0 to a clamped
        // texel of the box -- a test per tap would put every one of the 64 ds_add_f64 of a plane under its own branch.  Larger
        // than the box: one global fp32 atomic per tap.
#define MVS_GRAD_STEP_LDS(SS)                                                                                         \
        {                                                                                                             \
            int a0, a1, a2, a3;                                                                                       \
            quad_bcast_add4<SS>(ro[0], ro[1], ro[2], ro[3], g16_lds, a0, a1, a2, a3);                                 \
            const f2 gw = pok[SS] ? (f2){go[SS][0], go[SS][1]} * ((f2){wv[j][SS][0], wv[j][SS][1]} * splat(two_r) -   \
                                                                   (f2){S_[SS][0], S_[SS][1]} * splat(two_r2))          \
                                  : splat(0.0f);                                                                      \
            const int aa[4] = {a0, a1, a2, a3};                                                                       \
            _Pragma("unroll") for (int t = 0; t < 4; ++t) {                                                           \
                const f2 pr = gw * splat(__int_as_float(quad_bcast<SS>(rw[t])));                                      \
                lds_add_f64_at(aa[t], (double)pr.x);                                                                  \
                lds_add_f64_at(aa[t] ^ 8, (double)pr.y);                                                              \
            }                                                                                                         \
        }
#define MVS_GRAD_STEP_GLB(SS)                                                                                         \
        if (pok[SS]) {                                                                                                \
            _Pragma("unroll") for (int t = 0; t < 4; ++t) {                                                           \
                const float tw = __int_as_float(quad_bcast<SS>(rw[t]));                                               \
                float* cell = reinterpret_cast<float*>(reinterpret_cast<char*>(nb_grad[j]) + (size_t)(unsigned)(quad_bcast<SS>(ro[t]) + g8)); \
                _Pragma("unroll") for (int i = 0; i < 2; ++i) {                                                       \
                    const float gw = go[SS][i] * (two_r * wv[j][SS][i] - two_r2 * S_[SS][i]);                         \
                    if (tw != 0.0f) atomicAdd(cell + i, gw * tw);                                                     \
                }                                                                                                     \
            }                                                                                                         \
        }
#define MVS_GRAD_OF(QQ)                                                                                               \
        if constexpr (2 * p + QQ < K) {                                                                               \
            constexpr int j = 2 * p + QQ;                                                                             \
            const unsigned fj = fl >> (4 * j);                                                                        \
            if (fj & kFlagLive) {                                                                                     \
                int ro[4], rw[4];                                                                                     \
                _Pragma("unroll") for (int t = 0; t < 4; ++t) {                                                       \
                    ro[t] = from_quad<QQ>(dox[p][t]);                                                                 \
                    rw[t] = from_quad<QQ>(__float_as_int(dwx[p][t]));                                                 \
                }                                                                                                     \
                if (fj & kFlagStaged) {                                                                               \
                    MVS_GRAD_STEP_LDS(0) MVS_GRAD_STEP_LDS(1) MVS_GRAD_STEP_LDS(2) MVS_GRAD_STEP_LDS(3)               \
                } else {                                                                                              \
                    MVS_GRAD_STEP_GLB(0) MVS_GRAD_STEP_GLB(1) MVS_GRAD_STEP_GLB(2) MVS_GRAD_STEP_GLB(3)               \
                }                                                                                                     \
            }                                                                                                         \
        }
        auto grads = [&](auto pc) {
            constexpr int p = decltype(pc)::value;
            MVS_GRAD_OF(0)
            MVS_GRAD_OF(1)
        };
        if constexpr (NP > 0) grads(std::integral_constant<int, 0>{});
        if constexpr (NP > 1) grads(std::integral_constant<int, 1>{});
#undef MVS_GRAD_OF
#undef MVS_GRAD_STEP_LDS
#undef MVS_GRAD_STEP_GLB
    }
    // ---- the boxes still resident, then the reference term: once per block
    __syncthreads();
#pragma unroll
    for (int j = 0; j < K; ++j)
        if (have[j]) flush_box(j);
    float* gr = gpacked + ((size_t)n * S + slab) * slab_stride + kHalfSlab * half;
#pragma unroll
    for (int s = 0; s < 4; ++s)
        if (pok[s]) {
            const int pix = py * W + px0 + s;
#pragma unroll
            for (int i = 0; i < 2; ++i) atomicAdd(gr + (size_t)pix * kSlab + 2 * g + i, gref[s][i]);
        }
}


}  // namespace mvsdet

using namespace mvsdet;

// defined in planesweep.hip
extern "C" size_t mvsdet_plane_sweep_scratch_bytes(int N, int K, int D, int H, int W);
extern "C" int mvsdet_plane_sweep_table_f32(const float* proj, const float* depth, void* scratch, size_t scratch_bytes,
                                            int N, int K, int D, int H, int W, mvsdet_stream_t stream);
namespace mvsdet {
int sweep_tile_width(int W, int D);  // the tile shape the sweep geometry is built for
int sweep_box_cap(int K, int tw);
int sweep_max_box_cap(int K, int tw);
}

extern "C" size_t mvsdet_plane_sweep_bwd_workspace_bytes(int N, int K, int C, int D, int H, int W) {
    const size_t pb = (mvsdet_packed_bytes(N, C, H, W) + 255) / 256 * 256;
    return 2 * pb + mvsdet_plane_sweep_scratch_bytes(N, K, D, H, W);
}

// the kernels behind both entry points: packed maps and a sweep geometry in, the gradient of the features out
static int bwd_launch(const float* packed, const int64_t* nbr, void* scratch, const float* g, float* gfeat, float* gpacked, size_t pb,
                      int N, int K, int C, int D, int H, int W, hipStream_t stream);

extern "C" int mvsdet_plane_sweep_variance_bwd_f32(const float* feat, const int64_t* nbr, const float* proj,
                                                   const float* depth, const float* g, float* gfeat, void* workspace,
                                                   size_t workspace_bytes, int N, int K, int C, int D, int H, int W,
                                                   mvsdet_stream_t stream_) {
    hipStream_t stream = (hipStream_t)stream_;
    MVS_REQUIRE(feat && depth && g && gfeat && workspace, "plane_sweep_variance_bwd: NULL pointer");
    MVS_REQUIRE(K == 0 || (nbr && proj), "plane_sweep_variance_bwd: NULL neighbour arrays with K=%d", K);
    MVS_REQUIRE(N > 0 && C > 0 && D > 0 && H > 1 && W > 1, "plane_sweep_variance_bwd: bad shape");
    MVS_REQUIRE(K >= 0 && K <= MVSDET_MAX_NEIGHBORS, "plane_sweep_variance_bwd: K=%d outside [0,%d]", K, MVSDET_MAX_NEIGHBORS);
    MVS_REQUIRE(D <= MVSDET_MAX_DEPTH && H < 65535 && W < 65535, "plane_sweep_variance_bwd: D > %d, or H or W > 65534", MVSDET_MAX_DEPTH);
    MVS_REQUIRE((size_t)H * W * kSlab < (size_t)INT32_MAX, "plane_sweep_variance_bwd: one slab image exceeds 2^31 elements");
    const size_t pb = (mvsdet_packed_bytes(N, C, H, W) + 255) / 256 * 256;
    const size_t sb = mvsdet_plane_sweep_scratch_bytes(N, K, D, H, W);
    if (workspace_bytes < 2 * pb + sb) {
        set_error("plane_sweep_variance_bwd: workspace %zu B < %zu B", workspace_bytes, 2 * pb + sb);
        return MVSDET_ERR_WORKSPACE;
    }
    float* packed = (float*)workspace;
    float* gpacked = (float*)((char*)workspace + pb);
    void* scratch = (char*)workspace + 2 * pb;
    const int64_t fs[4] = {(int64_t)C * H * W, (int64_t)H * W, W, 1};
    if (int rc = mvsdet_pack_features_f32(feat, fs, packed, N, C, H, W, stream_)) return rc;
    if (K > 0)
        if (int rc = mvsdet_plane_sweep_table_f32(proj, depth, scratch, sb, N, K, D, H, W, stream_)) return rc;
    return bwd_launch(packed, nbr, scratch, g, gfeat, gpacked, pb, N, K, C, D, H, W, stream);
}

// The same with what the FORWARD pass already made: the packed maps (mvsdet_pack_features_f32) and the sweep geometry the forward
// call left in its scratch buffer (mvsdet_plane_sweep_variance_packed_f32 / mvsdet_plane_sweep_table_f32 of the same N, K, D, H,
// W and tile options: contiguous, not pitched).  Saves a packing pass and a geometry kernel per training step (0.18 of 2.4 ms at
// the reference-true shape).  workspace: mvsdet_packed_bytes rounded up to 256 (the packed gradient map).
extern "C" int mvsdet_plane_sweep_variance_bwd_packed_f32(const float* packed, const int64_t* nbr, const void* table,
                                                          size_t table_bytes, const float* g, float* gfeat, void* workspace,
                                                          size_t workspace_bytes, int N, int K, int C, int D, int H, int W,
                                                          mvsdet_stream_t stream_) {
    MVS_REQUIRE(packed && g && gfeat && workspace, "plane_sweep_variance_bwd_packed: NULL pointer");
    MVS_REQUIRE(K == 0 || (nbr && table), "plane_sweep_variance_bwd_packed: NULL neighbour ids or geometry with K=%d", K);
    MVS_REQUIRE(N > 0 && C > 0 && D > 0 && H > 1 && W > 1, "plane_sweep_variance_bwd_packed: bad shape");
    MVS_REQUIRE(K >= 0 && K <= MVSDET_MAX_NEIGHBORS, "plane_sweep_variance_bwd_packed: K=%d outside [0,%d]", K, MVSDET_MAX_NEIGHBORS);
    MVS_REQUIRE(D <= MVSDET_MAX_DEPTH && H < 65535 && W < 65535, "plane_sweep_variance_bwd_packed: D > %d, or H or W > 65534", MVSDET_MAX_DEPTH);
    MVS_REQUIRE((size_t)H * W * kSlab < (size_t)INT32_MAX, "plane_sweep_variance_bwd_packed: one slab image exceeds 2^31 elements");
    const size_t pb = (mvsdet_packed_bytes(N, C, H, W) + 255) / 256 * 256;
    if (workspace_bytes < pb) {
        set_error("plane_sweep_variance_bwd_packed: workspace %zu B < %zu B", workspace_bytes, pb);
        return MVSDET_ERR_WORKSPACE;
    }
    MVS_REQUIRE(K == 0 || table_bytes >= mvsdet_plane_sweep_scratch_bytes(N, K, D, H, W), "plane_sweep_variance_bwd_packed: geometry buffer too small");
    return bwd_launch(packed, nbr, const_cast<void*>(table), g, gfeat, (float*)workspace, pb, N, K, C, D, H, W, (hipStream_t)stream_);
}

static int bwd_launch(const float* packed, const int64_t* nbr, void* scratch, const float* g, float* gfeat, float* gpacked, size_t pb,
                      int N, int K, int C, int D, int H, int W, hipStream_t stream) {
    if (hipMemsetAsync(gpacked, 0, pb, stream) != hipSuccess) {
        set_error("plane_sweep_variance_bwd: hipMemsetAsync failed");
        return MVSDET_ERR_HIP;
    }
    const int tw = sweep_tile_width(W, D);
    const int th = kTilePix / tw;
    const int S = num_slabs(C);
    const int HW = H * W;
    const int tiles_x = (W + tw - 1) / tw, tiles = tiles_x * ((H + th - 1) / th);
    const long long nblocks = (long long)N * tiles * S;
    MVS_REQUIRE(nblocks <= INT32_MAX / 2 - 16, "plane_sweep_variance_bwd: grid too large");
    const SweepGeometry geo = sweep_geometry(scratch, N, K, D, tiles);
    const int box_cap = sweep_max_box_cap(K, tw);   // the geometry may be the forward pass's: slots for the largest capacity it can have been built with
    const size_t lds = sweep_lds_bytes(K, box_cap);
    dim3 grid((unsigned)nblocks);
    // plane groups of a block (header comment): option "bwd_groups" 0 = by the plane count, 1 / 2 force
    const int groups = options().bwd_groups ? std::min(2, std::max(1, options().bwd_groups)) : (D >= 32 ? 2 : 1);
    // two instances: the lower and the upper 16 packed floats of every texel = channels 8i + {0..3} and 8i + {4..7} of the
    // slab (pack.h), so the upper one has nothing to do only when C <= 4
#define MVS_BWD_LAUNCH1(KV, TWV, HV, GV)                                                                                 \
    {                                                                                                                  \
        auto* k = plane_sweep_variance_bwd_kernel<KV, TWV, HV, GV>;                                                    \
        if (lds > 48 * 1024 && hipFuncSetAttribute(reinterpret_cast<const void*>(k), hipFuncAttributeMaxDynamicSharedMemorySize, \
                                                   80 * 1024) != hipSuccess) {                                         \
            set_error("plane_sweep_variance_bwd: hipFuncSetAttribute(MaxDynamicSharedMemorySize) failed");             \
            return MVSDET_ERR_HIP;                                                                                     \
        }                                                                                                              \
        hipLaunchKernelGGL(k, grid, dim3(kThreads * GV), lds, stream, packed, nbr, geo.header, geo.proj, geo.depth, geo.boxes, geo.flags, g, \
                           gpacked, N, C, S, D, H, W, tiles_x, tiles, box_cap);                                        \
    }
#define MVS_BWD_LAUNCH(KV, TWV)                        \
    {                                                  \
        if (C > 4) {                                   \
            grid = dim3((unsigned)((nblocks + 7) / 8 * 16)); \
            if (groups == 2) MVS_BWD_LAUNCH1(KV, TWV, 2, 2) \
            else MVS_BWD_LAUNCH1(KV, TWV, 2, 1)        \
        } else                                         \
            MVS_BWD_LAUNCH1(KV, TWV, 0, 1)             \
    }
#define MVS_BWD_CASE(KV)                     \
    case KV:                                 \
        if (tw == 16) MVS_BWD_LAUNCH(KV, 16) \
        else MVS_BWD_LAUNCH(KV, 32)          \
        break;
    switch (K) {
        MVS_BWD_CASE(0)
        MVS_BWD_CASE(1)
        MVS_BWD_CASE(2)
        MVS_BWD_CASE(3)
        MVS_BWD_CASE(4)
    }
#undef MVS_BWD_CASE
#undef MVS_BWD_LAUNCH
#undef MVS_BWD_LAUNCH1
    MVS_LAUNCH_CHECK("plane_sweep_variance_bwd");
    if (HW % 4 == 0 && (uintptr_t)gfeat % 16 == 0) {
        dim3 dgrid((HW + 127) / 128, S, N);
        hipLaunchKernelGGL(unpack_features_dense_kernel, dgrid, dim3(kThreads), 0, stream, gpacked, gfeat, C, S, HW);
    } else {
        dim3 ugrid((HW + 63) / 64, S, N);
        hipLaunchKernelGGL(unpack_features_kernel, ugrid, dim3(kThreads), 0, stream, gpacked, gfeat, C, S, H, W);
    }
    MVS_LAUNCH_CHECK("unpack_features");
    return MVSDET_OK;
}
