// conv0 of the cost regularisation network (mvs_models/mvsnet.py:76: Conv3d(256 -> 64, 3, padding 1) on the fp32 variance volume) with
// ONE fp16 product and ONE block-scaled FP6 product per fp32-equivalent product instead of three bf16 products:
//
//     x = xh + xr,  w = wh + wr          xh = fp16(x) (11 significant bits), xr = x - xh (exact in fp32); the same for w
//     x * w ~= xh * wh                   v_mfma_f32_16x16x32_f16: the product of two 11-bit numbers is exact in fp32
//            + Q(xh) * Q(wr) + Q(xr) * Q(wh)     the two correction terms (2^-11 of the result: 4-5 bits of them are enough) as ONE
//                                        v_mfma_scale_f32_16x16x128_f8f6f4 on e2m3 operands (OCP MX FP6: 4 significant bits, one e8m0
//                                        scale per 32 elements) whose k runs over (channel, term) pairs
//
// Gate and price (profiles/r06_mixed_format_gate_emulation.txt, r06_mx_mix.txt, r06_conv0_terms_whatif.txt): whole-network logits
// 1-2e-5 from float64 (bf16x3: 2-4e-6; the bar is 1e-4); an e2m3 MX instruction of K = 128 takes 1.16 x the time of ONE 16x16x32, so a
// group of 8 channels costs 7 + 4 x 1.16 = 11.6 units against 21.
//
// Included by costreg_bf16.hip (it shares that file's BfOut, block map and epilogue conventions).
//
// Data flow = conv3d_k3_bf16x3_kernel's 16x16x32 form with the fp32 input cut in the kernel:
//   k-step  = 4 taps x 8 channels (lane group kg = lane >> 4 takes tap 4 ks + kg); 27 taps = 7 k-steps, the 28th slot empty
//   stage   = 8 channels: halo tile in LDS as  hi[slot] = 8 fp16 (16 B)  and  q[slot] = 16 e2m3 codes (12 B in a 16-B slot):
//             element 2c = Q(xh_c * 2^-E1), element 2c + 1 = Q(xr_c * 2^-(E1 - 11)) -- what ONE v_cvt_scalef32_2xpk16_fp6_f32 makes of
//             a thread's two voxels (it interleaves its two 16-float groups element by element: tools/micro/cvt_fp6_probe.hip)
//   scale   : a lane's 32 elements of an MX operand are 2 taps x 8 channels x 2 terms, i.e. TWO different voxels -- so the activations'
//             scale must be one per (block, stage): E1 = floor(log2 max|x|) - 2 over the halo tile's 8 channels (every value below 8
//             after scaling; e2m3's largest is 7.5), found by the staging threads (wave maximum -> LDS -> barrier) before they
//             quantise; xr rides on E1 - 11 (|xr| <= 2^-11 * 2^floor(log2|x|)).  Weights: per (output channel, 32 k) block,
//             Ew = floor(log2 max|wh|) - 2, wr on Ew - 11: both kinds of product then carry the SAME power of two,
//             2^(E1 + Ew - 11), which is what lets them share an instruction (scale bytes E1 + 127 and Ew - 11 + 127).
//   MX group: k-steps 2g, 2g+1 (g = 0..3; k-step 7 is empty: zero weights); the lane's 32 elements are e = 16 i + 2 c + term for
//             k-step 2g + i: its B fragment is the two 12-byte units it reads at its two taps -- no bit assembly
//   weights : [Cout/64][Cin/8][sub-stage 2][slot 32][lane 64] x 16 B (mvsdet_split_conv_weight_mx):
//             slots 0..15   hi of k-step 4s + (slot >> 2), row group slot & 3: 8 fp16 of the lane's (row m = lane & 15, tap 4 ks + kg)
//             slots 16..31  MX group 2s + gl: slot = 16 + 2 (4 gl + rg) + half; element 16 i + 2 c = Q(wr), + 1 = Q(wh) (wr meets
//                           Q(xh), wh meets Q(xr)); half 0 = fragment words 0..3, half 1 = words 4, 5, the e8m0 scale byte, padding
//   fp16    : fp16 ends at 65504.  A stage whose largest magnitude reaches 2^15 is cut as x * 2^-S (S = floor(log2 max|x|) - 14, a
//             block-uniform power of two; S = 0 for any input below 32768, i.e. always for a variance of backbone features); the
//             accumulators carry the power of two of the stage they were last added to and are re-scaled when it changes.
// Values: every product term is exact in fp32 and accumulated in fp32, like the bf16x3 kernel's; what is left out is the
// quantisation of the two correction terms (2^-4 relative of a 2^-11 term each).

typedef _Float16 f16x8 __attribute__((ext_vector_type(8)));
typedef int mx_i32x8 __attribute__((ext_vector_type(8)));
typedef float mx_f32x4 __attribute__((ext_vector_type(4)));

constexpr int kMxSlots = 32;   // 16-byte slots per lane and weight sub-stage
// Nothing moves across: neither the optimiser's code motion (it hoists every LDS read of a sub-stage above the staging code and
// sinks the MFMAs below it: all fragments live at once, the accumulators spilled) nor the machine scheduler.
#define MX_FENCE()                                  \
    do {                                            \
        asm volatile("" ::: "memory");              \
        __builtin_amdgcn_sched_barrier(0);          \
    } while (0)

// e2m3 code (sign bit 5, exponent bits 4:3 with bias 1, mantissa bits 2:0; subnormals m / 8 below 1.0; largest 7.5) of v * 2^-e, round
// to nearest even, saturating.  The same arithmetic as oracle / tools/study/mixed_format_gate.py (q_e2m3).
__host__ __device__ __forceinline__ unsigned mx_e2m3_code(float v, int e) {
    const float y = ldexpf(v, -e);                      // exact: a power of two
    float a = fabsf(y);
    a = a > 7.5f ? 7.5f : a;
    const float t = a < 1.0f ? 1.0f : a;
    const int ee = (int)((__builtin_bit_cast(unsigned, t) >> 23) & 255u) - 127;             // 0, 1, 2
    const float mult = __builtin_bit_cast(float, (unsigned)(130 - ee) << 23);              // 2^(3 - ee)
    const int r = (int)rintf(a * mult);                                                     // 0 .. 16, round to nearest even
    unsigned code = (unsigned)((ee << 3) + r);
    code = code > 31u ? 31u : code;
    return code | (y < 0.0f ? 32u : 0u);
}
// scale exponent of a block whose largest magnitude is amax: everything below 8 after the scaling (0 -> a harmless exponent)
__host__ __device__ __forceinline__ int mx_block_exp(float amax) {
    if (!(amax > 0.0f)) return -100;
    const int e = (int)((__builtin_bit_cast(unsigned, amax) >> 23) & 255u) - 127 - 2;       // floor(log2 amax) - 2 (subnormal amax: -129)
    return e < -116 ? -116 : (e > 120 ? 120 : e);
}
// weight (Cout,Cin,27) fp32 -> the layout above; thread = one (block of 64 outputs, channel group, sub-stage, slot, lane) unit
__global__ __launch_bounds__(kThreads) void split_conv_weight_mx_kernel(const float* __restrict__ w, uint4* __restrict__ out, int Cin, int C8,
                                                                        size_t units) {
    const size_t u = (size_t)blockIdx.x * kThreads + threadIdx.x;
    if (u >= units) return;
    const int lane = (int)(u & 63), slot = (int)((u >> 6) & 31), sub = (int)((u >> 11) & 1);
    const size_t r = u >> 12;
    const int c8 = (int)(r % C8), ob = (int)(r / C8);
    const int m = lane & 15, kg = lane >> 4;
    auto weight_at = [&](int rg, int ks, int c) -> float {       // (row group, k-step 0..7, channel of the group)
        const int o = ob * 64 + 32 * (rg >> 1) + 8 * (m >> 2) + 4 * (rg & 1) + (m & 3), t = 4 * ks + kg, ch = c8 * 8 + c;
        return (t <= 26 && ch < Cin) ? w[((size_t)o * Cin + ch) * 27 + t] : 0.0f;
    };
    uint4 res = make_uint4(0u, 0u, 0u, 0u);
    if (slot < 16) {
        const int ks = 4 * sub + (slot >> 2), rg = slot & 3;
        if (ks < 7) {
            unsigned short h[8];
#pragma unroll
            for (int c = 0; c < 8; ++c) h[c] = __builtin_bit_cast(unsigned short, (_Float16)weight_at(rg, ks, c));
            res = make_uint4((unsigned)h[0] | ((unsigned)h[1] << 16), (unsigned)h[2] | ((unsigned)h[3] << 16),
                             (unsigned)h[4] | ((unsigned)h[5] << 16), (unsigned)h[6] | ((unsigned)h[7] << 16));
        }
    } else {
        const int idx = (slot - 16) >> 1, half = (slot - 16) & 1, gl = idx >> 2, rg = idx & 3;
        const int ks0 = 4 * sub + 2 * gl;                      // the group's two k-steps: ks0, ks0 + 1
        float wh[16], wr[16];
        float amax = 0.0f;
#pragma unroll
        for (int i = 0; i < 2; ++i)
#pragma unroll
            for (int c = 0; c < 8; ++c) {
                const float f = weight_at(rg, ks0 + i, c);
                wh[8 * i + c] = (float)(_Float16)f;
                wr[8 * i + c] = f - wh[8 * i + c];
                amax = fmaxf(amax, fabsf(wh[8 * i + c]));
            }
        const int e = mx_block_exp(amax);                      // wh on 2^e, wr on 2^(e - 11)
        unsigned words[6] = {0u, 0u, 0u, 0u, 0u, 0u};
#pragma unroll
        for (int el = 0; el < 32; ++el) {                      // element 16 i + 2 c = Q(wr), + 1 = Q(wh)
            const int i = el >> 4, c = (el >> 1) & 7, term = el & 1;
            const unsigned code = term ? mx_e2m3_code(wh[8 * i + c], e) : mx_e2m3_code(wr[8 * i + c], e - 11);
            const int bit = 6 * el;
            words[bit >> 5] |= code << (bit & 31);
            if ((bit & 31) > 26) words[(bit >> 5) + 1] |= code >> (32 - (bit & 31));
        }
        res = half == 0 ? make_uint4(words[0], words[1], words[2], words[3]) : make_uint4(words[4], words[5], (unsigned)(e - 11 + 127), 0u);
    }
    out[u] = res;
}

typedef float mx_f32x16 __attribute__((ext_vector_type(16)));
typedef unsigned mx_u32x6 __attribute__((ext_vector_type(6)));
typedef unsigned mx_u32x4 __attribute__((ext_vector_type(4)));
typedef int mx_i32x4 __attribute__((ext_vector_type(4)));

template <int TD, int TH>
__global__ __launch_bounds__(TD * TH * 16) void conv3d_k3_fp16mx_kernel(
    const float* __restrict__ xf, long long sN, long long sC, long long sD, long long sH, int Cin, const uint4* __restrict__ wq,
    const float* __restrict__ scale, const float* __restrict__ shift, BfOut dst, int C8, int Cout, int D, int H, int W, int tiles_w,
    int relu, int xcd_map) {
    constexpr int TW = 16, CGN = 4;
    constexpr int NW = TD * TH * TW / 64, NT = 64 * NW;
    constexpr bool kTight = NW > 8;                        // three waves per SIMD: 168 registers per lane
    constexpr int HD = TD + 2, HH = TH + 2, HW = TW + 2;
    constexpr int NVOX = HD * HH * HW;
    constexpr int INS = bf_in_slots(TD, TH, TW);
    static_assert(INS - NVOX >= 4, "the stage's spare slots hold the waves' maxima");
    static_assert(NW <= 14, "one float per wave in three spare slots");
    constexpr int kWSlots = kMxSlots * 64;                 // 16-byte slots of one weight buffer
    constexpr int NV = (NVOX + NT - 1) / NT;               // halo voxels per thread
    constexpr int NVP = (NV + 1) / 2;                      // ... in pairs: one conversion instruction per pair
    extern __shared__ uint4 s_mx[];   // [2 stages][hi INS | q INS], then [2][kWSlots] weights
    constexpr int STAGE = 2 * INS;    // uint4 per input stage
    uint4* s_in = s_mx;
    uint4* s_w = s_mx + 2 * STAGE;
    float* __restrict__ out = dst.f32;

    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    unsigned bx, by, bz;
    xcd_block_id(xcd_map, bx, by, bz);
    const int bw = bx % tiles_w, bh = bx / tiles_w;
    const int nob = Cout / 64;
    const int n = bz / nob, ob64 = bz % nob;
    const int w0 = bw * TW, h0 = bh * TH, d0 = by * TD;

    // ---- the fp32 input by BUFFER loads: one descriptor for the view's Cin channel volumes; a lane's byte offset inside a channel
    // volume, or an offset beyond the descriptor's range for a halo voxel outside the volume -- the hardware then returns 0: no
    // branch and no 64-bit address per load
    unsigned f_off[NV];
    float f_reg[NV][8];
    const float* xfn = xf + (size_t)n * sN;
    const unsigned long long span = ((unsigned long long)(Cin - 1) * (unsigned long long)sC + (unsigned long long)(D - 1) * sD +
                                     (unsigned long long)(H - 1) * sH + W) * 4ull;   // bytes of the view this kernel may touch (< 2^32: host)
    const __amdgpu_buffer_rsrc_t rsrc = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(xfn), 0, (int)(unsigned)span, 0x00020000);
#pragma unroll
    for (int k = 0; k < NV; ++k) {
        const int slot = tid + k * NT;
        const int dz = slot / (HH * HW), r = slot - dz * (HH * HW), hy = r / HW, wx = r - hy * HW;
        const int d = d0 + dz - 1, h = h0 + hy - 1, w = w0 + wx - 1;
        const bool ok = slot < NVOX && d >= 0 && d < D && h >= 0 && h < H && w >= 0 && w < W;
        f_off[k] = ok ? (unsigned)(((long long)d * sD + (long long)h * sH + w) * 4) : 0xfffffff0u;
    }
    const unsigned sC4 = (unsigned)(sC * 4);
    auto fetch_f32 = [&](int c8) {
#pragma unroll
        for (int j = 0; j < 8; ++j) {
            const int c = c8 * 8 + j;
            const unsigned soff = (c < Cin ? (unsigned)c : (unsigned)Cin) * sC4;   // a channel beyond Cin: out of range -> 0
#pragma unroll
            for (int k = 0; k < NV; ++k)
                f_reg[k][j] = __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(rsrc, f_off[k], (int)soff, 0));
        }
    };
    // the wave's largest magnitude of the fetched values -> the stage buffer's spare slots
    auto publish_amax = [&](int buf) {
        float a = 0.0f;
#pragma unroll
        for (int k = 0; k < NV; ++k)
#pragma unroll
            for (int j = 0; j < 8; ++j) a = fmaxf(a, fabsf(f_reg[k][j]));     // (NaN inputs: fmaxf drops them; the products carry them)
#pragma unroll
        for (int off = 32; off > 0; off >>= 1) a = fmaxf(a, __shfl_xor(a, off, 64));
        float* spare = reinterpret_cast<float*>(s_in + (size_t)buf * STAGE + NVOX);
        if (lane == 0) spare[wave] = a;
    };
    auto stage_exp = [&](int buf) -> int {
        const float* spare = reinterpret_cast<const float*>(s_in + (size_t)buf * STAGE + NVOX);
        float a = 0.0f;
#pragma unroll
        for (int i = 0; i < NW; ++i) a = fmaxf(a, spare[i]);
        return __builtin_amdgcn_readfirstlane(mx_block_exp(a));
    };
    auto stage_shift = [](int e1) -> int { return e1 > 12 ? e1 - 12 : 0; };   // S: max|x| * 2^-S < 2^15
    // registers -> LDS: hi (8 fp16) and the 16 e2m3 codes of (xh, xr * 2^11) on the stage's scale: ONE conversion per two voxels
    auto stage_cut = [&](int buf, int e1) {
        uint4* hi_s = s_in + (size_t)buf * STAGE;
        uint4* q_s = hi_s + INS;
        const int S = stage_shift(e1);
        const float down = __builtin_bit_cast(float, (unsigned)(127 - S) << 23);        // 2^-S (1.0 unless the stage is huge)
        const float sdiv = __builtin_bit_cast(float, (unsigned)(e1 - S + 127) << 23);   // 2^(e1 - S) (the instruction divides by it)
#pragma unroll
        for (int kp = 0; kp < NVP; ++kp) {
            mx_f32x16 ga, gb;
            unsigned hw[2][4];
#pragma unroll
            for (int v = 0; v < 2; ++v) {
                const int k = 2 * kp + v;
#pragma unroll
                for (int j = 0; j < 8; ++j) {
                    const float f = (k < NV ? f_reg[k < NV ? k : 0][j] : 0.0f) * down;
                    const _Float16 h = (_Float16)f;
                    const float hf = (float)h;
                    const unsigned hb = __builtin_bit_cast(unsigned short, h);
                    if (j & 1) hw[v][j >> 1] |= hb << 16; else hw[v][j >> 1] = hb;
                    ga[8 * v + j] = hf;
                    gb[8 * v + j] = (f - hf) * 2048.0f;          // exact: on the scale 2^(e1 - 11)
                }
            }
            // (wait states between the vector instructions that write the 32 source registers and the conversion that reads them as two
            // 16-register tuples: without them the FIRST conversion of a thread read stale registers in the wave-specialised kernel --
            // a hazard the compiler does not cover for this instruction)
            asm volatile("s_nop 7\n\ts_nop 7" : "+v"(ga), "+v"(gb));
            const mx_u32x6 q = __builtin_amdgcn_cvt_scalef32_2xpk16_fp6_f32(ga, gb, sdiv);   // element 2 i = ga[i], 2 i + 1 = gb[i]
#pragma unroll
            for (int v = 0; v < 2; ++v) {
                const int k = 2 * kp + v, slot = tid + k * NT;
                if (k < NV && slot < NVOX) {
                    hi_s[slot] = make_uint4(hw[v][0], hw[v][1], hw[v][2], hw[v][3]);
                    q_s[slot] = make_uint4(q[3 * v], q[3 * v + 1], q[3 * v + 2], 0u);
                }
            }
        }
    };
    const uint4* wn = wq + (size_t)ob64 * C8 * (2 * kWSlots);
    auto dma_weights = [&](int c8, int s, int buf) {
        const uint4* src0 = wn + ((size_t)c8 * 2 + s) * kWSlots + lane;
        for (int i = wave; i < kMxSlots; i += NW) {
            uint4* dstp = s_w + (size_t)buf * kWSlots + i * 64;
            __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)(src0 + i * 64),
                                             (__attribute__((address_space(3))) void*)dstp, 16, 0, 0);
        }
    };

    // ---- the wave's four column groups of 16 voxels (as the 16x16x32 form of conv3d_k3_bf16x3_kernel)
    constexpr int RG16 = 16 / TW > 0 ? 16 / TW : 1;
    const int col16 = lane & 15, kg = lane >> 4;
    int vb16[CGN];
#pragma unroll
    for (int cg = 0; cg < CGN; ++cg) {
        const int g = CGN * wave + cg;
        const int dz = g / (TH / RG16), hy = RG16 * (g % (TH / RG16)) + col16 / TW;
        vb16[cg] = (dz * HH + hy) * HW + col16 % TW;
    }
    // tap offsets: lane group kg reads tap 4 ks + kg of k-step ks (k-step 7 -- taps 28..31 -- is empty: zero weights, any voxel of the
    // tile).  Eight registers per lane; with 168 of them (12 waves) the table sits in ONE register instead -- lane i holds the offset
    // of tap i -- and a lane fetches its entry with ds_bpermute when a k-step needs it.
    int toffs[kTight ? 1 : 8];
    int ttab = 0;
    if constexpr (kTight) {
        int t = lane & 31;
        t = t > 26 ? 26 : t;
        ttab = ((t / 9) * HH + (t / 3) % 3) * HW + t % 3;
    } else {
#pragma unroll
        for (int ks = 0; ks < 8; ++ks) {
            const int k7 = ks < 7 ? ks : 6;
            toffs[ks] = kg == 0 ? bf_tap_off<HH, HW>(4 * k7) : kg == 1 ? bf_tap_off<HH, HW>(4 * k7 + 1)
                      : kg == 2 ? bf_tap_off<HH, HW>(4 * k7 + 2) : bf_tap_off<HH, HW>(4 * k7 + 3);
        }
    }
    auto tap_off = [&](int ks) -> int {
        if constexpr (kTight) return __builtin_amdgcn_ds_bpermute((4 * ks + kg) << 2, ttab);
        else return toffs[ks];
    };
    mx_f32x4 acc[4][CGN];
#pragma unroll
    for (int a = 0; a < 4; ++a)
#pragma unroll
        for (int b = 0; b < CGN; ++b) acc[a][b] = (mx_f32x4){0.0f, 0.0f, 0.0f, 0.0f};

    auto compute = [&](auto sc, int ibuf, int wbuf, int e1) {
        constexpr int s = decltype(sc)::value;
        constexpr int nks = s == 0 ? 4 : 3;
        const uint4* hi_s = s_in + (size_t)ibuf * STAGE;
        const uint4* q_s = hi_s + INS;
        const uint4* ain = s_w + (size_t)wbuf * kWSlots + lane;
        const int sb = e1 - stage_shift(e1) + 127;
        // the fp16 products, k-step by k-step
#ifndef MXX_NO_F16
#pragma unroll
        for (int kl = 0; kl < nks; ++kl) {
            const int toff = tap_off(4 * s + kl);
            f16x8 A[4], B[CGN];
#pragma unroll
            for (int a = 0; a < 4; ++a) A[a] = __builtin_bit_cast(f16x8, ain[(kl * 4 + a) * 64]);
#pragma unroll
            for (int b = 0; b < CGN; ++b) B[b] = __builtin_bit_cast(f16x8, hi_s[vb16[b] + toff]);
#pragma unroll
            for (int a = 0; a < 4; ++a)
#pragma unroll
                for (int b = 0; b < CGN; ++b) acc[a][b] = __builtin_amdgcn_mfma_f32_16x16x32_f16(A[a], B[b], acc[a][b], 0, 0, 0);
            if constexpr (kTight) MX_FENCE();   // 12 waves: 168 registers -- the next k-step's fragments are not fetched ahead
        }
#endif
        // the correction terms of the sub-stage's two pairs of k-steps
#ifndef MXX_NO_MX
#pragma unroll
        for (int gl = 0; gl < 2; ++gl) {
            mx_i32x8 FB[CGN];
            const int t0 = tap_off(4 * s + 2 * gl), t1 = tap_off(4 * s + 2 * gl + 1);
#pragma unroll
            for (int b = 0; b < CGN; ++b) {
                const uint4 u0 = q_s[vb16[b] + t0], u1 = q_s[vb16[b] + t1];
                FB[b] = (mx_i32x8){(int)u0.x, (int)u0.y, (int)u0.z, (int)u1.x, (int)u1.y, (int)u1.z, 0, 0};
            }
#pragma unroll
            for (int a = 0; a < 4; ++a) {
                const uint4 lo = ain[(16 + 2 * (4 * gl + a)) * 64], hi = ain[(16 + 2 * (4 * gl + a) + 1) * 64];
                const mx_i32x8 FA = {(int)lo.x, (int)lo.y, (int)lo.z, (int)lo.w, (int)hi.x, (int)hi.y, 0, 0};
                const int sa = (int)hi.z;
#pragma unroll
                for (int b = 0; b < CGN; ++b)
                    acc[a][b] = __builtin_amdgcn_mfma_scale_f32_16x16x128_f8f6f4(FA, FB[b], acc[a][b], 2, 2, 0, sa, 0, sb);
            }
            if constexpr (kTight) {
                asm volatile("" : "+v"(acc[0][0]), "+v"(acc[0][1]), "+v"(acc[0][2]), "+v"(acc[0][3]), "+v"(acc[1][0]), "+v"(acc[1][1]),
                                  "+v"(acc[1][2]), "+v"(acc[1][3]), "+v"(acc[2][0]), "+v"(acc[2][1]), "+v"(acc[2][2]), "+v"(acc[2][3]),
                                  "+v"(acc[3][0]), "+v"(acc[3][1]), "+v"(acc[3][2]), "+v"(acc[3][3]));
                MX_FENCE();
            }
        }
        // pinned HERE, once per sub-stage: the instruction is pure, and the compiler otherwise sinks the products below the staging
        // code that follows them (their 12-register operands then stay live across that code: the accumulators go to scratch).  One
        // statement for all sixteen accumulators: a statement per instruction makes each wait for its own result.
        asm volatile("" : "+v"(acc[0][0]), "+v"(acc[0][1]), "+v"(acc[0][2]), "+v"(acc[0][3]), "+v"(acc[1][0]), "+v"(acc[1][1]),
                          "+v"(acc[1][2]), "+v"(acc[1][3]), "+v"(acc[2][0]), "+v"(acc[2][1]), "+v"(acc[2][2]), "+v"(acc[2][3]),
                          "+v"(acc[3][0]), "+v"(acc[3][1]), "+v"(acc[3][2]), "+v"(acc[3][3]));
#endif
    };

    // ---- pipeline: per channel group two sub-stages (k-steps 0..3 + MX groups 0, 1; k-steps 4..6 + MX groups 2, 3).  The fp32 values of
    // stage c + 2 are requested as soon as stage c + 1 has been cut (the registers are free again) and are waited for a whole stage
    // later: with one block of 8-12 waves per CU, all at the same point of the same stage, nothing else would cover that latency.
    int e_cur = -100, s_acc = 0;      // s_acc: the accumulators hold sums of (x * 2^-s_acc) * w
    constexpr int kFetchLoads = NV * 8;
    static_assert(kFetchLoads <= 48, "the counted wait below");
    if (C8 > 0) {
        fetch_f32(0);
        dma_weights(0, 0, 0);
        publish_amax(0);
        __builtin_amdgcn_s_waitcnt(0xc07f);   // lgkmcnt(0): the wave's maximum is in LDS
        __builtin_amdgcn_s_barrier();
        e_cur = stage_exp(0);
        stage_cut(0, e_cur);
        s_acc = stage_shift(e_cur);
        if (C8 > 1) fetch_f32(1);
        __builtin_amdgcn_s_waitcnt(0xc07f);
    }
    for (int c8 = 0; c8 < C8; ++c8) {
        const int ibuf = c8 & 1;
        const bool more = c8 + 1 < C8;
        // the sums vb16[b] + toffs[k] are loop-invariant and the compiler would keep all 32 in registers across the loop: opaque to
        // it, they are re-formed per stage -- 32 additions
#pragma unroll
        for (int cg = 0; cg < CGN; ++cg) asm volatile("" : "+v"(vb16[cg]));
        if constexpr (kTight) asm volatile("" : "+v"(ttab));
        // sub-stage 0: this wave's weight DMAs of the sub-stage have landed -- they are OLDER than the fetch of the next stage's values
        // (requested behind them), which stays in flight: a counted wait
        if (more) __builtin_amdgcn_s_waitcnt(0x0f70 | (kFetchLoads & 15) | ((kFetchLoads >> 4) << 14));   // vmcnt(kFetchLoads)
        else __builtin_amdgcn_s_waitcnt(0x0f70);                                                          // vmcnt(0)
        __builtin_amdgcn_s_barrier();         // everybody's have, the stage buffer is complete, the other buffers are free
        dma_weights(c8, 1, 1);
        MX_FENCE();
        compute(std::integral_constant<int, 0>{}, ibuf, 0, e_cur);
        MX_FENCE();
        // sub-stage 1
        __builtin_amdgcn_s_waitcnt(0x0f70);   // vmcnt(0): the sub-stage's weights and the next stage's values
        if (more) {
            publish_amax(ibuf ^ 1);
            __builtin_amdgcn_s_waitcnt(0xc07f);
        }
        __builtin_amdgcn_s_barrier();
        if (more) dma_weights(c8 + 1, 0, 0);
        int e_next = e_cur;
        if (more) {
            e_next = stage_exp(ibuf ^ 1);
            stage_cut(ibuf ^ 1, e_next);
            if (c8 + 2 < C8) fetch_f32(c8 + 2);   // behind the weight DMAs just issued (the counted wait above relies on that order)
        }
        MX_FENCE();   // the cut's temporaries and the products' fragments are not to be live together
        compute(std::integral_constant<int, 1>{}, ibuf, 1, e_cur);
        e_cur = e_next;
        if (stage_shift(e_next) != s_acc) {   // block-uniform and, for inputs below 32768, never taken
            const float f = __builtin_bit_cast(float, (unsigned)(127 + s_acc - stage_shift(e_next)) << 23);
#pragma unroll
            for (int a = 0; a < 4; ++a)
#pragma unroll
                for (int b = 0; b < CGN; ++b) acc[a][b] *= f;
            s_acc = stage_shift(e_next);
        }
    }

    // ---- epilogue (the 16x16 C/D map: column = lane & 15 = voxel, register r of lane group kg = row 4 kg + r of the row group; row
    // groups 2q and 2q+1 together give a lane the eight consecutive channels 32 q + 8 kg .. + 7)
    const size_t plane = (size_t)H * W, vol = (size_t)D * plane;
    const float up = __builtin_bit_cast(float, (unsigned)(127 + s_acc) << 23);   // 2^s_acc
#pragma unroll
    for (int cg = 0; cg < CGN; ++cg) {
        const int g = CGN * wave + cg;
        const int d = d0 + g / (TH / RG16), h = h0 + RG16 * (g % (TH / RG16)) + col16 / TW, w = w0 + col16 % TW;
        if (d >= D || h >= H || w >= W) continue;
#pragma unroll
        for (int q = 0; q < 2; ++q) {
            float sc[8], sh[8], v[8];
            const int o0 = ob64 * 64 + 32 * q + 8 * kg;
            const size_t idx0 = ((size_t)n * Cout + o0) * vol + (size_t)d * plane + (size_t)h * W + w;
#pragma unroll
            for (int j = 0; j < 8; ++j) {
                sc[j] = scale ? scale[o0 + j] : 1.0f;
                sh[j] = scale ? shift[o0 + j] : 0.0f;
            }
#pragma unroll
            for (int j = 0; j < 8; ++j) {
                v[j] = (j < 4 ? acc[2 * q][cg][j & 3] : acc[2 * q + 1][cg][j & 3]) * up;
                if (scale) v[j] = fmaf(v[j], sc[j], sh[j]);
                if (relu) v[j] = fmaxf(v[j], 0.0f);
                if (out) out[idx0 + (size_t)j * vol] = v[j];
            }
            if (dst.scl || dst.pscl) bf_store_units(dst, v, n, Cout / 8, ob64 * 8 + 4 * q + kg, d, h, w);
        }
    }
}

// -----------------------------------------------------------------------------------------------------------------------------------
// The same convolution with the waves SPECIALISED (round 6, late): 8 consumer waves multiply (a 4 x 8 x 16 tile: 64 voxels x 64
// channels each), 4 producer waves fetch, scale, cut and stage the next channel group and issue every weight DMA.  In the kernel above
// all waves of the one block a CU holds walk through the same phases together -- fetch / cut (vector pipe, LDS writes), products
// (matrix pipes, LDS reads) -- so the pipes take turns instead of overlapping; here the producers' phase runs BESIDE the consumers'.
// One block-wide barrier per sub-stage.  The consumers hold no input registers and no cut temporaries: they fit the 168 registers a
// lane has at 12 waves.  Same LDS image, same weight image, same values bit for bit.
//
//   phase A of channel group c (sub-stage 0)                          phase B (sub-stage 1)
//   consumers: products of k-steps 0..3 + MX groups 0, 1              products of k-steps 4..6 + MX groups 2, 3
//   producers: DMA weights (c, 1); the values of c + 1 have           DMA weights (c + 1, 0); stage exponent of c + 1; cut and write
//              arrived: wave maxima -> the spare slots                its stage buffer; request the values of c + 2
// What-if builds (-DMX_WS_WHATIF=n, WRONG results, measurement only: profiles/r06_conv0_mx_whatif.txt): 1 = producers idle (DMAs and
// barriers only), 2 = consumers idle, 4 = no global fetch, 5 = no weight DMAs inside the loop, 6 = every block fetches one L2-resident tile,
// 7 = no epilogue.
template <int TD, int TH>
__global__ __launch_bounds__(TD * TH * 16 + 256) void conv3d_k3_fp16mx_ws_kernel(
    const float* __restrict__ xf, long long sN, long long sC, long long sD, long long sH, int Cin, const uint4* __restrict__ wq,
    const float* __restrict__ scale, const float* __restrict__ shift, BfOut dst, int C8, int Cout, int D, int H, int W, int tiles_w,
    int relu, int xcd_map) {
    constexpr int TW = 16, CGN = 4;
    constexpr int NCW = TD * TH * TW / 64;                 // consumer waves
    constexpr int NPW = 4, NPT = 64 * NPW;                 // producer waves / threads
    constexpr int HD = TD + 2, HH = TH + 2, HW = TW + 2;
    constexpr int NVOX = HD * HH * HW;
    constexpr int INS = bf_in_slots(TD, TH, TW);
    static_assert(INS - NVOX >= 2, "the stage's spare slots hold the producer waves' maxima");
    constexpr int kWSlots = kMxSlots * 64;
    constexpr int NV = (NVOX + NPT - 1) / NPT;             // halo voxels per producer thread
    constexpr int NVP = (NV + 1) / 2;
    constexpr int kFetchLoads = NV * 8;
    static_assert(kFetchLoads <= 63, "the counted wait");
    extern __shared__ uint4 s_mx[];
    constexpr int STAGE = 2 * INS;
    uint4* s_in = s_mx;
    uint4* s_w = s_mx + 2 * STAGE;
    float* __restrict__ out = dst.f32;

    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const bool producer = wave >= NCW;                     // wave-uniform
    unsigned bx, by, bz;
    xcd_block_id(xcd_map, bx, by, bz);
    const int bw = bx % tiles_w, bh = bx / tiles_w;
    const int nob = Cout / 64;
    const int n = bz / nob, ob64 = bz % nob;
    const int w0 = bw * TW, h0 = bh * TH, d0 = by * TD;
    auto stage_shift = [](int e1) -> int { return e1 > 12 ? e1 - 12 : 0; };
    auto stage_exp = [&](int buf) -> int {
        const float* spare = reinterpret_cast<const float*>(s_in + (size_t)buf * STAGE + NVOX);
        float a = 0.0f;
#pragma unroll
        for (int i = 0; i < NPW; ++i) a = fmaxf(a, spare[i]);
        return __builtin_amdgcn_readfirstlane(mx_block_exp(a));
    };

    if (producer) {
        // =============================================================================================== producers
        const int ptid = tid - 64 * NCW, pwave = wave - NCW;
        unsigned f_off[NV];
        float f_reg[2][NV][8];   // two sets: the values of channel group c + 2 are requested while those of c + 1 are still being cut
#if defined(MX_WS_WHATIF) && MX_WS_WHATIF == 6
        const float* xfn = xf;
#else
        const float* xfn = xf + (size_t)n * sN;
#endif
        const unsigned long long span = ((unsigned long long)(Cin - 1) * (unsigned long long)sC + (unsigned long long)(D - 1) * sD +
                                         (unsigned long long)(H - 1) * sH + W) * 4ull;
        const __amdgpu_buffer_rsrc_t rsrc = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(xfn), 0, (int)(unsigned)span, 0x00020000);
#pragma unroll
        for (int k = 0; k < NV; ++k) {
            const int slot = ptid + k * NPT;
            const int dz = slot / (HH * HW), r = slot - dz * (HH * HW), hy = r / HW, wx = r - hy * HW;
#if defined(MX_WS_WHATIF) && MX_WS_WHATIF == 6
            const int d = dz + 1, h = hy + 1, w = wx + 1;
#else
            const int d = d0 + dz - 1, h = h0 + hy - 1, w = w0 + wx - 1;
#endif
            const bool ok = slot < NVOX && d >= 0 && d < D && h >= 0 && h < H && w >= 0 && w < W;
            f_off[k] = ok ? (unsigned)(((long long)d * sD + (long long)h * sH + w) * 4) : 0xfffffff0u;
        }
        const unsigned sC4 = (unsigned)(sC * 4);
        auto fetch_f32 = [&](auto Ptag, int c8) __attribute__((always_inline)) {
            constexpr int P = decltype(Ptag)::value;
#if defined(MX_WS_WHATIF) && (MX_WS_WHATIF == 1 || MX_WS_WHATIF == 4)
            return;
#endif
#pragma unroll
            for (int j = 0; j < 8; ++j) {
                const int c = c8 * 8 + j;
                const unsigned soff = (c < Cin ? (unsigned)c : (unsigned)Cin) * sC4;
#pragma unroll
                for (int k = 0; k < NV; ++k)
                    f_reg[P][k][j] = __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(rsrc, f_off[k], (int)soff, 0));
            }
        };
        auto publish_amax = [&](auto Ptag, int buf) __attribute__((always_inline)) {
            constexpr int P = decltype(Ptag)::value;
            float a = 0.0f;
#pragma unroll
            for (int k = 0; k < NV; ++k)
#pragma unroll
                for (int j = 0; j < 8; ++j) a = fmaxf(a, fabsf(f_reg[P][k][j]));
#pragma unroll
            for (int off = 32; off > 0; off >>= 1) a = fmaxf(a, __shfl_xor(a, off, 64));
            float* spare = reinterpret_cast<float*>(s_in + (size_t)buf * STAGE + NVOX);
            if (lane == 0) spare[pwave] = a;
        };
        // voxel pairs [kp0, kp1) of the thread's NVP
        auto stage_cut = [&](auto Ptag, int buf, int e1, int kp0, int kp1) __attribute__((always_inline)) {
            constexpr int P = decltype(Ptag)::value;
#if defined(MX_WS_WHATIF) && MX_WS_WHATIF == 1
            return;
#endif
            uint4* hi_s = s_in + (size_t)buf * STAGE;
            uint4* q_s = hi_s + INS;
            const int S = stage_shift(e1);
            const float down = __builtin_bit_cast(float, (unsigned)(127 - S) << 23);
            const float sdiv = __builtin_bit_cast(float, (unsigned)(e1 - S + 127) << 23);
#pragma unroll
            for (int kp = 0; kp < NVP; ++kp) {
                if (kp < kp0 || kp >= kp1) continue;
                mx_f32x16 ga, gb;
                unsigned hw[2][4];
#pragma unroll
                for (int v = 0; v < 2; ++v) {
                    const int k = 2 * kp + v;
#pragma unroll
                    for (int j = 0; j < 8; ++j) {
                        const float f = (k < NV ? f_reg[P][k < NV ? k : 0][j] : 0.0f) * down;
                        const _Float16 h = (_Float16)f;
                        const float hf = (float)h;
                        const unsigned hb = __builtin_bit_cast(unsigned short, h);
                        if (j & 1) hw[v][j >> 1] |= hb << 16; else hw[v][j >> 1] = hb;
                        ga[8 * v + j] = hf;
                        gb[8 * v + j] = (f - hf) * 2048.0f;
                    }
                }
                asm volatile("s_nop 7\n\ts_nop 7" : "+v"(ga), "+v"(gb));   // (see the kernel above)
                const mx_u32x6 q = __builtin_amdgcn_cvt_scalef32_2xpk16_fp6_f32(ga, gb, sdiv);
#pragma unroll
                for (int v = 0; v < 2; ++v) {
                    const int k = 2 * kp + v, slot = ptid + k * NPT;
                    if (k < NV && slot < NVOX) {
                        // written by hand: before an LDS store it can see, the compiler waits for every LDS-DMA in flight (it cannot tell the
                        // weight buffers from the stage buffers) and, vmcnt being in-order, for the values requested behind them
                        const unsigned ah = (unsigned)(uintptr_t)(__attribute__((address_space(3))) void*)(hi_s + slot);
                        const mx_u32x4 dh = {hw[v][0], hw[v][1], hw[v][2], hw[v][3]}, dq = {q[3 * v], q[3 * v + 1], q[3 * v + 2], 0u};
                        asm volatile("ds_write_b128 %0, %1\n\tds_write_b128 %0, %2 offset:%3" ::"v"(ah), "v"(dh), "v"(dq), "n"(INS * 16) : "memory");
                    }
                }
            }
        };
        const uint4* wn = wq + (size_t)ob64 * C8 * (2 * kWSlots);
        auto dma_weights = [&](int c8, int s, int buf) __attribute__((always_inline)) {
            const uint4* src0 = wn + ((size_t)c8 * 2 + s) * kWSlots + lane;
#if defined(MX_WS_WHATIF) && MX_WS_WHATIF == 5
            if (c8 > 1) return;
#endif
            for (int i = pwave; i < kMxSlots; i += NPW) {
                uint4* dstp = s_w + (size_t)buf * kWSlots + i * 64;
                __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)(src0 + i * 64),
                                                 (__attribute__((address_space(3))) void*)dstp, 16, 0, 0);
            }
        };
        constexpr int KPA = NVP > 1 ? NVP - 1 : NVP;   // voxel pairs cut in phase A; the rest in phase B
        using P0 = std::integral_constant<int, 0>;
        using P1 = std::integral_constant<int, 1>;
        // prologue: stage 0 complete, the maxima of stage 1 published, the weights of (0, 0) landed
        if (C8 > 0) {
            dma_weights(0, 0, 0);
            fetch_f32(P0{}, 0);
            publish_amax(P0{}, 0);
            __builtin_amdgcn_s_waitcnt(0xc07f);
        }
        __builtin_amdgcn_s_barrier();                                  // P1: the maxima of stage 0
        if (C8 > 0) {
            if (C8 > 1) fetch_f32(P1{}, 1);
            stage_cut(P0{}, 0, stage_exp(0), 0, NVP);
            if (C8 > 1) publish_amax(P1{}, 1);
            __builtin_amdgcn_s_waitcnt(0x0f70);
            __builtin_amdgcn_s_waitcnt(0xc07f);
        }
        __builtin_amdgcn_s_barrier();                                  // P2
        // Channel group c (parity P): the values of c + 1 sit in set 1 - P and their maxima are published; phase A cuts most of them and
        // requests the values of c + 2 into set P; phase B cuts the rest, then -- a whole phase after the request -- takes the maxima
        // of c + 2.  (One set, requested at the end of B and awaited at the start of A, left the consumers waiting at barrier A for
        // the full latency of the fetch: together 3.65 ms where the consumers alone take 2.73 and the producers alone 2.38.)
        auto group = [&](auto Ptag, int c8) __attribute__((always_inline)) {
            constexpr int P = decltype(Ptag)::value;
            using Q = std::integral_constant<int, 1 - P>;
            const int ibuf = P;
            const bool more1 = c8 + 1 < C8, more2 = c8 + 2 < C8;
            int e_next = 0;
            // phase A
            if (more1) e_next = stage_exp(ibuf ^ 1);   // (an LDS read the compiler sees: before the DMAs, or it waits for them)
            dma_weights(c8, 1, 1);
            if (more2) fetch_f32(Ptag, c8 + 2);   // first thing: the whole group is its latency cover (without it conv0 takes 2.6 ms, with
                                                  // the request at the end of phase A 3.6)
            MX_FENCE();                           // (the scheduler otherwise moves the request behind the cut)
            if (more1) stage_cut(Q{}, ibuf ^ 1, e_next, 0, KPA);
            if (more2) __builtin_amdgcn_s_waitcnt(0x0f70 | (kFetchLoads & 15) | ((kFetchLoads >> 4) << 14));   // the DMAs (older) have landed
            else __builtin_amdgcn_s_waitcnt(0x0f70);
            __builtin_amdgcn_s_waitcnt(0xc07f);
            __builtin_amdgcn_s_barrier();                              // end of A
            // phase B
            if (more1) {
                dma_weights(c8 + 1, 0, 0);
                stage_cut(Q{}, ibuf ^ 1, e_next, KPA, NVP);
            }
            if (more2) publish_amax(Ptag, ibuf);                       // the compiler's own wait for the values; into the spare slots of buffer
                                                                       // ibuf, which the consumers read at the top of a group only
            __builtin_amdgcn_s_waitcnt(0x0f70);
            __builtin_amdgcn_s_waitcnt(0xc07f);
            __builtin_amdgcn_s_barrier();                              // end of B
        };
        for (int c8 = 0; c8 < C8; c8 += 2) {
            group(P0{}, c8);
            if (c8 + 1 < C8) group(P1{}, c8 + 1);
        }
        return;
    }

    // =================================================================================================== consumers
    constexpr int RG16 = 16 / TW > 0 ? 16 / TW : 1;
    const int col16 = lane & 15, kg = lane >> 4;
    // A wave's four column groups are four consecutive h-rows of the tile (TW = 16: one row per group; TH = 8: groups 4 wave .. + 3 never
    // wrap into the next plane): their fragments sit HW slots apart -- ONE lane address per k-step, the group in the instruction's offset.
    static_assert(TW == 16 && TH == 8 && CGN == 4, "the column groups of a wave are consecutive rows");
    int badr[7];   // byte offset, inside a stage buffer, of the lane's first voxel + the tap its lane group reads at k-step ks
    {
        const int g = CGN * wave;
        const int vb0 = ((g / TH) * HH + g % TH) * HW + col16;
#pragma unroll
        for (int ks = 0; ks < 7; ++ks) {
            const int toff = kg == 0 ? bf_tap_off<HH, HW>(4 * ks) : kg == 1 ? bf_tap_off<HH, HW>(4 * ks + 1)
                           : kg == 2 ? bf_tap_off<HH, HW>(4 * ks + 2) : bf_tap_off<HH, HW>(4 * ks + 3);
            badr[ks] = (vb0 + toff) * 16;
        }
    }
    mx_f32x4 acc[4][CGN];
#pragma unroll
    for (int a = 0; a < 4; ++a)
#pragma unroll
        for (int b = 0; b < CGN; ++b) acc[a][b] = (mx_f32x4){0.0f, 0.0f, 0.0f, 0.0f};

    // The products of one sub-stage, SOFTWARE-PIPELINED by hand: the fragments of step i + 1 are requested before the 16 matrix
    // instructions of step i are issued (two fragment sets; 256 - 300 clocks of cover), so that a consumer wave waits for LDS once per
    // sub-stage (behind the barrier) instead of before every group of four instructions -- with two such waves per SIMD the matrix
    // pipe was 0.39 busy.  The fences keep the compiler from re-ordering the requests behind the products; the accumulation order of
    // every accumulator is unchanged (fp16 steps, then the MX groups): the same bits.
    auto compute = [&](auto sc, int ibuf, int wbuf, int e1) {
#if defined(MX_WS_WHATIF) && MX_WS_WHATIF == 2
        return;
#endif
        constexpr int s = decltype(sc)::value;
        constexpr int nks = s == 0 ? 4 : 3;
        const char* const sbase = reinterpret_cast<const char*>(s_in + (size_t)ibuf * STAGE);
        const uint4* ain = s_w + (size_t)wbuf * kWSlots + lane;
        const int sb = e1 - stage_shift(e1) + 127;
        f16x8 A[4], B[2][CGN];
        mx_i32x8 FB[CGN], FA[2];
        int sa[2];
        auto load_a = [&](int kl, int a0) __attribute__((always_inline)) {   // weight fragments a0, a0 + 1 of k-step kl
#pragma unroll
            for (int a = a0; a < a0 + 2; ++a) A[a] = __builtin_bit_cast(f16x8, ain[(kl * 4 + a) * 64]);
        };
        auto load_b = [&](int kl, int set) __attribute__((always_inline)) {
            const char* const p = sbase + badr[4 * s + kl];
#pragma unroll
            for (int b = 0; b < CGN; ++b) B[set][b] = __builtin_bit_cast(f16x8, *reinterpret_cast<const uint4*>(p + b * (HW * 16)));
        };
        auto load_fb = [&](int gl) __attribute__((always_inline)) {
            // (sub-stage 1 has three k-steps: its second MX group pairs taps 24 .. 27 with themselves, as the weight image does)
            const char* const p0 = sbase + INS * 16 + badr[4 * s + 2 * gl], * const p1 = sbase + INS * 16 + badr[(4 * s + 2 * gl + 1) < 7 ? 4 * s + 2 * gl + 1 : 6];
#pragma unroll
            for (int b = 0; b < CGN; ++b) {
                const uint4 u0 = *reinterpret_cast<const uint4*>(p0 + b * (HW * 16)), u1 = *reinterpret_cast<const uint4*>(p1 + b * (HW * 16));
                FB[b] = (mx_i32x8){(int)u0.x, (int)u0.y, (int)u0.z, (int)u1.x, (int)u1.y, (int)u1.z, 0, 0};
            }
        };
        auto load_fa = [&](int m, int set) __attribute__((always_inline)) {   // m = 4 gl + a
            const uint4 lo = ain[(16 + 2 * m) * 64], hi = ain[(16 + 2 * m + 1) * 64];
            FA[set] = (mx_i32x8){(int)lo.x, (int)lo.y, (int)lo.z, (int)lo.w, (int)hi.x, (int)hi.y, 0, 0};
            sa[set] = (int)hi.z;
        };
        auto pin = [&]() __attribute__((always_inline)) {   // the optimiser otherwise sinks products below the code that follows them
            asm volatile("" : "+v"(acc[0][0]), "+v"(acc[0][1]), "+v"(acc[0][2]), "+v"(acc[0][3]), "+v"(acc[1][0]), "+v"(acc[1][1]),
                              "+v"(acc[1][2]), "+v"(acc[1][3]), "+v"(acc[2][0]), "+v"(acc[2][1]), "+v"(acc[2][2]), "+v"(acc[2][3]),
                              "+v"(acc[3][0]), "+v"(acc[3][1]), "+v"(acc[3][2]), "+v"(acc[3][3]));
        };
        // fp16 steps in two halves of 8 instructions (weight fragments 0, 1 | 2, 3): the voxel fragments of step i + 1 are requested
        // before step i (two sets), a weight fragment pair of step i + 1 as soon as the half that reads its registers is issued: every
        // request is >= 128 matrix-pipe clocks ahead of its first use, on 48 fragment registers (64 with both kinds in two sets spill)
        load_a(0, 0);
        load_a(0, 2);
        load_b(0, 0);
#pragma unroll
        for (int kl = 0; kl < nks; ++kl) {
            const bool last = kl + 1 == nks;
            if (!last) load_b(kl + 1, (kl + 1) & 1);
            else load_fb(0);
            MX_FENCE();
#pragma unroll
            for (int a = 0; a < 2; ++a)
#pragma unroll
                for (int b = 0; b < CGN; ++b) acc[a][b] = __builtin_amdgcn_mfma_f32_16x16x32_f16(A[a], B[kl & 1][b], acc[a][b], 0, 0, 0);
            pin();
            MX_FENCE();
            if (!last) load_a(kl + 1, 0);
            else load_fa(0, 0);
            MX_FENCE();
#pragma unroll
            for (int a = 2; a < 4; ++a)
#pragma unroll
                for (int b = 0; b < CGN; ++b) acc[a][b] = __builtin_amdgcn_mfma_f32_16x16x32_f16(A[a], B[kl & 1][b], acc[a][b], 0, 0, 0);
            pin();
            MX_FENCE();
            if (!last) load_a(kl + 1, 2);
        }
        // MX groups: one set of voxel fragments (the second group's are requested behind the first group's last products: one exposed
        // LDS latency per sub-stage), the weight fragment of product m + 1 requested before product m
#pragma unroll
        for (int m = 0; m < 8; ++m) {
            const int gl = m >> 2, a = m & 3;
            (void)gl;
            if (m + 1 < 8) load_fa(m + 1, (m + 1) & 1);
            MX_FENCE();
#pragma unroll
            for (int b = 0; b < CGN; ++b)
                acc[a][b] = __builtin_amdgcn_mfma_scale_f32_16x16x128_f8f6f4(FA[m & 1], FB[b], acc[a][b], 2, 2, 0, sa[m & 1], 0, sb);
            asm volatile("" : "+v"(acc[a][0]), "+v"(acc[a][1]), "+v"(acc[a][2]), "+v"(acc[a][3]));
            MX_FENCE();
            if (m == 3) load_fb(1);
        }
    };

    int s_acc = 0;
    __builtin_amdgcn_s_barrier();                                      // P1
    __builtin_amdgcn_s_barrier();                                      // P2
    if (C8 > 0) s_acc = stage_shift(stage_exp(0));
    for (int c8 = 0; c8 < C8; ++c8) {
        const int ibuf = c8 & 1;
        const int e_cur = stage_exp(ibuf);
        if (stage_shift(e_cur) != s_acc) {   // block-uniform and, for inputs below 32768, never taken
            const float f = __builtin_bit_cast(float, (unsigned)(127 + s_acc - stage_shift(e_cur)) << 23);
#pragma unroll
            for (int a = 0; a < 4; ++a)
#pragma unroll
                for (int b = 0; b < CGN; ++b) acc[a][b] *= f;
            s_acc = stage_shift(e_cur);
        }
        compute(std::integral_constant<int, 0>{}, ibuf, 0, e_cur);
        __builtin_amdgcn_s_waitcnt(0xc07f);
        __builtin_amdgcn_s_barrier();                                  // end of A: every consumer is done with weight buffer 0
        compute(std::integral_constant<int, 1>{}, ibuf, 1, e_cur);
        __builtin_amdgcn_s_waitcnt(0xc07f);
        __builtin_amdgcn_s_barrier();                                  // end of B: done with weight buffer 1 and stage buffer ibuf
    }

    const size_t plane = (size_t)H * W, vol = (size_t)D * plane;
    const float up = __builtin_bit_cast(float, (unsigned)(127 + s_acc) << 23);
#if defined(MX_WS_WHATIF) && MX_WS_WHATIF == 7
    if (acc[0][0][0] != 12345.678f) return;   // (what-if: no epilogue)
#endif
#pragma unroll
    for (int cg = 0; cg < CGN; ++cg) {
        const int g = CGN * wave + cg;
        const int d = d0 + g / (TH / RG16), h = h0 + RG16 * (g % (TH / RG16)) + col16 / TW, w = w0 + col16 % TW;
        if (d >= D || h >= H || w >= W) continue;
#pragma unroll
        for (int q = 0; q < 2; ++q) {
            float sc[8], sh[8], v[8];
            const int o0 = ob64 * 64 + 32 * q + 8 * kg;
            const size_t idx0 = ((size_t)n * Cout + o0) * vol + (size_t)d * plane + (size_t)h * W + w;
#pragma unroll
            for (int j = 0; j < 8; ++j) {
                sc[j] = scale ? scale[o0 + j] : 1.0f;
                sh[j] = scale ? shift[o0 + j] : 0.0f;
            }
#pragma unroll
            for (int j = 0; j < 8; ++j) {
                v[j] = (j < 4 ? acc[2 * q][cg][j & 3] : acc[2 * q + 1][cg][j & 3]) * up;
                if (scale) v[j] = fmaf(v[j], sc[j], sh[j]);
                if (relu) v[j] = fmaxf(v[j], 0.0f);
                if (out) out[idx0 + (size_t)j * vol] = v[j];
            }
            if (dst.scl || dst.pscl) bf_store_units(dst, v, n, Cout / 8, ob64 * 8 + 4 * q + kg, d, h, w);
        }
    }
}
