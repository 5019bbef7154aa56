// The transposed layer of the cost network (mvsnet.py:92-100: ConvTranspose3d k3 s2 p1 op1 + BatchNorm + ReLU, the skip tensor
// added last, :110-111) as a PERSISTENT kernel: one resident block per CU walks a queue of work items and stores the results of
// item i while it multiplies item i + 1.  Included by costreg_bf16.hip inside namespace mvsdet, after the all-classes kernel
// (convT3d_k3_s2_bf16x3_fused_kernel), whose sums it repeats bit for bit.
//
// Why: the all-classes kernel needs the whole LDS of a CU (159.8 KB: three stages of halo tile + 27 taps of weights), so one block
// runs per CU and its loop (matrix pipes) and its epilogue (1.2 GB of skip tensor read + result written at conv11) are strictly
// serial: 0.33 + 0.30 ms.  A second accumulator set does not fit (8 classes x 16 registers = 128 of the 168 a wave has at 12 waves).
//
// How: a work item is HALF of the eight output parity classes of a (3 x 16 x 8 coarse tile, 32 output channels):
//     half 0 = (pd, ph) in {(0,0), (1,1)}: classes 0, 1, 6, 7 -- 8 tap pairs, 4 accumulators
//     half 1 = (pd, ph) in {(0,1), (1,0)}: classes 2, 3, 4, 5 -- 6 tap pairs, 4 accumulators
// A wave holds both sets (128 registers, as before): while it accumulates one, the other -- finished in the item before -- leaves in
// four chunks of (one (pd, ph), 8 channels): the skip tensor's values arrive by LDS-DMA (no registers) at the start of a quarter of
// the item's channel loop, scale / shift / ReLU / addition and the stores happen at its end.  Every accumulator still sums the same
// (channel group, tap pair) sequence as in the all-classes kernel: the same bits.  The halo tile is staged twice per tile (once per
// half: +46 % LDS-DMA bytes), the weights once (a half needs only its own taps).  The DMA pipeline runs across items: the first two
// stages of item i + 1 are requested during the last two of item i.
//
// LDS: 3 stages x (2 pieces x 640 slots of halo tile + 16 chunks x 64 slots of weights) x 16 B = 110,592 B; 12 waves x 4 KB of
// skip values; scale and shift of up to 256 channels: 161,792 B.
//
// Counted waits: every vector-memory instruction of a wave is issued unconditionally (lanes outside the volume, and the chunks of an
// item that does not exist, go to offsets beyond the buffer descriptors' sizes), so "all but the newest N" is exact:
// per stage 3 DMAs; per quarter boundary S stores (8 float2 / 4 units) + R skip-tensor DMAs (16).  No spill may exist (scratch
// traffic counts in vmcnt): the build checks.  The accounting takes vmcnt to retire loads, LDS-DMAs and stores in issue order -- the model
// LLVM's own wait insertion uses on gfx9 (one event class for all vector memory, no separate store counter); the data of a stage is
// requested two stages (microseconds) before the wait that guards it.  One more reason why this kernel is an option, not the default.
constexpr int kCtpIn = 640;                                  // slots per piece per stage: 10 DMAs (612 halo voxels)
constexpr int kCtpW = 16 * 64;                               // (local pair 8, piece 2) chunks of 64 slots
constexpr int kCtpStage = 2 * kCtpIn + kCtpW;                // 2304 slots = 36,864 B
constexpr int kCtpResFloats = 12 * 1024;                     // per wave [j 8][x / y][lane 64]
constexpr int kCtpMaxCout = 256;
__host__ __device__ constexpr size_t ctp_lds_bytes() { return (size_t)3 * kCtpStage * 16 + (size_t)kCtpResFloats * 4 + 2 * kCtpMaxCout * 4; }

#define CTP_FENCE()                            \
    do {                                       \
        asm volatile("" ::: "memory");         \
        __builtin_amdgcn_sched_barrier(0);     \
    } while (0)

typedef unsigned ctp_u32x2 __attribute__((ext_vector_type(2)));
typedef unsigned ctp_u32x4 __attribute__((ext_vector_type(4)));

template <int N>
__device__ __forceinline__ void ctp_wait_vm() {
    static_assert(N >= 0 && N < 64, "vmcnt has six bits");
    __builtin_amdgcn_s_waitcnt(0x0f70 | (N & 15) | ((N >> 4) << 14));
}

// whatif (what-if BUILDS only, -DMVS_CONVT_WHATIF=n; 0 in the product): bit 0 sends the skip-tensor loads, bit 1 the stores, bit 2 / 3 make every
// block read one halo tile / one weights stage -- measurement of what bounds the kernel, wrong results.
// OUT_SCL: the result leaves as the SCL form (bf16 pieces, conv9); else as fp32 NCDHW (conv11).  RES: a skip tensor is added.
template <bool OUT_SCL, bool RES>
__global__ __launch_bounds__(768) void convT3d_k3_s2_bf16x3_persist_kernel(
    const uint4* __restrict__ xs, const uint4* __restrict__ wq, const float* __restrict__ scale, const float* __restrict__ shift,
    const float* __restrict__ residual, BfOut dst, int C8, int Cout, int Di, int Hi, int Wi, int Dp, int Hp, int Wp, size_t piece_stride,
    int tiles_w, int tiles_h, int tiles_d, int N, int relu, unsigned out_bytes, unsigned res_bytes, int xcd_map, int whatif) {
    constexpr int TD = 3, TH = 16, TW = 8, NW = 12, RG = 4;
    constexpr int HH = TH + 1, HW = TW + 1, NVOX = (TD + 1) * HH * HW;
    static_assert(NVOX <= kCtpIn, "the halo tile fits its slots");
    constexpr int S_OPS = OUT_SCL ? 4 : 8, R_OPS = RES ? 16 : 0, K_OPS = S_OPS + R_OPS;
    extern __shared__ uint4 s_ctp[];
    float* const s_res_all = reinterpret_cast<float*>(s_ctp + 3 * kCtpStage);
    float* const s_ss = s_res_all + kCtpResFloats;           // [0, 256): scale, [256, 512): shift

    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int col = lane & 31, hh = lane >> 5;
    float* const s_res = s_res_all + wave * 1024;
    const int nob = Cout / 32;

    // scale / shift of every channel, once
    for (int c = tid; c < Cout; c += 64 * NW) {
        s_ss[c] = scale ? scale[c] : 1.0f;
        s_ss[kCtpMaxCout + c] = scale ? shift[c] : 0.0f;
    }
    __builtin_amdgcn_s_waitcnt(0x0070);   // vmcnt 0, lgkmcnt 0: the counted waits below start from an empty queue
    __syncthreads();

    // this block's range of tiles (a tile = its two halves, one after the other); an XCD takes a contiguous eighth of the queue
    const unsigned tiles = (unsigned)(tiles_w * tiles_h * tiles_d) * (unsigned)N * (unsigned)nob;
    const unsigned G = gridDim.x;
    const unsigned lb = xcd_map ? (blockIdx.x & 7u) * (G >> 3) + (blockIdx.x >> 3) : blockIdx.x;
    const unsigned tlo = (unsigned)((unsigned long long)tiles * lb / G), thi = (unsigned)((unsigned long long)tiles * (lb + 1) / G);
    if (tlo >= thi) return;   // block-uniform

    const size_t c8_stride = (size_t)Dp * Hp * Wp;
    const int Do = 2 * Di, Ho = 2 * Hi, Wo = 2 * Wi;
    const unsigned oplane = (unsigned)Ho * Wo, ovol = (unsigned)Do * oplane;
    // the lane's coarse voxel inside a tile
    const int ld = wave / (TH / RG);
    const int vb = (ld * HH + RG * (wave % (TH / RG)) + col / TW) * HW + col % TW;

    // The 36 DMAs of a stage, three per wave: job = wave + 12 k.  Jobs 0 .. 19: the halo tile (piece job / 10, chunk job % 10);
    // 20 .. 35: weights chunk job - 20 (half 1 has 12: its last four jobs repeat chunks 0 .. 3 into spare slots).
    // k = 0 is always input, k = 2 always weights, k = 1 input for waves 0 .. 7.  Lane offsets in bytes, constant for the kernel:
    unsigned voff[3];
#pragma unroll
    for (int k = 0; k < 2; ++k) {
        const int job = wave + NW * k;
        const int slot = (job % 10) * 64 + lane;
        const int sv = slot < NVOX ? slot : 0;
        const int dz = sv / (HH * HW), r = sv - dz * (HH * HW), hy = r / HW, wx = r - hy * HW;
        voff[k] = job < 20 ? (unsigned)(((size_t)dz * Hp + hy) * Wp + wx) * 16u : (unsigned)lane * 16u;
    }
    voff[2] = (unsigned)lane * 16u;
    const bool in1 = wave < 8;
    const int piece0 = wave >= 10 ? 1 : 0;

    auto uniform_ptr = [](const uint4* p) __attribute__((always_inline)) {
        const unsigned long long v = (unsigned long long)(uintptr_t)p;
        const unsigned lo32 = (unsigned)__builtin_amdgcn_readfirstlane((int)(unsigned)v), hi32 = (unsigned)__builtin_amdgcn_readfirstlane((int)(unsigned)(v >> 32));
        return reinterpret_cast<const char*>((uintptr_t)(((unsigned long long)hi32 << 32) | lo32));
    };
    struct Item {
        const uint4* xn;    // the tile's first halo voxel, channel group 0, piece 0
        const uint4* wn;    // weights of the item's 32 output channels: [c8][pair 14][row group 2][piece 2][lane 64] + row group
        int n, ob32, d0, h0, w0;
    };
    auto decode = [&](unsigned t) __attribute__((always_inline)) {
        Item I;
        I.ob32 = (int)(t % (unsigned)nob); t /= (unsigned)nob;
        const int tw = (int)(t % (unsigned)tiles_w); t /= (unsigned)tiles_w;
        const int th = (int)(t % (unsigned)tiles_h); t /= (unsigned)tiles_h;
        const int td = (int)(t % (unsigned)tiles_d); t /= (unsigned)tiles_d;
        I.n = (int)t;
        I.d0 = td * TD; I.h0 = th * TH; I.w0 = tw * TW;
        I.xn = xs + ((size_t)I.n * C8) * c8_stride + ((size_t)(I.d0 + 1) * Hp + (I.h0 + 1)) * Wp + (I.w0 + 1);
        I.wn = wq + (size_t)(I.ob32 >> 1) * C8 * (kBfPairs * 4 * 64) + (size_t)(I.ob32 & 1) * (2 * 64);
        return I;
    };
    // where the lane's outputs of an item go: base offsets (elements / units) without the chunk's terms.  Worked out again at every
    // boundary from the item's (scalar) coordinates and a lane index read off the execution mask: nothing of it stays in registers over the
    // stages in between (the loop has none to spare: 128 accumulators + fragments of the 168).
    struct Dest {
        unsigned obase, rbase;
        int hh, lane;
        unsigned away;   // 0: the lane's voxel exists; 2^31: it does not -- or-ed into the lane's byte offset it sends loads and stores beyond
    };                   // the descriptors' sizes (< 2^31: host), where loads return 0 and stores are dropped.  Opaque to the compiler, which
                         // otherwise turns the selects into two masked copies of every memory instruction
    auto dest_of = [&](const Item& I, int valid) __attribute__((always_inline)) {
        int t;   // the lane index, from no register (volatile: not hoisted out of the loop and kept)
        asm volatile("v_mbcnt_lo_u32_b32 %0, -1, 0\n\tv_mbcnt_hi_u32_b32 %0, -1, %0" : "=v"(t));
        Dest d;
        d.lane = t;
        d.hh = (t >> 5) & 1;
        const int c = t & 31;
        const int di = I.d0 + ld, hi_ = I.h0 + RG * (wave % (TH / RG)) + c / TW, wi = I.w0 + c % TW;
        d.away = (valid && di < Di && hi_ < Hi && wi < Wi) ? 0u : 0x80000000u;
        asm volatile("" : "+v"(d.away));
        d.rbase = ((unsigned)I.n * Cout + I.ob32 * 32) * ovol + (unsigned)(2 * di) * oplane + (unsigned)(2 * hi_) * Wo + 2 * wi;
        if (OUT_SCL)
            d.obase = (((unsigned)I.n * (Cout / 8) + I.ob32 * 4) * dst.Dp + (2 * di + 1)) * (unsigned)(dst.Hp * dst.Wp) + (unsigned)(2 * hi_ + 1) * dst.Wp + (2 * wi + 1);
        else
            d.obase = d.rbase;
        return d;
    };

    const __amdgpu_buffer_rsrc_t out_rsrc = __builtin_amdgcn_make_buffer_rsrc(OUT_SCL ? (void*)dst.scl : (void*)dst.f32, 0, (int)out_bytes, 0x00020000);
    const __amdgpu_buffer_rsrc_t res_rsrc = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(residual), 0, (int)res_bytes, 0x00020000);

    // one stage of an item of half H into LDS buffer buf: 3 DMAs per wave, every source = a wave-uniform base + the lane's constant offset
    auto dma_stage = [&](auto Htag, const Item& I, int c8, int buf) __attribute__((always_inline)) {
        constexpr int H = decltype(Htag)::value;
        uint4* const sb = s_ctp + (size_t)buf * kCtpStage;
        const uint4* const xc = ((whatif & 4) ? xs + ((size_t)Hp + 1) * Wp + 1 : I.xn) + (size_t)((whatif & 4) ? 0 : c8) * c8_stride;
        const uint4* const wc = I.wn + (size_t)((whatif & 8) ? 0 : c8) * (kBfPairs * 4 * 64);
        auto wsrc = [&](int j) {   // weights chunk j of the half: (local pair j >> 1, piece j & 1) -> its place in the layer's 14 pairs
            const int sc = (H == 1 && j >= 12) ? j - 12 : j;
            const int lp = sc >> 1, pl = H == 0 ? (lp < 2 ? lp : lp + 6) : lp + 2;
            return wc + (size_t)(pl * 4 + (sc & 1)) * 64;
        };
        const char* const b0 = uniform_ptr(xc + (size_t)piece0 * piece_stride);
        const char* const b1 = uniform_ptr(in1 ? xc + piece_stride : wsrc(wave - 8));
        const char* const b2 = uniform_ptr(wsrc(wave + 4));
        uint4* const d0 = sb + piece0 * kCtpIn + (wave - 10 * piece0) * 64;
        uint4* const d1 = in1 ? sb + kCtpIn + (wave + 2) * 64 : sb + 2 * kCtpIn + (wave - 8) * 64;
        uint4* const d2 = sb + 2 * kCtpIn + (wave + 4) * 64;
        __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)(b0 + voff[0]), (__attribute__((address_space(3))) void*)d0, 16, 0, 0);
        __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)(b1 + voff[1]), (__attribute__((address_space(3))) void*)d1, 16, 0, 0);
        __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)(b2 + voff[2]), (__attribute__((address_space(3))) void*)d2, 16, 0, 0);
    };

    f32x16b acc[2][4];   // [half][class of the half]: half 0 = classes 0, 1, 6, 7; half 1 = classes 2, 3, 4, 5
    const bf16x8* const s8 = reinterpret_cast<const bf16x8*>(s_ctp);

    auto compute = [&](auto Htag, int buf) __attribute__((always_inline)) {
        constexpr int H = decltype(Htag)::value;
        const bf16x8* bin = s8 + (size_t)buf * kCtpStage + vb;
        const bf16x8* ain = s8 + (size_t)buf * kCtpStage + 2 * kCtpIn + lane;
#pragma unroll
        for (int ai = 0; ai < 4; ++ai) {
            constexpr int kPi[2][4] = {{0, 1, 6, 7}, {2, 3, 4, 5}};
            const int pi = kPi[H][ai];
#pragma unroll
            for (int pj = 0; pj < s2_pairs(pi); ++pj) {
                const int pl = s2_first_pair(pi) + pj;
                const int lp = H == 0 ? (pl < 2 ? pl : pl - 6) : pl - 2;
                const int toff = hh ? ct_tap_off<HH, HW>(pi, 2 * pj + 1) : ct_tap_off<HH, HW>(pi, 2 * pj);
                const bf16x8 a0 = ain[(lp * 2 + 0) * 64], a1 = ain[(lp * 2 + 1) * 64];
                const bf16x8 b0 = bin[toff], b1 = bin[(size_t)kCtpIn + toff];
                acc[H][ai] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a1, b0, acc[H][ai], 0, 0, 0);
                acc[H][ai] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a0, b1, acc[H][ai], 0, 0, 0);
                acc[H][ai] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a0, b0, acc[H][ai], 0, 0, 0);
            }
        }
    };

    // chunk c of the set of half Hd: (pd, ph) class c >> 1 of the half, channels 8 (2 q + hh) .. + 7 of the item's 32, q = c & 1
    // (half 0: (pd, ph) = (0,0), (1,1); half 1: (0,1), (1,0)).  R: its 16 skip-tensor values per lane, by LDS-DMA into the wave's own 4 KB
    auto res_dma = [&](auto Htag, int c, const Item& I, int valid) __attribute__((always_inline)) {
        constexpr int Hd = decltype(Htag)::value;
        if (!RES) return;
        const Dest d = dest_of(I, valid);
        const int cls2 = c >> 1, q = c & 1;
        const int pd = cls2, ph = Hd == 0 ? cls2 : 1 - cls2;
        // one lane offset for the chunk; (channel j, x / y) ride on the instruction's scalar offset
        const unsigned v = ((d.rbase + (unsigned)(16 * q + 8 * d.hh) * ovol) * 4u) | d.away | ((whatif & 1) ? 0x80000000u : 0u);
        const unsigned s0 = ((unsigned)pd * oplane + (unsigned)ph * Wo) * 4u;
#pragma unroll
        for (int j = 0; j < 8; ++j)
#pragma unroll
            for (int xy = 0; xy < 2; ++xy)
                __builtin_amdgcn_raw_ptr_buffer_load_lds(res_rsrc, (__attribute__((address_space(3))) void*)(s_res + (2 * j + xy) * 64), 4, v,
                                                         s0 + ((unsigned)j * ovol + xy) * 4u, 0, 0);
    };
    // F: finish and store the chunk (S stores).  The caller has waited for the chunk's skip values.
    auto finish = [&](auto Htag, int c, const Item& I, int valid) __attribute__((always_inline)) {
        constexpr int Hd = decltype(Htag)::value;
        const Dest d = dest_of(I, valid);
        const int hh = d.hh, lane = d.lane, ob32 = I.ob32;
        const int cls2 = c >> 1, q = c & 1;
        const int pd = cls2, ph = Hd == 0 ? cls2 : 1 - cls2;
        const int ch0 = ob32 * 32 + 16 * q + 8 * hh;
        const unsigned v = ((d.obase + (unsigned)(16 * q + 8 * hh) * ovol) * 4u) | d.away | ((whatif & 2) ? 0x80000000u : 0u);          // fp32 form
        const unsigned s0 = ((unsigned)pd * oplane + (unsigned)ph * Wo) * 4u;
        float v0[8], v1[8];
#pragma unroll
        for (int j = 0; j < 8; ++j) {
            float2 val = make_float2(acc[Hd][2 * cls2][8 * q + j], acc[Hd][2 * cls2 + 1][8 * q + j]);
            if (scale) {
                const float sc = s_ss[ch0 + j], sh = s_ss[kCtpMaxCout + ch0 + j];
                val.x = fmaf(val.x, sc, sh);
                val.y = fmaf(val.y, sc, sh);
            }
            if (relu) {
                val.x = fmaxf(val.x, 0.0f);
                val.y = fmaxf(val.y, 0.0f);
            }
            if (RES) {
                val.x = s_res[(2 * j) * 64 + lane] + val.x;
                val.y = s_res[(2 * j + 1) * 64 + lane] + val.y;
            }
            if (OUT_SCL) {
                v0[j] = val.x;
                v1[j] = val.y;
            } else {
                __builtin_amdgcn_raw_buffer_store_b64(ctp_u32x2{__float_as_uint(val.x), __float_as_uint(val.y)}, out_rsrc, v, s0 + (unsigned)j * ovol * 4u, 0);
            }
        }
        if (OUT_SCL) {
            uint4 h0, m0, h1, m1;
            bf_cut8(v0, h0, m0);
            bf_cut8(v1, h1, m1);
            const unsigned u = d.obase + (unsigned)(2 * q + hh) * (unsigned)(dst.Dp * dst.Hp * dst.Wp);
            const unsigned vu = (u * 16u) | d.away | ((whatif & 2) ? 0x80000000u : 0u);
            const unsigned su = ((unsigned)pd * (unsigned)(dst.Hp * dst.Wp) + (unsigned)ph * dst.Wp) * 16u, sp = (unsigned)dst.piece * 16u;
            __builtin_amdgcn_raw_buffer_store_b128(ctp_u32x4{h0.x, h0.y, h0.z, h0.w}, out_rsrc, vu, su, 0);
            __builtin_amdgcn_raw_buffer_store_b128(ctp_u32x4{m0.x, m0.y, m0.z, m0.w}, out_rsrc, vu, su + sp, 0);
            __builtin_amdgcn_raw_buffer_store_b128(ctp_u32x4{h1.x, h1.y, h1.z, h1.w}, out_rsrc, vu, su + 16u, 0);
            __builtin_amdgcn_raw_buffer_store_b128(ctp_u32x4{m1.x, m1.y, m1.z, m1.w}, out_rsrc, vu, su + sp + 16u, 0);
        }
    };

    Item cur = decode(tlo), nxt = cur, drn = cur;
    int drn_valid = 0;
    const int SQ = C8 >> 2;     // stages per quarter (host: C8 % 4 == 0, SQ >= 3)
    int buf = 0;

    // One half of a tile: accumulates set H over the C8 stages while the set of the half before (1 - H) leaves in four chunks.
    // A stage requests the stage two ahead: of this half, or -- from its last two stages -- the first two of the half that follows
    // (item `after`, half 1 - H).
    auto phase = [&](auto Htag, const Item& after) __attribute__((always_inline)) {
        constexpr int H = decltype(Htag)::value;
        using Hd = std::integral_constant<int, 1 - H>;
        // zeros written in place (volatile: the compiler otherwise keeps ONE tuple of zeros alive through the kernel to copy from)
#pragma unroll
        for (int ai = 0; ai < 4; ++ai)
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                float z;
                asm volatile("v_mov_b32 %0, 0" : "=v"(z));
                acc[H][ai][r] = z;
            }
        int c8 = 0;
#pragma unroll
        for (int qd = 0; qd < 4; ++qd) {
            CTP_FENCE();
            res_dma(Hd{}, qd, drn, drn_valid);
            CTP_FENCE();
#pragma clang loop unroll(disable)
            for (int i = 0; i < SQ; ++i, ++c8) {
                // the S + R operations of the boundary are newer than the DMAs of its first two stages
                if (i < 2) ctp_wait_vm<3 + K_OPS>();
                else ctp_wait_vm<3>();
                __builtin_amdgcn_s_barrier();
                const int pbuf = buf == 0 ? 2 : buf - 1;
                if (c8 + 2 < C8) dma_stage(Htag, cur, c8 + 2, pbuf);
                else dma_stage(Hd{}, after, c8 + 2 - C8, pbuf);
                compute(Htag, buf);
                buf = buf == 2 ? 0 : buf + 1;
            }
            CTP_FENCE();
            finish(Hd{}, qd, drn, drn_valid);
            CTP_FENCE();
        }
    };

    // prologue: the queue's first two stages, and the S stores of a boundary that has nothing to finish
    dma_stage(std::integral_constant<int, 0>{}, cur, 0, 0);
    dma_stage(std::integral_constant<int, 0>{}, cur, 1, 1);
#pragma unroll
    for (int ai = 0; ai < 4; ++ai)
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            float z;
            asm volatile("v_mov_b32 %0, 0" : "=v"(z));
            acc[1][ai][r] = z;
        }
    CTP_FENCE();
    finish(std::integral_constant<int, 1>{}, 3, drn, drn_valid);   // all lanes outside: S stores that go nowhere

    for (unsigned t = tlo; t < thi; ++t) {
        nxt = t + 1 < thi ? decode(t + 1) : cur;
        phase(std::integral_constant<int, 0>{}, cur);     // drains half 1 of the tile before (nothing on the first tile)
        drn = cur;
        drn_valid = 1;
        phase(std::integral_constant<int, 1>{}, nxt);     // drains half 0 of this tile
        cur = nxt;
    }

    // the last tile's half 1: nothing left to overlap with
    __builtin_amdgcn_s_waitcnt(0x0f70);   // the two stages requested past the end
#pragma unroll
    for (int c = 0; c < 4; ++c) {
        CTP_FENCE();
        res_dma(std::integral_constant<int, 1>{}, c, drn, drn_valid);
        __builtin_amdgcn_s_waitcnt(0x0f70);
        CTP_FENCE();
        finish(std::integral_constant<int, 1>{}, c, drn, drn_valid);
    }
}
