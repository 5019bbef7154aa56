// Microbenchmark: the sweep's store stream at the reference-true shape (40 x 256 x 12 x 60 x 80, 16x8 tiles, 4 planes per block)
// in its own fp32 NCDHW pattern (64-byte runs per channel row) against the pattern an SCL-writing sweep would have (bf16 hi / mid
// pieces, 8 channels of a voxel = 16 bytes, a row of a tile = 256 contiguous bytes per channel octet and piece): the same
// bytes, the same blocks, stores only.  Question: how much higher is the ceiling of the second pattern?
// Build: hipcc -O3 --offload-arch=gfx950 tools/micro/store_pattern_scl.hip -o tools/micro/store_pattern_scl
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>

typedef float v4f __attribute__((ext_vector_type(4)));
typedef unsigned v4u __attribute__((ext_vector_type(4)));

constexpr int N = 40, C = 256, D = 12, H = 60, W = 80, TW = 16, TH = 8, DPB = 4;
constexpr int Dp = 14, Hp = 66, Wp = 82;   // ops.scl_geometry(40, 256, 12, 60, 80): one-voxel border + tile overhang

template <int MODE>   // 0: fp32 NCDHW (the shipped kernel's stores); 1: SCL, lane = (pixel pair, half of an octet's lanes); 2: SCL, one lane = 2 adjacent pixels
__global__ __launch_bounds__(256) void k(void* out, int tiles_x, int tiles) {
    const int S = C / 32;
    const int id = blockIdx.x, slab = id % S, bt = id / S, tile = bt % tiles, n = bt / tiles;
    const int tx0 = (tile % tiles_x) * TW, ty0 = (tile / tiles_x) * TH;
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6, g = lane & 7, ps = lane >> 3;
    const int p0 = 32 * wave + 4 * ps, px0 = tx0 + p0 % TW, py = ty0 + p0 / TW;
    if (py >= H || px0 + 4 > W) return;
    const int d0 = blockIdx.y * DPB, d1 = min(D, d0 + DPB);
    const v4f vv = {1.0f * id, 2.0f, 3.0f, (float)lane};
    const v4u uu = {(unsigned)id, (unsigned)lane, 3u, 4u};
    for (int d = d0; d < d1; ++d) {
        if (MODE == 0 || MODE == 6) {
            float* var = (float*)out;
#pragma unroll
            for (int i = 0; i < 4; ++i) {
                const int c = slab * 32 + 8 * i + g;
                v4f* dst = reinterpret_cast<v4f*>(var + (((size_t)n * C + c) * D + d) * (H * W) + (size_t)py * W + px0);
                if (MODE == 6) *dst = vv;   // plain stores: may the L2 merge the two tiles' halves of a 128-byte line?
                else __builtin_nontemporal_store(vv, dst);
            }
        } else {
            v4u* scl = (v4u*)out;
            const size_t piece = (size_t)N * (C / 8) * Dp * Hp * Wp;
            const int c8 = slab * 4 + (g >> 1);
            const size_t row = (((size_t)n * (C / 8) + c8) * Dp + (d + 1)) * Hp * Wp + (size_t)(py + 1) * Wp + (px0 + 1);
            if (MODE <= 3) {
#pragma unroll
                for (int j = 0; j < 2; ++j) {
                    // MODE 1: instruction j stores pixels 2j + (g & 1); MODE 2 / 3: the lane's pixels 2 (g & 1) + j
                    const int px = MODE == 1 ? 2 * j + (g & 1) : 2 * (g & 1) + j;
#pragma unroll
                    for (int p = 0; p < 2; ++p) {
                        if (MODE == 3) scl[p * piece + row + px] = uu;   // plain stores: the L2 may merge the halves of a line
                        else __builtin_nontemporal_store(uu, scl + p * piece + row + px);
                    }
                }
            } else {
                // MODE 4 / 5: what an ideal hand-over would allow -- the 8 lanes of a (row, octet) write 8 CONSECUTIVE units per
                // instruction (128 contiguous bytes), four instructions = the 16 pixels of the row in both pieces
                const int L = 2 * (ps & 3) + (g & 1);
                const size_t row0 = (((size_t)n * (C / 8) + c8) * Dp + (d + 1)) * Hp * Wp + (size_t)(py + 1) * Wp + (tx0 + 1);
#pragma unroll
                for (int t = 0; t < 4; ++t) {
                    v4u* dst = scl + (t >> 1) * piece + row0 + 8 * (t & 1) + L;
                    if (MODE == 5) *dst = uu;
                    else __builtin_nontemporal_store(uu, dst);
                }
            }
        }
    }
}

template <int MODE>
static void run(const char* name, void* buf) {
    const int tiles_x = (W + TW - 1) / TW, tiles = tiles_x * ((H + TH - 1) / TH);
    dim3 grid(N * tiles * (C / 32), (D + DPB - 1) / DPB);
    hipEvent_t e0, e1;
    hipEventCreate(&e0); hipEventCreate(&e1);
    for (int i = 0; i < 3; ++i) k<MODE><<<grid, 256>>>(buf, tiles_x, tiles);
    hipEventRecord(e0);
    for (int i = 0; i < 10; ++i) k<MODE><<<grid, 256>>>(buf, tiles_x, tiles);
    hipEventRecord(e1);
    hipEventSynchronize(e1);
    float ms;
    hipEventElapsedTime(&ms, e0, e1);
    ms /= 10;
    const double bytes = (double)N * C * D * H * W * 4;
    printf("%-44s %7.3f ms  %7.1f GB/s\n", name, ms, bytes / ms / 1e6);
}

int main() {
    void* buf;
    const size_t bytes = (size_t)2 * N * (C / 8) * Dp * Hp * Wp * 16 + (1 << 20);
    hipMalloc(&buf, bytes);
    hipMemset(buf, 0, bytes);
    for (int r = 0; r < 2; ++r) {
        run<0>("fp32 NCDHW (64-byte runs)", buf);
        run<6>("fp32 NCDHW, plain stores", buf);
        run<1>("SCL, an instruction = every second pixel pair", buf);
        run<2>("SCL, a lane's two pixels adjacent", buf);
        run<3>("SCL, adjacent pixels, plain stores", buf);
        run<4>("SCL, 128 contiguous bytes per 8 lanes", buf);
        run<5>("SCL, 128 contiguous bytes, plain stores", buf);
    }
    hipFree(buf);
    return 0;
}
