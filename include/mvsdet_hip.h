/*
 * mvsdet_hip.h -- C ABI of the MI355X (gfx950) plane-sweep hot path of MVSDet.
 *
 * This is the drop-in boundary.  The reference has no FFI on this path: it
 * is Python calling PyTorch operators.  Each entry point below replaces a
 * block of reference Python (file:line relative to projects/NeRF-Det/nerfdet/ of Pixie8888/MVSDet)
 * and is what a maintainer would bind from that Python with ctypes (INTEGRATION.md shows the
 * stub).  mvsdet_amd/_lib.py is exactly such a binding.
 *
 * Conventions
 *   - every pointer is a DEVICE pointer (hipMalloc / torch.cuda tensor storage) unless marked HOST;
 *   - tensors are fp32, dense row-major in the stated shape unless a stride array is given
 *     (strides are in ELEMENTS, HOST arrays of 4 int64);
 *   - `stream` is a hipStream_t (NULL = default stream); calls only enqueue work: no allocation, no
 *     host synchronisation, so they can be captured in a hipGraph;
 *   - return 0 on success, an MVSDET_ERR_* code otherwise; mvsdet_last_error() (HOST, thread-local)
 *     describes the failure.  Nothing is launched when an argument check fails.
 */
#ifndef MVSDET_HIP_H
#define MVSDET_HIP_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define MVSDET_OK 0
#define MVSDET_ERR_INVALID_ARG 1 /* bad shape / NULL pointer / unsupported size */
#define MVSDET_ERR_WORKSPACE 2   /* workspace too small */
#define MVSDET_ERR_HIP 3         /* a HIP runtime call failed (launch error) */

#define MVSDET_MAX_NEIGHBORS 4 /* k source views per reference view (reference: k = min(2, N-1)) */
#define MVSDET_MAX_TOPK 8      /* depth candidates per pixel (reference: topk = 3) */
#define MVSDET_MAX_DEPTH 512   /* depth planes (reference: 12; BASELINE configs go to 128) */

typedef void* mvsdet_stream_t; /* hipStream_t */

/* ABI version: major*1000 + minor. */
int mvsdet_version(void);
/* HOST, thread-local message of the last failing call on this thread ("" if none). */
const char* mvsdet_last_error(void);
/* Tuning options (schedule only: results are bit-identical under every setting).  Read ONCE from the environment
 * (MVSDET_SWEEP_TW, MVSDET_SWEEP_BOXCAP, MVSDET_SWEEP_XCD) when the library first needs them,
 * afterwards changed only through mvsdet_set_option -- the launch path never calls getenv.  HOST, process-global:
 * set them before launching from several threads.
 *   "sweep_tw"      0 (by the map shape) | 16 | 32   pixel-tile shape of the sweep (16x8 / 32x4)
 *   "sweep_boxcap"  texels of one LDS footprint box (default: what fits, 312 / 200 by the tile shape; 0 = gather every tap
 *                   from global memory)
 *   "sweep_xcd"     0 | 1   XCD-aware block map for fewer than 8 channel slabs
 *   "sweep_dsplit", "sweep_groups", "conv_*", "convT_cg", "convT_persist"   schedules of the sweep's plane split, of the bf16x3
 *                   convolutions and of the fp16 + MX convolution ("conv_mx_th") (csrc/common.h: struct Options);
 *                   "probe_f16_pair" shapes mvsdet_store_pattern_probe_f16 only
 * "sweep_tw" decides the layout of the sweep geometry: consume one (mvsdet_plane_sweep_variance_tabled_f32) under the
 * "sweep_tw" it was built with (mvsdet_plane_sweep_table_f32).  "sweep_boxcap" is baked into the geometry (union boxes,
 * staged / refill flags); the consuming call sizes its LDS slots for the largest capacity a geometry of that tile shape
 * can carry, so "sweep_boxcap" and "sweep_xcd" may change between the two calls. */
int mvsdet_set_option(const char* name /*HOST*/, int value);
int mvsdet_get_option(const char* name /*HOST*/, int* value /*HOST*/);
/* HOST check of neighbour view ids before they are uploaded (mvsdet.py:434 feeds them to an index gather, where an
 * id outside [0, n_src) raises): MVSDET_ERR_INVALID_ARG names the first offending entry.  The kernels additionally
 * clamp ids, so a tensor that never passed through this check cannot make them read outside the packed maps. */
int mvsdet_validate_neighbors(const int64_t* nbr /*HOST (M,K)*/, int M, int K, int n_src);

/* ---------------------------------------------------------------------------------------------
 * Packed feature maps.
 * The sweep and the fused voxel lifting read 2-D features channel-last so that one bilinear tap /
 * one voxel sample is a single contiguous run of all channels.  Packed layout:
 *     packed[n][s][y][x][q],  s in [0, S), S = ceil(C/32) channel slabs,  q in [0, 32),
 *     q = 4*g + i  <->  channel c = 32*s + 8*i + g      (g in [0,8), i in [0,4))
 * (zero where c >= C): one texel of one slab is one 128-byte line, and every slab of a view is a
 * contiguous H*W*128-byte image.  mvsdet_packed_bytes() = N*S*H*W*32*sizeof(float).
 * mvsdet_pack_features_f32 reads element (n,c,y,x) at feat[n*fs[0] + c*fs[1] + y*fs[2] + x*fs[3]],
 * so the reference's non-contiguous crop feature[:, :, :h, :w] (mvsdet.py:499) needs no copy.
 * ------------------------------------------------------------------------------------------- */
size_t mvsdet_packed_bytes(int N, int C, int H, int W);
int mvsdet_pack_features_f32(const float* feat, const int64_t* feat_strides /*HOST[4]*/, float* packed,
                             int N, int C, int H, int W, mvsdet_stream_t stream);
/* Same for IEEE binary16 feature maps (BASELINE configs[4], `--amp` style fp16 features): element strides of the
 * half tensor; the packed maps are fp32 (conversion is exact), so every consumer below is unchanged. */
int mvsdet_pack_features_f16(const void* feat_f16, const int64_t* feat_strides /*HOST[4]*/, float* packed,
                             int N, int C, int H, int W, mvsdet_stream_t stream);

/* ---------------------------------------------------------------------------------------------
 * a3  homo_warping -- mvs_models/module.py:105-146.
 *   src  (B,C,H,W)   source-view features
 *   proj (B,4,4)     src_proj @ inverse(ref_proj)  (module.py:116; computed by the caller with the
 *                    same torch ops as the reference -- see DESIGN.md "geometry stays on the host")
 *   depth(B,D)       plane depths
 *   out  (B,C,D,H,W) warped features: bilinear, zero padding, align_corners=False sampling of a grid
 *                    normalised with the align_corners=True formula (module.py:137-143), no z>0 test.
 * ------------------------------------------------------------------------------------------- */
int mvsdet_homo_warp_f32(const float* src, const float* proj, const float* depth, float* out,
                         int B, int C, int D, int H, int W, mvsdet_stream_t stream);

/* ---------------------------------------------------------------------------------------------
 * a3+a4  plane-sweep variance cost volume -- mvsdet.py:439-467 (k calls of homo_warping fused with
 * the running sum / sum of squares and the variance).
 *   packed (see above) features of all N views; nbr (N,K) int64 neighbour view ids (mvsdet.py:434);
 *   proj (N,K,4,4) = nei_proj[n][j] @ inverse(ref_proj[n]);  depth (N,D);
 *   var (N,C,D,H,W) = sum_sq/(K+1) - (sum/(K+1))^2 over {ref, warped_1..K}  (mvsdet.py:467).
 * `scratch` (16-byte aligned, >= mvsdet_plane_sweep_scratch_bytes(N,K,D,H,W)) receives the
 * channel-independent sweep geometry the sweep builds first: per (view, 128-pixel tile, plane,
 * neighbour) the footprint box that the sweep keeps resident in LDS -- shared by runs of
 * consecutive planes (sweep_kernel.h) --, one flags word per (view, tile, plane), and copies of
 * proj / depth: 20 B per (view, tile, plane) and neighbour, 0.03 % of the cost volume.  The sample
 * positions themselves are recomputed by the sweep (the reference builds its grid per plane too).
 * The _f32 form packs `feat` (N,C,H,W dense) into `workspace` first
 * (workspace_bytes >= mvsdet_plane_sweep_workspace_bytes(N,K,C,D,H,W) = packed + scratch).
 * ------------------------------------------------------------------------------------------- */
/* The pixel-tile shape (32x4 or 16x8) and the LDS box capacity in texels the sweep uses for this problem; the layout of
 * the geometry at the head of the scratch buffer follows from them: boxes int4[N][tiles][D][K], then flags
 * uint32[N][tiles][D], tiles = ceil(W / tile_w) * ceil(H / tile_h).  For tools and statistics (bench.py). */
int mvsdet_plane_sweep_tile_shape(int K, int D, int H, int W, int* tile_w, int* tile_h, int* box_texels);
size_t mvsdet_plane_sweep_scratch_bytes(int N, int K, int D, int H, int W);
size_t mvsdet_plane_sweep_workspace_bytes(int N, int K, int C, int D, int H, int W);
int mvsdet_plane_sweep_variance_packed_f32(const float* packed, const int64_t* nbr, const float* proj,
                                           const float* depth, float* var, void* scratch,
                                           size_t scratch_bytes, int N, int K, int C, int D, int H, int W,
                                           mvsdet_stream_t stream);
/* View shard of the same sweep (intra-scene split): the M reference views ref_first .. ref_first+M-1
 * of a scene whose N_src views are ALL in `packed` (neighbours are arbitrary views).  nbr (M,K) holds GLOBAL view
 * ids; nbr, proj (M,K,4,4), depth (M,D), var (M,C,D,H,W) and scratch (sized for M views) are indexed by the local
 * view.  Row m of the result is bit-identical to row ref_first+m of the unsharded call. */
int mvsdet_plane_sweep_variance_shard_f32(const float* packed, const int64_t* nbr, const float* proj,
                                          const float* depth, float* var, void* scratch, size_t scratch_bytes,
                                          int N_src, int ref_first, int M, int K, int C, int D, int H, int W,
                                          mvsdet_stream_t stream);
/* fp16 storage of the cost volume: identical fp32 arithmetic, var_f16 (M,C,D,H,W) IEEE binary16 = the fp32 result
 * rounded to nearest-even at the store (values above 65504 become +inf).  Halves the write stream, which is 96 % of
 * the sweep's bytes; with view shards it is how the 100-view / 128-plane / 240x320 configuration fits in HBM. */
int mvsdet_plane_sweep_variance_shard_f16(const float* packed, const int64_t* nbr, const float* proj,
                                          const float* depth, void* var_f16, void* scratch, size_t scratch_bytes,
                                          int N_src, int ref_first, int M, int K, int C, int D, int H, int W,
                                          mvsdet_stream_t stream);
/* The two halves of the call above, for callers that want to time / overlap / reuse them (the geometry of a scene can
 * be built once and swept with any number of feature sets):
 *   mvsdet_plane_sweep_table_f32           builds the sweep geometry of a scene's cameras into `scratch`;
 *   mvsdet_plane_sweep_variance_tabled_f32 runs the per-channel sweep on a geometry built by it for the SAME
 *                                          (N,K,D,H,W) (table_bytes = the scratch size passed there). */
int mvsdet_plane_sweep_table_f32(const float* proj, const float* depth, void* scratch, size_t scratch_bytes,
                                 int N, int K, int D, int H, int W, mvsdet_stream_t stream);
int mvsdet_plane_sweep_variance_tabled_f32(const float* packed, const int64_t* nbr, const void* table,
                                           size_t table_bytes, float* var, int N, int K, int C, int D, int H,
                                           int W, mvsdet_stream_t stream);
/* The same two calls for a PITCHED cost volume: `var` is (N,C,D,H,out_w_pitch) in memory and columns [0, W) of every row are
 * written (the caller hands out that view; mvsdet.py:467 materialises a contiguous volume -- same values, other strides).
 * With out_w_pitch a multiple of 32 every row starts on a 128-byte line and the sweep writes whole lines from 32x4 pixel
 * tiles even when W is no multiple of 32 (the 80-wide maps of the shipped configs: 64-byte runs otherwise).  The geometry
 * must be built with the pitch it is consumed with. */
int mvsdet_plane_sweep_table_pitched_f32(const float* proj, const float* depth, void* scratch, size_t scratch_bytes, int N,
                                         int K, int D, int H, int W, int out_w_pitch, mvsdet_stream_t stream);
int mvsdet_plane_sweep_variance_tabled_pitched_f32(const float* packed, const int64_t* nbr, const void* table,
                                                   size_t table_bytes, float* var, int N, int K, int C, int D, int H, int W,
                                                   int out_w_pitch, mvsdet_stream_t stream);
int mvsdet_plane_sweep_variance_f32(const float* feat, const int64_t* nbr, const float* proj,
                                    const float* depth, float* var, void* workspace, size_t workspace_bytes,
                                    int N, int K, int C, int D, int H, int W, mvsdet_stream_t stream);
/* backward of the above w.r.t. feat (the sampling grid carries no gradient, module.py:115).
 *   g (N,C,D,H,W) = dL/dvar;  gfeat (N,C,H,W) is OVERWRITTEN with dL/dfeat.
 *   workspace_bytes >= mvsdet_plane_sweep_bwd_workspace_bytes(N,K,C,D,H,W)
 *   (packed features + packed gradient + sweep geometry). */
size_t mvsdet_plane_sweep_bwd_workspace_bytes(int N, int K, int C, int D, int H, int W);
int mvsdet_plane_sweep_variance_bwd_f32(const float* feat, const int64_t* nbr, const float* proj,
                                        const float* depth, const float* g, float* gfeat, void* workspace,
                                        size_t workspace_bytes, int N, int K, int C, int D, int H, int W,
                                        mvsdet_stream_t stream);
/* The same from what the forward pass already made (a training step keeps them instead of making them again): `packed` =
 * mvsdet_pack_features_f32 of the features, `table` = the scratch buffer the forward call
 * (mvsdet_plane_sweep_variance_packed_f32 / mvsdet_plane_sweep_table_f32, same N, K, D, H, W and tile options; contiguous, not
 * pitched) left its sweep geometry in.  workspace >= mvsdet_packed_bytes(N,C,H,W) rounded up to 256 (the packed gradient map). */
int mvsdet_plane_sweep_variance_bwd_packed_f32(const float* packed, const int64_t* nbr, const void* table, size_t table_bytes,
                                               const float* g, float* gfeat, void* workspace, size_t workspace_bytes, int N, int K,
                                               int C, int D, int H, int W, mvsdet_stream_t stream);

/* ---------------------------------------------------------------------------------------------
 * a5-a7  depth probability, top-k plane selection, depth expectation --
 * mvsdet.py:470-475 (softmax / sigmoid), :266-283 (sample_depth_prob), :298-317 (compute_avg_depth).
 *   cost_reg, off_logit (N,D,H,W)  the two output channels of CostRegNet_3DGS
 *   prob, off (N,D,H,W)            softmax over D / sigmoid
 *   est_depth, est_dens (N,topk,H,W) the topk most probable planes, descending; depth =
 *                                  idx*interval + near + off[idx]*interval; dens = prob[idx] (raw)
 *   est_idx (N,topk,H,W) int32     chosen plane indices (may be NULL); ties -> lower index
 *   avg_depth (N,H,W)              sum_d prob_d * (d*interval + near + off_d*interval)
 * ------------------------------------------------------------------------------------------- */
int mvsdet_depth_prob_topk_f32(const float* cost_reg, const float* off_logit, float* prob, float* off,
                               float* est_depth, float* est_dens, int32_t* est_idx, float* avg_depth,
                               int N, int D, int H, int W, int topk, float near, float interval,
                               mvsdet_stream_t stream);
/* The same with the two inputs as channel slices of the network's (N, 2, D, H, W) output (mvsdet.py:469 torch.unbind):
 * view n of either input starts view_stride floats after view n-1, its (D, H, W) block is dense.  Saves the two copies
 * that making the slices contiguous costs. */
int mvsdet_depth_prob_topk_strided_f32(const float* cost_reg, const float* off_logit, long long view_stride, float* prob,
                                       float* off, float* est_depth, float* est_dens, int32_t* est_idx, float* avg_depth,
                                       int N, int D, int H, int W, int topk, float near, float interval,
                                       mvsdet_stream_t stream);
/* a6+a7 only, for callers that already hold prob = softmax(cost_reg) and off = sigmoid(off_logit)
 * (signature-level parity with MVSDet.sample_depth_prob / compute_avg_depth, mvsdet.py:266,298). */
int mvsdet_sample_depth_prob_f32(const float* prob, const float* off, float* est_depth, float* est_dens,
                                 int32_t* est_idx, float* avg_depth, int N, int D, int H, int W, int topk,
                                 float near, float interval, mvsdet_stream_t stream);
/* NVS-branch input -- mvsdet.py:1158-1216 compute_depth_scale / compute_depth_scale_MultiIntrin and :494:
 *   intr (N,5) = {fx, fy, cx, cy, skew} of the feature-level intrinsics of every view (K[:2] / ratio, mvsdet.py:1180-1181);
 *   depth_scale (N,h,w) = z component of the normalised camera ray through pixel (x, y) (lift :1300 + normalize :1295);
 *   est_ray_depth (N,J,h,w) = est_depth[:, :, :h, :w] / (depth_scale + 1e-8), est_depth (N,J,H,W) padded maps (both NULL to
 *   get the scale alone). */
int mvsdet_ray_depth_f32(const float* intr, const float* est_depth, float* depth_scale, float* est_ray_depth, int N, int J,
                         int H, int W, int h, int w, mvsdet_stream_t stream);
/* backward: any of g_prob (N,D,H,W), g_depth, g_dens (N,topk,H,W), g_avg (N,H,W) may be NULL. */
int mvsdet_depth_prob_topk_bwd_f32(const float* prob, const float* off, const int32_t* est_idx,
                                   const float* g_prob, const float* g_depth, const float* g_dens,
                                   const float* g_avg, float* g_cost, float* g_offlogit, int N, int D,
                                   int H, int W, int topk, float near, float interval, mvsdet_stream_t stream);

/* ---------------------------------------------------------------------------------------------
 * Head of the cost regularisation network (first step): mvs_models/mvsnet.py:102,112
 *   self.prob = nn.Conv3d(64, 2, 3, stride=1, padding=1);  x = self.prob(x)
 * x (N,Cin,D,H,W) dense fp32, weight (2,Cin,3,3,3), bias (2) or NULL -> out (N,2,D,H,W): the (cost, offset)
 * logits that mvsdet_depth_prob_topk_f32 consumes.  Forward only (training keeps the framework's convolution).
 * ------------------------------------------------------------------------------------------- */
/* The stride-1 ConvBnReLU3D layers of the same network, mvs_models/mvsnet.py:76,79,82 (Conv3d k=3, padding 1, no bias;
 * conv0 256->64, conv2 128->128, conv4 256->256):
 *   x (N,Cin,D,H,W) dense fp32 -> out (N,Cout,D,H,W), Cout a multiple of 64, on the fp32 matrix cores (exact fp32 FMA sums).
 *   weight_perm: the (Cout,Cin,3,3,3) weight permuted to [c][kd][kh][kw][o] (Cin rounded up to even, zero padded),
 *   16-byte aligned.  scale / shift (Cout each, or both NULL): out = v*scale[o] + shift[o] -- eval-mode BatchNorm folded;
 *   relu != 0 clamps at 0.  Forward only. */
int mvsdet_conv3d_k3_mfma_f32(const float* x, const float* weight_perm, const float* scale, const float* shift, float* out,
                              int N, int Cin, int Cout, int D, int H, int W, int relu, mvsdet_stream_t stream);
/* The same convolution with a residual added between the affine and the ReLU: out = act(v*scale[o] + shift[o] + residual)
 * -- the tail of the 3-D neck's ResModule, mmdet3d/models/necks/imvoxel_neck.py:219-230 (x = conv1(conv0(x)); x = x +
 * identity; x = relu(x)).  residual: shape of out, or NULL. */
int mvsdet_conv3d_k3_res_mfma_f32(const float* x, const float* weight_perm, const float* scale, const float* shift,
                                  const float* residual, float* out, int N, int Cin, int Cout, int D, int H, int W, int relu,
                                  mvsdet_stream_t stream);
/* The stride-2 layers (mvsnet.py:78,81: conv1 64->128, conv3 128->256): same arguments, x (N,Cin,D,H,W) ->
 * out (N,Cout,(D-1)/2+1,(H-1)/2+1,(W-1)/2+1). */
int mvsdet_conv3d_k3_s2_mfma_f32(const float* x, const float* weight_perm, const float* scale, const float* shift,
                                 float* out, int N, int Cin, int Cout, int D, int H, int W, int relu, mvsdet_stream_t stream);
/* The general form of the three entry points above, for small volumes (mmdet3d/models/necks/imvoxel_neck.py:183-231, the
 * 3-D neck's residual blocks at one scene, and nerfdet_head.py:96-101, the head's convolutions: 400 / 50 / 8 tiles of
 * 256 voxels): with a workspace of mvsdet_conv3d_k3_mfma_workspace_bytes(...) bytes (0 = the grid fills the chip unsplit)
 * the input-channel loop is split over several blocks per tile, which write raw partial sums; a second kernel adds them in
 * ascending split order and applies affine, residual (stride 1 only, NULL = none) and ReLU.  workspace NULL or too small:
 * unsplit.  stride 1 or 2. */
size_t mvsdet_conv3d_k3_mfma_workspace_bytes(int N, int Cin, int Cout, int D, int H, int W, int stride);
int mvsdet_conv3d_k3_mfma_ws_f32(const float* x, const float* weight_perm, const float* scale, const float* shift,
                                 const float* residual, float* out, void* workspace, size_t workspace_bytes, int N, int Cin,
                                 int Cout, int D, int H, int W, int stride, int relu, mvsdet_stream_t stream);
/* The up-sampling layers (mvsnet.py:92-100,110-111: conv9 256->128, conv11 128->64): ConvTranspose3d(kernel 3, stride 2,
 * padding 1, output_padding 1, no bias) [+ affine + ReLU] [+ residual]: x (N,Cin,D,H,W) -> out (N,Cout,2D,2H,2W);
 * weight_perm: the (Cin,Cout,3,3,3) weight permuted to [c][kd][kh][kw][o]; residual (shape of out, or NULL) is added
 * AFTER the affine + ReLU (x = skip + conv(x)).  Four launches, one per output parity in (d, h); a lane writes the two w
 * parities as one float2 (out and residual 8-byte aligned). */
int mvsdet_convT3d_k3_s2_mfma_f32(const float* x, const float* weight_perm, const float* scale, const float* shift,
                                  const float* residual, float* out, int N, int Cin, int Cout, int D, int H, int W, int relu,
                                  mvsdet_stream_t stream);
/* Weight gradient of the stride-1 layers (training): dW[o][c][tap] = sum_voxels grad_out[n][o][v] * x[n][c][v + tap - 1]
 * on the fp32 matrix cores.  partial (nsplit,Cout,Cin,27) fp32 receives one partial sum per voxel split
 * (mvsdet_conv3d_k3_dw_partial_bytes); the caller adds them up (deterministic, no atomics).  The input gradient of
 * these layers is mvsdet_conv3d_k3_mfma_f32 on grad_out with the weights transposed and flipped. */
size_t mvsdet_conv3d_k3_dw_partial_bytes(int Cin, int Cout, int nsplit);
int mvsdet_conv3d_k3_dw_mfma_f32(const float* x, const float* grad_out, float* partial, size_t partial_bytes, int nsplit,
                                 int N, int Cin, int Cout, int D, int H, int W, mvsdet_stream_t stream);
/* The same sum on the bf16 matrix cores with three-term split operands (the arithmetic of mvsdet_conv3d_k3_bf16x3: both
 * fp32 tensors are cut into bf16 pieces on the way into the LDS, products accumulate in fp32; relative error of a product
 * 2^-16).  Same arguments and partial layout. */
int mvsdet_conv3d_k3_dw_bf16x3(const float* x, const float* grad_out, float* partial, size_t partial_bytes, int nsplit,
                               int N, int Cin, int Cout, int D, int H, int W, mvsdet_stream_t stream);
/* The same for the stride-2 layers (mvsnet.py:77,80: conv1, conv3): x (N,Cin,D,H,W) with even D, H, W, grad_out
 * (N,Cout,D/2,H/2,W/2); dW[o][c][tap] = sum grad_out[n][o][v] * x[n][c][2v + tap - 1].  With the two tensors exchanged --
 * x := the layer's grad_out (fine), grad_out := the layer's input (coarse) -- the result is the (Cin,Cout,3,3,3) weight
 * gradient of the transposed layers (mvsnet.py:92-100: conv9, conv11).  Their input gradients are each other's forward:
 * dX of a stride-2 layer = mvsdet_convT3d_k3_s2_mfma_f32 on grad_out with the layer's own (Cout,Cin,3,3,3) weight read
 * as a transposed-convolution weight, dX of a transposed layer = mvsdet_conv3d_k3_s2_mfma_f32 likewise. */
int mvsdet_conv3d_k3_s2_dw_mfma_f32(const float* x, const float* grad_out, float* partial, size_t partial_bytes, int nsplit,
                                    int N, int Cin, int Cout, int D, int H, int W, mvsdet_stream_t stream);
/* The same sum (either orientation) on the bf16 matrix cores with three-term split operands (csrc/costreg_dw_s2_bf16.hip:
 * blocks of 64 coarse x 16 fine channels on v_mfma_f32_16x16x32_bf16; the fine rows are de-interleaved along w while they
 * are cut into bf16 pieces, so that every tap's fragment is one aligned LDS read).  Same arguments and partial layout;
 * W a multiple of 8 (the rows of both tensors are read as float4), D and H even. */
int mvsdet_conv3d_k3_s2_dw_bf16x3(const float* x, const float* grad_out, float* partial, size_t partial_bytes, int nsplit,
                                  int N, int Cin, int Cout, int D, int H, int W, mvsdet_stream_t stream);
/* Training-mode BatchNorm3d [+ ReLU] of the network's ConvBnReLU3D / Sequential(ConvTranspose3d, BatchNorm3d, ReLU) blocks
 * (mvs_models/module.py:26-37, mvsnet.py:92-100): batch statistics over (N, D, H, W) per channel (biased variance),
 * running statistics updated in place with `momentum` (unbiased variance, as torch.nn.BatchNorm3d; NULL = not tracked),
 * out = [relu](gamma * (x - mean) * invstd + beta) (gamma / beta NULL = 1 / 0); save_mean / save_invstd (C floats) are what
 * the backward needs.  x and out are (N, C, vol) contiguous, vol = D*H*W.  workspace: mvsdet_bn3d_workspace_bytes(C).
 * Backward: grad_x, grad_gamma, grad_beta (the last two may be NULL) from x, grad_out and the saved statistics; the ReLU
 * mask is recomputed from x with the forward's arithmetic. */
size_t mvsdet_bn3d_workspace_bytes(int C);
int mvsdet_bn3d_relu_train_fwd_f32(const float* x, const float* gamma, const float* beta, float* running_mean,
                                   float* running_var, float* out, float* save_mean, float* save_invstd, void* workspace,
                                   size_t workspace_bytes, int N, int C, long long vol, float momentum, float eps, int relu,
                                   mvsdet_stream_t stream);
/* The same with a residual tensor (N, C, vol; NULL = none) added AFTER the activation: out = [relu](bn(x)) + residual, the
 * `skip + Sequential(ConvTranspose3d, BatchNorm3d, ReLU)(x)` of mvsnet.py:109-111 without a pass of its own.  The backward is
 * unchanged (the residual's gradient is grad_out itself). */
int mvsdet_bn3d_relu_train_fwd_res_f32(const float* x, const float* gamma, const float* beta, const float* residual,
                                       float* running_mean, float* running_var, float* out, float* save_mean, float* save_invstd,
                                       void* workspace, size_t workspace_bytes, int N, int C, long long vol, float momentum, float eps,
                                       int relu, mvsdet_stream_t stream);
/* The same with the statistics taken from partial sums a producing convolution left (mvsdet_conv3d_k3_bf16x3_stats): partial[c * parts
 * + i] = (sum, sum of squares) of (x - pivot[c]) of channel c over block i, double2; pivot as given to the producer (NULL = zeros);
 * the entries are added in a fixed order. */
int mvsdet_bn3d_relu_train_fwd_parts_f32(const float* x, const void* partial, size_t parts, const float* pivot, const float* gamma, const float* beta,
                                         const float* residual, float* running_mean, float* running_var, float* out, float* save_mean,
                                         float* save_invstd, void* workspace, size_t workspace_bytes, int N, int C, long long vol,
                                         float momentum, float eps, int relu, mvsdet_stream_t stream);
int mvsdet_bn3d_relu_bwd_f32(const float* x, const float* grad_out, const float* gamma, const float* beta,
                             const float* save_mean, const float* save_invstd, float* grad_x, float* grad_gamma,
                             float* grad_beta, void* workspace, size_t workspace_bytes, int N, int C, long long vol, int relu,
                             mvsdet_stream_t stream);
int mvsdet_conv3d_k3_cout2_f32(const float* x, const float* weight, const float* bias, float* out, int N, int Cin,
                               int D, int H, int W, mvsdet_stream_t stream);
/* The same layer on the SUM of two tensors, x + x2 (x2 NULL: x alone), formed while the halo tiles are staged: mvsnet.py:111-112
 * `x = conv0 + self.conv11(x); x = self.prob(x)` without the skip addition in the transposed layer's epilogue (which then is a
 * pure store stream).  A second input needs W % 4 == 0 and 16-byte aligned tensors. */
int mvsdet_conv3d_k3_cout2_sum_f32(const float* x, const float* x2, const float* weight, const float* bias, float* out, int N,
                                   int Cin, int D, int H, int W, mvsdet_stream_t stream);

/* Backward of the head (training): grad_x (N,Cin,D,H,W) from grad_out (N,2,D,H,W) and weight (2,Cin,3,3,3); and the
 * weight gradient as partial (nsplit,2,Cin,27) sums (one per voxel split, added up by the caller). */
int mvsdet_conv3d_k3_cout2_dx_f32(const float* grad_out, const float* weight, float* grad_x, int N, int Cin, int D, int H,
                                  int W, mvsdet_stream_t stream);
int mvsdet_conv3d_k3_cout2_dw_f32(const float* x, const float* grad_out, float* partial, size_t partial_bytes, int nsplit,
                                  int N, int Cin, int D, int H, int W, mvsdet_stream_t stream);
/* The weight gradient on the bf16 matrix cores with three-term split operands (within ~1e-5 of the fp32 sums' scale): x streams
 * from global memory straight into the MFMA fragments, grad_out's 3 x 3 neighbouring rows are staged per x row.  nsplit = blocks =
 * rows of partial (nsplit,2,Cin,27); Cin in {16, 32, 64}, W % 4 == 0, 16-byte aligned tensors (else MVSDET_ERR_INVALID_ARG: use
 * the fp32 call). */
int mvsdet_conv3d_k3_cout2_dw_bf16x3(const float* x, const float* grad_out, float* partial, size_t partial_bytes, int nsplit,
                                     int N, int Cin, int D, int H, int W, mvsdet_stream_t stream);
/* 1 if the call above takes this (Cin, W) -- channel count, row width and the LDS of a row stage (W up to ~470 at Cin = 64) --,
 * else 0: then mvsdet_conv3d_k3_cout2_dw_f32 is the entry point to use. */
int mvsdet_conv3d_k3_cout2_dw_bf16x3_ok(int Cin, int W);

/* ---------------------------------------------------------------------------------------------
 * a9  backproject_Weigh -- mvsdet.py:1372-1492 (gt_depth=None).
 *   feat + feat_strides: (N,C,h,w) view of the 2-D features (crop allowed, see pack above)
 *   points (3,V) voxel coordinates from get_points (mvsdet.py:1316); projection (N,3,4)
 *   depth, dens + dd_strides: candidate j of pixel (y,x) of view i at
 *       ptr[i*s[0] + j*s[1] + y*s[2] + x*s[3]]   (the reference's (N,h*w,1,J) tensors are views of this)
 *   vz = voxel_size[-1]
 *   volume (N,C,V) fp32; valid (N,V) uint8 (1 = voxel is in view i's frustum AND inside the depth
 *   window of at least one candidate).  xi, yi (N,V) int32 rounded pixel coordinates may be NULL.
 * ------------------------------------------------------------------------------------------- */
int mvsdet_backproject_weigh_f32(const float* feat, const int64_t* feat_strides /*HOST[4]*/, const float* points,
                                 const float* projection, const float* depth, const float* dens,
                                 const int64_t* dd_strides /*HOST[4]*/, float* volume, uint8_t* valid,
                                 int32_t* xi, int32_t* yi, int N, int C, int h, int w, int V, int J, float vz,
                                 mvsdet_stream_t stream);
/* a9+a10 fused: per-voxel mean over views, mvsdet.py:511-515.
 *   packed: packed features of the FULL (N,C,H,W) maps; only pixels y<h, x<w are addressed.
 *   mean (C,V) fp32 = sum_i volume_i / (count + 1e-8), 0 where count == 0; count (V) int32. */
int mvsdet_backproject_weigh_mean_packed_f32(const float* packed, const float* points, const float* projection,
                                             const float* depth, const float* dens,
                                             const int64_t* dd_strides /*HOST[4]*/, float* mean, int32_t* count,
                                             int N, int C, int H, int W, int h, int w, int V, int J, float vz,
                                             mvsdet_stream_t stream);
/* Same kernel without the division: sum (C,V) = sum_i volume_i over the N views handed in, count (V).  A rank that
 * owns a contiguous shard of the views passes packed / projection / depth / dens offset to its first view; the
 * ranks' sums and counts are then all-reduced and divided once (mvsdet_amd/parallel.py). */
int mvsdet_backproject_weigh_sum_packed_f32(const float* packed, const float* points, const float* projection,
                                            const float* depth, const float* dens,
                                            const int64_t* dd_strides /*HOST[4]*/, float* sum, int32_t* count,
                                            int N, int C, int H, int W, int h, int w, int V, int J, float vz,
                                            mvsdet_stream_t stream);
/* backward of a9 w.r.t. features and dens (depth carries no gradient).
 *   g (N,C,V); gfeat (N,C,h,w) dense and gdens (N,J,h,w) dense are OVERWRITTEN. */
int mvsdet_backproject_weigh_bwd_f32(const float* feat, const int64_t* feat_strides /*HOST[4]*/, const float* points,
                                     const float* projection, const float* depth, const float* dens,
                                     const int64_t* dd_strides /*HOST[4]*/, const float* g, float* gfeat,
                                     float* gdens, int N, int C, int h, int w, int V, int J, float vz,
                                     mvsdet_stream_t stream);
/* backward of a9+a10 (fused mean): g (C,V) = dL/dmean. */
int mvsdet_backproject_weigh_mean_bwd_f32(const float* feat, const int64_t* feat_strides /*HOST[4]*/,
                                          const float* points, const float* projection, const float* depth,
                                          const float* dens, const int64_t* dd_strides /*HOST[4]*/,
                                          const int32_t* count, const float* g, float* gfeat, float* gdens,
                                          int N, int C, int h, int w, int V, int J, float vz,
                                          mvsdet_stream_t stream);

/* ---------------------------------------------------------------------------------------------
 * The cost network's 3x3x3 convolutions on the bf16 matrix cores, fp32 operands cut into bf16 pieces ("bf16x3":
 * x*w ~ x_hi*w_hi + x_hi*w_mid + x_mid*w_hi, fp32 accumulation; csrc/costreg_bf16.hip).  Replaces the same layers of
 * mvs_models/mvsnet.py:76-82,104-108 as mvsdet_conv3d_k3_mfma_f32 does, at 3/16 of the fp32 matrix-core time; results
 * differ from an fp32 convolution by ~2^-16 of the products' size (logits of the network: 2e-6 .. 4e-6, bar 1e-4).
 *
 * "SCL" (split channel-last) activations: xs[piece 2][n][c8 = ceil(C/8)][Dp][Hp][Wp][8] bf16, piece = hi | mid,
 * (dp,hp,wp) = (d,h,w) + 1 inside a zero border; Dp/Hp/Wp (the border plus the largest tile overhang of any kernel that
 * reads the form) and the size in bytes come from mvsdet_scl_bytes.
 * mvsdet_scl_pack_f32 cuts an (N,C,D,H,W) fp32 tensor (possibly row-pitched: mvsdet_plane_sweep_variance_tabled_pitched_f32; zero_border 1: clear the buffer first -- a buffer reused for the
 * same shape needs that once; 2: the packing kernel runs over the padded volume and writes the border's zeros itself -- for a
 * buffer fresh from an allocator on every call, no clearing pass).  weight_split: [Cout/64][c8][14 tap pairs][2 groups of 32 outputs][2 pieces][64 lanes][8]
 * bf16, lane = 32 * (tap parity) + MFMA row m, row m carrying output channel 8*((m>>4)*2 + ((m>>2)&1)) + (m&3) + 4*((m>>3)&1)
 * of its group (so that a lane's accumulator registers 8q..8q+7 are eight consecutive channels = one SCL unit), tap 27 and
 * channels >= Cin zero (mvsdet_amd.ops.split_conv_weight).
 * out (N,Cout,D,H,W) fp32 = [relu]([scale *] conv [+ shift] [+ residual]).  Stride 1, Cout % 64 == 0.
 * ------------------------------------------------------------------------------------------- */
size_t mvsdet_scl_bytes(int N, int C, int D, int H, int W, int* Dp /*HOST, may be NULL*/, int* Hp, int* Wp);
/* weight (Cout = 64*m, Cin, 3,3,3) fp32 -> weight_split (mvsdet_split_conv_weight_bytes) on the device: one small launch */
size_t mvsdet_split_conv_weight_bytes(int Cout, int Cin);
int mvsdet_split_conv_weight(const float* weight, void* weight_split, int Cout, int Cin, mvsdet_stream_t stream);
/* order 0 = the stride-1 layout above; 1 = for mvsdet_conv3d_k3_s2_bf16x3_f32in (tap pairs grouped by the parity class of the
 * input voxel); 2 = for the transposed convolution: `weight` is a ConvTranspose3d weight (Cin,Cout,3,3,3), pairs grouped by the
 * parity class of the output voxel. */
int mvsdet_split_conv_weight_ordered(const float* weight, void* weight_split, int Cout, int Cin, int order, mvsdet_stream_t stream);
/* up to 8 weight tensors in ONE launch (a network's layers: run per forward call, so in-place weight updates are always seen);
 * weights / weight_splits: HOST arrays of `count` device pointers; Cout / Cin / orders: HOST arrays */
int mvsdet_split_conv_weights_batched(const float* const* weights, void* const* weight_splits, const int* Cout, const int* Cin,
                                      const int* orders, int count, mvsdet_stream_t stream);
int mvsdet_scl_pack_f32(const float* x, const int64_t* x_strides /*HOST[4] = element strides of n, c, d, h; w stride 1; NULL = contiguous*/,
                        void* xs, int N, int C, int D, int H, int W, int zero_border, mvsdet_stream_t stream);
int mvsdet_conv3d_k3_bf16x3(const void* xs, const void* weight_split, const float* scale, const float* shift,
                            const float* residual, float* out, int N, int Cin, int Cout, int D, int H, int W, int relu,
                            mvsdet_stream_t stream);
/* The same convolution on the fp32 tensor itself (x_strides as for mvsdet_scl_pack_f32): the pieces are cut inside the
 * kernel while the previous channel group is multiplied -- no packing pass, no SCL copy; results identical bit for bit. */
int mvsdet_conv3d_k3_bf16x3_f32in(const float* x, const int64_t* x_strides /*HOST[4], NULL = contiguous*/, const void* weight_split,
                                  const float* scale, const float* shift, const float* residual, float* out, int N, int Cin,
                                  int Cout, int D, int H, int W, int relu, mvsdet_stream_t stream);
/* Small volumes (the 3-D neck's 20x20x8 and 10x10x4 levels: a handful of tiles): the input channels are split over blocks so
 * that the grid fills the chip, raw partial sums go to `workspace` and a second kernel adds them (ascending split order)
 * before the affine, the residual and the ReLU.  workspace_bytes from mvsdet_conv3d_k3_bf16x3_workspace_bytes (0: the grid
 * is large enough unsplit); NULL or too small a workspace: unsplit, same results up to the order of the fp32 sums. */
size_t mvsdet_conv3d_k3_bf16x3_workspace_bytes(int N, int Cin, int Cout, int D, int H, int W);
int mvsdet_conv3d_k3_bf16x3_ws(const void* xs, const void* weight_split, const float* scale, const float* shift,
                               const float* residual, float* out, void* workspace, size_t workspace_bytes, int N, int Cin, int Cout,
                               int D, int H, int W, int relu, mvsdet_stream_t stream);
int mvsdet_conv3d_k3_bf16x3_f32in_ws(const float* x, const int64_t* x_strides /*HOST[4], NULL = contiguous*/, const void* weight_split,
                                     const float* scale, const float* shift, const float* residual, float* out, void* workspace,
                                     size_t workspace_bytes, int N, int Cin, int Cout, int D, int H, int W, int relu,
                                     mvsdet_stream_t stream);

/* Conv3d(kernel 3, stride 2, padding 1, no bias) [+ affine] [+ ReLU] (mvsnet.py:77,79) on the bf16 matrix cores, three-term
 * split: x (N,Cin,D,H,W) fp32 -> out (N,Cout,(D-1)/2+1,(H-1)/2+1,(W-1)/2+1); weight_split of order 1. */
int mvsdet_conv3d_k3_s2_bf16x3_f32in(const float* x, const int64_t* x_strides /*HOST[4], NULL = contiguous*/, const void* weight_split,
                                     const float* scale, const float* shift, float* out, int N, int Cin, int Cout, int D, int H,
                                     int W, int relu, mvsdet_stream_t stream);
/* The stride-2 layer with a workspace for small volumes (as mvsdet_conv3d_k3_bf16x3_f32in_ws). */
size_t mvsdet_conv3d_k3_s2_bf16x3_workspace_bytes(int N, int Cin, int Cout, int D, int H, int W);
int mvsdet_conv3d_k3_s2_bf16x3_f32in_ws(const float* x, const int64_t* x_strides /*HOST[4], NULL = contiguous*/, const void* weight_split,
                                        const float* scale, const float* shift, float* out, void* workspace, size_t workspace_bytes,
                                        int N, int Cin, int Cout, int D, int H, int W, int relu, mvsdet_stream_t stream);

/* ConvTranspose3d(kernel 3, stride 2, padding 1, output_padding 1, no bias) [+ affine] [+ ReLU] [+ residual, added last]
 * (mvsnet.py:92-100,110-111) on the bf16 matrix cores, three-term split: xs = SCL form of x (N,Cin,D,H,W); weight_split of
 * order 2 from the (Cin,Cout,3,3,3) weight; out / residual (N,Cout,2D,2H,2W) fp32, 8-byte aligned. */
int mvsdet_convT3d_k3_s2_bf16x3(const void* xs, const void* weight_split, const float* scale, const float* shift,
                                const float* residual, float* out, int N, int Cin, int Cout, int D, int H, int W, int relu,
                                mvsdet_stream_t stream);

/* ---------------------------------------------------------------------------------------------
 * Layer-to-layer forms (round 4): a convolution can leave its result ALREADY CUT into bf16 pieces for the next one -- no fp32
 * round trip through a packing pass, the consumer feeds its LDS by DMA, and the producer issues one 16-byte store per piece
 * and 8 channels instead of eight 4-byte ones.  Chain of mvs_models/mvsnet.py:104-112:
 *   conv0 -(out_f32: skip of conv11, out_pscl)-> conv1 -(out_scl)-> conv2 -(out_f32: skip of conv9, out_pscl)-> conv3
 *   -(out_scl)-> conv4 -(out_scl)-> conv9 -(out_scl)-> conv11 -(out_f32)-> prob.
 * "PSCL" (parity-split SCL) of an (N,C,D,H,W) tensor: [piece 2][class 8][n][c8][cDp][cHp][cWp][8] bf16, class = 4*(d&1) +
 * 2*(h&1) + (w&1), voxel (d,h,w) at index (d/2+1, h/2+1, w/2+1) of its class, zero elsewhere: the eight stride-1 grids a
 * stride-2 convolution reads (mvsdet_pscl_bytes).  Output buffers' borders must be zero before the call (cleared once per
 * buffer: the kernels write interior voxels only); any of out_f32 / out_scl / out_pscl may be NULL.
 * ------------------------------------------------------------------------------------------- */
size_t mvsdet_pscl_bytes(int N, int C, int D, int H, int W, int* cDp /*HOST, may be NULL*/, int* cHp, int* cWp);
/* stride 1: input = xs (SCL) or x (fp32 + x_strides), exactly one of them; workspace as mvsdet_conv3d_k3_bf16x3_ws, used
 * only when out_f32 is the sole output */
int mvsdet_conv3d_k3_bf16x3_io(const void* xs, const float* x, const int64_t* x_strides /*HOST[4], NULL = contiguous*/,
                               const void* weight_split, const float* scale, const float* shift, const float* residual,
                               float* out_f32, void* out_scl, void* out_pscl, void* workspace, size_t workspace_bytes, int N,
                               int Cin, int Cout, int D, int H, int W, int relu, mvsdet_stream_t stream);
/* stride 1 on ONE fp16 and TWO block-scaled FP6 (OCP MX e2m3) products per fp32-equivalent product instead of three bf16 ones
 * (csrc/costreg_mx.h; v_mfma_f32_16x16x32_f16 + v_mfma_scale_f32_16x16x128_f8f6f4): the layer that reads the fp32 variance volume in
 * place, mvs_models/mvsnet.py:76 (conv0).  x, x_strides, scale / shift / relu and the three outputs as for
 * mvsdet_conv3d_k3_bf16x3_io (no residual, no split over the input channels); weight_split_mx from mvsdet_split_conv_weight_mx
 * (mvsdet_split_conv_weight_mx_bytes).  Values within ~2^-15 relative of the exact convolution (bf16x3: 2^-16; whole-network logits
 * 1-2e-5 from float64, the bar is 1e-4).  Inputs beyond fp16's range are cut on a
 * block-uniform power of two (no cliff at 65504). */
size_t mvsdet_split_conv_weight_mx_bytes(int Cout, int Cin);   /* 0 unless Cout is a positive multiple of 64 */
int mvsdet_split_conv_weight_mx(const float* weight, void* weight_split_mx, int Cout, int Cin, mvsdet_stream_t stream);
int mvsdet_conv3d_k3_fp16mx_f32in(const float* x, const int64_t* x_strides /*HOST[4], NULL = contiguous*/, const void* weight_split_mx,
                                  const float* scale, const float* shift, float* out_f32, void* out_scl, void* out_pscl, int N, int Cin,
                                  int Cout, int D, int H, int W, int relu, mvsdet_stream_t stream);
/* stride 1 in front of a training-mode BatchNorm (module.py:26-37 ConvBnReLU3D under model.train()): the raw fp32 output plus, from
 * the kernel's epilogue, per-channel partial sums of the outputs and of their squares -- stats[c * parts + i] = double2 of channel c
 * in block i of the grid, parts = mvsdet_conv3d_k3_bf16x3_stats_parts(N, D, H, W, x != NULL); stats_bytes >= Cout * parts * 16.
 * The sums are those of (value - pivot[c]) (pivot: Cout floats near the channels' means, e.g. the BatchNorm's running mean; NULL =
 * zeros): fp32 lane sums of raw values cancel when |mean| >> spread.  mvsdet_bn3d_relu_train_fwd_parts_f32 (same pivot) finishes
 * them: the BatchNorm reads the tensor once instead of twice. */
size_t mvsdet_conv3d_k3_bf16x3_stats_parts(int N, int D, int H, int W, int f32_input);
int mvsdet_conv3d_k3_bf16x3_stats(const void* xs, const float* x, const int64_t* x_strides /*HOST[4], NULL = contiguous*/,
                                  const void* weight_split, float* out_f32, void* stats, size_t stats_bytes, const float* pivot, int N,
                                  int Cin, int Cout, int D, int H, int W, mvsdet_stream_t stream);
/* ... and of the transposed layer (mvsnet.py:92-100): xs = SCL form of the coarse (N,Cin,D,H,W) input, out_f32 (N,Cout,2D,2H,2W) raw;
 * parts = mvsdet_convT3d_k3_s2_bf16x3_stats_parts(N, D, H, W), 0 where the shape has no statistics form (use the plain call then). */
size_t mvsdet_convT3d_k3_s2_bf16x3_stats_parts(int N, int D, int H, int W);
int mvsdet_convT3d_k3_s2_bf16x3_stats(const void* xs, const void* weight_split, float* out_f32, void* stats, size_t stats_bytes,
                                      const float* pivot, int N, int Cin, int Cout, int D, int H, int W, mvsdet_stream_t stream);
/* stride 2: input = x (fp32 + x_strides) or x_pscl (the PSCL form of the (N,Cin,D,H,W) input), exactly one of them */
int mvsdet_conv3d_k3_s2_bf16x3_io(const float* x, const int64_t* x_strides /*HOST[4], NULL = contiguous*/, const void* x_pscl,
                                  const void* weight_split, const float* scale, const float* shift, float* out_f32,
                                  void* out_scl, void* workspace, size_t workspace_bytes, int N, int Cin, int Cout, int D, int H,
                                  int W, int relu, mvsdet_stream_t stream);
/* transposed: out_scl = SCL form of the (N,Cout,2D,2H,2W) result */
int mvsdet_convT3d_k3_s2_bf16x3_io(const void* xs, const void* weight_split, const float* scale, const float* shift,
                                   const float* residual, float* out_f32, void* out_scl, int N, int Cin, int Cout, int D, int H,
                                   int W, int relu, mvsdet_stream_t stream);

/* ---------------------------------------------------------------------------------------------
 * Measurement helper for bench.py: runs `fn`-independent HIP-event timing is done by the caller;
 * this only exposes a device-to-device float4 copy so the achievable HBM ceiling can be calibrated
 * on the same box (SURVEY.md section 8d "Roofline").
 * ------------------------------------------------------------------------------------------- */
int mvsdet_copy_f32(const float* src, float* dst, size_t n_floats, mvsdet_stream_t stream);

/* The plane sweep's store stream alone: the block -> address map of the fused variance kernel (mvsdet.py:467's volume,
 * (N,C,D,H,out_w_pitch) fp32; C % 32 == 0, W % 4 == 0), tile width 16 or 32, `planes_per_block` planes per block (0 = all),
 * non-temporal 16-byte stores and nothing else.  bench.py times it to report what the output LAYOUT allows on the box at hand
 * (`frac_of_store_pattern_ceiling`); `var` is overwritten with arbitrary values. */
int mvsdet_store_pattern_probe_f32(float* var, int N, int C, int D, int H, int W, int out_w_pitch, int tile_w,
                                   int planes_per_block, mvsdet_stream_t stream);
/* The same for fp16 storage (BASELINE configs[4]): a lane's four pixels are 8 bytes, a wave-instruction writes 8 channel rows of
 * 64 bytes -- half a 128-byte line per row.  Option "probe_f16_pair" = 1: lanes of adjacent pixel quads store 16 bytes of two
 * channel rows instead (an experiment; the sweep's own flush is the unpaired pattern). */
int mvsdet_store_pattern_probe_f16(void* var, int N, int C, int D, int H, int W, int out_w_pitch, int tile_w,
                                   int planes_per_block, mvsdet_stream_t stream);

/* ---------------------------------------------------------------------------------------------
 * The 3-D neck's GEMM-shaped layers (mmdet3d/models/necks/imvoxel_neck.py:166-180, 196-217) on the bf16 matrix cores with
 * three-term split operands, bias / ReLU / the 2x2x2 interleave in the epilogue (csrc/neck_gemm.hip).
 * wsplit = mvsdet_gemm_split_weight of the (M, K) fp32 row-major matrix with the eval-mode BatchNorm's scale folded in
 * (M % 128 == 0, K % 32 == 0): [M/32][K/16][2 pieces][64 lanes][8] bf16.
 *   mvsdet_conv3d_k1_s2_bf16x3:  out (N,Cout,D/2,H/2,W/2) = [relu](W (Cout,Cin) x[:, :, ::2, ::2, ::2] + bias), D, H, W even
 *   mvsdet_convT3d_k2_s2_bf16x3: out (N,Cout,2D,2H,2W)[.., 2d+p, 2h+q, 2w+r] = [relu](sum_c x[c][d][h][w] W[c][o][p][q][r] + bias[o]);
 *                                the matrix has the 8*Cout rows m = 8 o + 4 p + 2 q + r (ConvTranspose3d weight permuted (1,2,3,4,0))
 * ------------------------------------------------------------------------------------------- */
size_t mvsdet_gemm_split_weight_bytes(int M, int K);   /* 0 if (M, K) is not supported */
int mvsdet_gemm_split_weight(const float* wmat, void* wsplit, int M, int K, mvsdet_stream_t stream);
int mvsdet_conv3d_k1_s2_bf16x3(const float* x, const void* wsplit, const float* bias, float* out, int N, int Cin, int Cout, int D,
                               int H, int W, int relu, mvsdet_stream_t stream);
int mvsdet_convT3d_k2_s2_bf16x3(const float* x, const void* wsplit, const float* bias, float* out, int N, int Cin, int Cout, int D,
                                int H, int W, int relu, mvsdet_stream_t stream);

#ifdef __cplusplus
}
#endif
#endif /* MVSDET_HIP_H */
