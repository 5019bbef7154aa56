"""Pricing of a GATHER-form adjoint of the plane sweep: a block owns a SOURCE tile and, for every texel, enumerates the
reference pixels whose bilinear taps hit it (the inverse of a plane homography is a homography), accumulating in registers -- no LDS
images, no atomics.  Counted here on the synthetic geometry of the bench workloads (float64 positions, the forward's sampling rule
ix = X/Z * W/(W-1) - 0.5):
  * contributions: taps that really land on a texel, per (reference view, neighbour, plane) map;
  * candidates: reference pixels a gather kernel has to TEST per texel and map -- the integer pixels of the bounding box of the
    pre-image of the texel's 2 x 2 tap square, one pixel of margin for the rounding of the inverse;
  * bytes: what each form moves.
CPU, seconds: python tools/study/r06_bwd_gather_pricing.py [workload ...]"""
import os
import sys

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch  # noqa: E402

import bench  # noqa: E402
from mvsdet_amd.hotpath import MVSDetHotPath  # noqa: E402

for name in sys.argv[1:] or ("scannet_ref_40v_12d_60x80", "arkit_test_100v_12d_60x80"):
    w = bench.WORKLOADS[name]
    N, C, D, H, W = w["N"], w["C"], w["D"], w["H"], w["W"]
    hp = MVSDetHotPath(bench.N_VOXELS, bench.VOXEL_SIZE, list(w["near_far"]), D)
    meta = bench.unseen_metas(w, 0, 1)[0]
    nbr, proj_rel, depth, *_ = hp._host_geometry(meta)
    P = proj_rel.double().numpy()
    K = nbr.shape[1]
    ys, xs = np.meshgrid(np.arange(H, dtype=np.float64), np.arange(W, dtype=np.float64), indexing="ij")
    xyz = np.stack([xs.ravel(), ys.ravel(), np.ones(H * W)])
    contrib = cand = maps = live_px = 0
    cand_hist = []
    for n in range(N):
        for j in range(K):
            R, t = P[n, j, :3, :3], P[n, j, :3, 3:4]
            rot = R @ xyz
            for d in range(D):
                q = rot * float(depth[n, d]) + t
                with np.errstate(divide="ignore", invalid="ignore"):
                    ix = q[0] / q[2] * W / (W - 1) - 0.5
                    iy = q[1] / q[2] * H / (H - 1) - 0.5
                ok = np.isfinite(ix) & np.isfinite(iy) & (ix > -1) & (ix < W) & (iy > -1) & (iy < H)
                maps += 1
                live_px += int(ok.sum())
                if not ok.any():
                    continue
                # taps that land inside the source image
                x0, y0 = np.floor(ix[ok]), np.floor(iy[ok])
                contrib += int(((x0 >= 0) & (y0 >= 0)).sum() + ((x0 + 1 < W) & (y0 >= 0)).sum() + ((x0 >= 0) & (y0 + 1 < H)).sum() + ((x0 + 1 < W) & (y0 + 1 < H)).sum())
                # local Jacobian of the forward map (finite differences on the pixel grid) -> pre-image of a 2 x 2 square of texels
                IX, IY = ix.reshape(H, W), iy.reshape(H, W)
                ax, ay = np.gradient(IX, axis=1), np.gradient(IX, axis=0)
                bx, by = np.gradient(IY, axis=1), np.gradient(IY, axis=0)
                det = ax * by - ay * bx
                okm = ok.reshape(H, W) & np.isfinite(det) & (np.abs(det) > 1e-6)
                # inverse Jacobian rows; the pre-image of [-1,1]^2 is a parallelogram: its bounding box in pixels (+1 margin each side)
                wx = (np.abs(by) + np.abs(ay)) / np.abs(det)
                wy = (np.abs(bx) + np.abs(ax)) / np.abs(det)
                c = (np.ceil(2 * wx[okm]) + 2) * (np.ceil(2 * wy[okm]) + 2)
                cand += float(c.sum()) / max(float(np.abs(det[okm]).mean()), 1e-9) * 0 + float(c.mean()) * float(okm.sum() * np.abs(det[okm]).mean())
                cand_hist.append(float(c.mean()))
    texel_maps = live_px   # texels under a live map ~ live pixels x |det| ~ live pixels (scale ~ 1)
    gvar_bytes = N * C * D * H * W * 4
    print(f"== {name}: {N} views, K = {K}, {D} planes, {H} x {W}; {maps} maps, live pixels per map {live_px / maps / (H * W):.2f} of the image")
    print(f"   contributions (taps that land): {contrib / 1e6:.1f} M per scene = {contrib / max(live_px, 1):.2f} per live (pixel, map) = per texel under a live map")
    print(f"   candidates a gather kernel TESTS per texel and map: mean {np.mean(cand_hist):.1f} (bounding box of the pre-image of the 2 x 2 tap square + 1 px margin)")
    print(f"   -> position evaluations (four IEEE divisions each): scatter form {live_px / 1e6:.1f} M (one per live pixel and map), "
          f"gather form {np.mean(cand_hist) * live_px / 1e6:.0f} M = {np.mean(cand_hist):.0f} x")
    print(f"   bytes: scatter form reads dL/dvar once: {gvar_bytes / 1e9:.2f} GB (+ features and gradient {2 * N * C * H * W * 4 / 1e9:.2f}) = {(gvar_bytes + 2 * N * C * H * W * 4) / 1e9:.2f} GB")
    print(f"          gather form: dL/dw_j = 2r dL/dvar w_j - 2r^2 dL/dvar S needs S (the sum over the views at the REFERENCE pixel): a pre-pass that writes "
          f"dL/dvar*S ({gvar_bytes / 1e9:.2f} GB) and reads dL/dvar ({gvar_bytes / 1e9:.2f}); the gather pass reads both volumes once per neighbour "
          f"({2 * K * gvar_bytes / 1e9:.2f} GB, tiles staged in LDS) -> {(2 + 2 * K) * gvar_bytes / 1e9:.1f} GB = {(2 + 2 * K) * gvar_bytes / (gvar_bytes + 2 * N * C * H * W * 4):.1f} x")
