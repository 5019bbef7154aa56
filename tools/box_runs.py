#!/usr/bin/env python3
"""CPU-only footprint statistics for sweep-kernel design: per (view, tile, neighbour) the tap bounding boxes of all
planes, greedy runs of consecutive planes whose UNION box fits a cap, and what that means for LDS-DMA traffic.
Usage: python tools/box_runs.py [workload] [tw th] [cap ...]"""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np  # noqa: E402

import bench  # noqa: E402
from mvsdet_amd import synthetic  # noqa: E402
from oracle import oracle as O  # noqa: E402


def boxes_for(w, tw, th, seed=0):
    N, D, H, W = w["N"], w["D"], w["H"], w["W"]
    meta = synthetic.make_img_meta(N, (H, W), seed=seed, per_view_intrinsics=w["per_view_K"])
    ext = np.array(meta["lidar2img"]["extrinsic"])
    Kf = O.feat_intrinsics(np.array(meta["lidar2img"]["intrinsic"]), meta["img_shape"], meta["ori_shape"])
    c2w = np.linalg.inv(ext)
    nbr = O.knn_neighbors(c2w, 2)
    P, _ = O.relative_projections(ext, Kf, nbr)           # (N,K,4,4)
    depth = O.depth_planes(w["near_far"][0], w["near_far"][1], D).astype(np.float64)
    ys, xs = np.mgrid[0:H, 0:W].astype(np.float64)
    K = nbr.shape[1]
    tiles_x, tiles_y = (W + tw - 1) // tw, (H + th - 1) // th
    out = np.zeros((N, tiles_y * tiles_x, K, D, 4), np.int64)   # xlo,xhi,ylo,yhi (xhi<xlo = empty)
    for n in range(N):
        for j in range(K):
            p = P[n, j].astype(np.float64)
            rx = p[0, 0] * xs + p[0, 1] * ys + p[0, 2]
            ry = p[1, 0] * xs + p[1, 1] * ys + p[1, 2]
            rz = p[2, 0] * xs + p[2, 1] * ys + p[2, 2]
            for d in range(D):
                X, Y, Z = rx * depth[d] + p[0, 3], ry * depth[d] + p[1, 3], rz * depth[d] + p[2, 3]
                ix = (X / Z) / ((W - 1) * 0.5) * (W * 0.5) - 0.5
                iy = (Y / Z) / ((H - 1) * 0.5) * (H * 0.5) - 0.5
                x0, y0 = np.floor(ix), np.floor(iy)
                x0in, x1in = (x0 >= 0) & (x0 <= W - 1), (x0 >= -1) & (x0 <= W - 2)
                y0in, y1in = (y0 >= 0) & (y0 <= H - 1), (y0 >= -1) & (y0 <= H - 2)
                ok = (x0in | x1in) & (y0in | y1in)
                xlo = np.where(ok, np.where(x0in, x0, x0 + 1), 1e9)
                xhi = np.where(ok, np.where(x1in, x0 + 1, x0), -1e9)
                ylo = np.where(ok, np.where(y0in, y0, y0 + 1), 1e9)
                yhi = np.where(ok, np.where(y1in, y0 + 1, y0), -1e9)
                for ty in range(tiles_y):
                    for tx in range(tiles_x):
                        sl = (slice(ty * th, min(H, (ty + 1) * th)), slice(tx * tw, min(W, (tx + 1) * tw)))
                        out[n, ty * tiles_x + tx, j, d] = (xlo[sl].min(), xhi[sl].max(), ylo[sl].min(), yhi[sl].max())
    return out


def runs(b, cap):
    """b: (..., D, 4).  Greedy union runs along D.  Returns (#live, #dma (run starts), dma texels, #over cap)."""
    flat = b.reshape(-1, b.shape[-2], 4)
    live = dma = tex = over = 0
    for seq in flat:
        cur = None
        for (xlo, xhi, ylo, yhi) in seq:
            if xhi < xlo or yhi < ylo:
                continue  # skipped plane: the resident box stays valid
            live += 1
            a = (xhi - xlo + 1) * (yhi - ylo + 1)
            if a > cap:
                over += 1
                cur = None
                continue
            if cur is not None:
                u = (min(cur[0], xlo), max(cur[1], xhi), min(cur[2], ylo), max(cur[3], yhi))
                if (u[1] - u[0] + 1) * (u[3] - u[2] + 1) <= cap:
                    cur = u
                    continue
            cur = (xlo, xhi, ylo, yhi)
            dma += 1
            tex += a   # lower bound: the run's final union is larger
    return live, dma, tex, over


if __name__ == "__main__":
    name = sys.argv[1] if len(sys.argv) > 1 else "scannet_40v_64d_120x160"
    w = dict(bench.WORKLOADS[name])
    w["N"] = min(w["N"], int(os.environ.get("NVIEWS", "8")))
    tw, th = (int(sys.argv[2]), int(sys.argv[3])) if len(sys.argv) > 3 else (32, 4)
    caps = [int(c) for c in sys.argv[4:]] or [128, 160, 208, 256, 320, 448]
    b = boxes_for(w, tw, th)
    tot = b.shape[0] * b.shape[1] * b.shape[2] * b.shape[3]
    area = np.where((b[..., 1] >= b[..., 0]) & (b[..., 3] >= b[..., 2]), (b[..., 1] - b[..., 0] + 1) * (b[..., 3] - b[..., 2] + 1), 0)
    la = area[area > 0]
    print(f"{name} tile {tw}x{th} ({w['N']} views): triples {tot}, live {len(la) / tot:.3f}; single-plane area mean {la.mean():.0f} "
          f"p50 {np.percentile(la, 50):.0f} p90 {np.percentile(la, 90):.0f} p99 {np.percentile(la, 99):.0f}")
    for cap in caps:
        live, dma, tex, over = runs(b, cap)
        print(f"  cap {cap:4d}: over-cap {over / live:.3f}  DMA events / live triple {dma / live:.3f}  "
              f"(planes per run {live / max(dma, 1):.2f})  DMA texels per tile pixel-plane-nbr >= {tex / (live * tw * th):.3f}")
