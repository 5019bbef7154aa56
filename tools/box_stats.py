#!/usr/bin/env python3
"""Footprint-box statistics of a workload's sampling table: how many (tile, plane, neighbour) boxes are empty, fit
the LDS box (staged) or fall back to global gathers, and how large they are.  Usage: python tools/box_stats.py [workload ...]"""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np  # noqa: E402
import torch  # noqa: E402

import bench  # noqa: E402
from mvsdet_amd import ops  # noqa: E402
from mvsdet_amd.hotpath import MVSDetHotPath  # noqa: E402

CAP = 256
for name in sys.argv[1:] or ["scannet_40v_64d_120x160"]:
    w = bench.WORKLOADS[name]
    dev = torch.device("cuda:0")
    hp = MVSDetHotPath(bench.N_VOXELS, bench.VOXEL_SIZE, list(w["near_far"]), w["D"])
    from mvsdet_amd import synthetic
    meta = synthetic.make_img_meta(w["N"], (w["H"], w["W"]), seed=0, per_view_intrinsics=w["per_view_K"])
    geo = hp.prepare_scene(meta, dev)
    N, K, D, H, W = w["N"], geo.neighbor_ids.shape[1], w["D"], w["H"], w["W"]
    from mvsdet_amd import _lib
    tw, th, CAP = _lib.sweep_tile_shape(K, D, H, W)
    tiles = ((W + tw - 1) // tw) * ((H + th - 1) // th)
    table = ops.plane_sweep_table(geo.proj_rel, geo.depth_values, H, W)
    nent = N * tiles * D * K
    boxes = table[:nent * 4].view(torch.int32).view(nent, 4).cpu().numpy().astype(np.int64)
    nc, nr = boxes[:, 1] - boxes[:, 0] + 1, boxes[:, 3] - boxes[:, 2] + 1
    empty = (nc <= 0) | (nr <= 0)
    area = np.where(empty, 0, nc * nr)
    staged = ~empty & (area <= CAP)
    fb = ~empty & (area > CAP)
    pieces = np.where(staged, nr * ((nc * 8 + 63) // 64), 0)
    print(f"{name} TW={tw}: {nent} boxes; empty {empty.mean():.3f} staged {staged.mean():.3f} fallback {fb.mean():.3f}; "
          f"staged area mean {area[staged].mean():.0f} p90 {np.percentile(area[staged], 90):.0f}; "
          f"DMA pieces/box mean {pieces[staged].mean():.1f} (useful {area[staged].mean() / 8:.1f}); "
          f"fallback area median {np.median(area[fb]) if fb.any() else 0:.0f}")
    if K == 2:  # per (tile, plane): how the two neighbours combine
        st = np.where(empty, 0, np.where(staged, 1, 2)).reshape(-1, 2)      # 0 skipped, 1 staged, 2 gathered
        ar = area.reshape(-1, 2)
        both_skip = (st == 0).all(1).mean()
        one_live = ((st == 0).sum(1) == 1).mean()
        both_staged = (st == 1).all(1)
        print(f"    planes: both skipped {both_skip:.3f}, one live {one_live:.3f}, both live {1 - both_skip - one_live:.3f}; "
              f"both staged with area0+area1 <= {CAP}: {(both_staged & (ar.sum(1) <= CAP)).mean():.3f} (of both staged {both_staged.mean():.3f})")
