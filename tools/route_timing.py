#!/usr/bin/env python3
"""Stage 1 through the two integration routes of INTEGRATION.md, timed on one scene:
  function-level: the reference's loop structure on the patched functions -- k calls of homo_warping (HIP warp
                  kernel, one materialised (N,C,D,H,W) volume per neighbour) + torch sum / square / variance;
  fused:          pack + plane_sweep_variance (what MVSDetHotPath.forward_scene runs).
Usage: python tools/route_timing.py [workload] [reps]"""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np  # noqa: E402
import torch  # noqa: E402

import bench  # noqa: E402
from mvsdet_amd import functional as F_, ops  # noqa: E402
from mvsdet_amd.hotpath import MVSDetHotPath  # noqa: E402

name = sys.argv[1] if len(sys.argv) > 1 else "scannet_ref_40v_12d_60x80"
reps = int(sys.argv[2]) if len(sys.argv) > 2 else 5
w = bench.WORKLOADS[name]
dev = torch.device("cuda:0")
hp = MVSDetHotPath(bench.N_VOXELS, bench.VOXEL_SIZE, list(w["near_far"]), w["D"])
s = bench.SceneInputs(w, 0, dev)
geo = hp.prepare_scene(s.meta, dev)
feat = s.features
N, C, H, W = feat.shape
K = geo.neighbor_ids.shape[1]
ident = torch.eye(4, device=dev).expand(N, 4, 4).contiguous()


def function_level():
    # volume_sum / volume_sq_sum loop of the reference (mvsdet.py:439-467), restated; homo_warping(src, src_proj,
    # ref_proj, depth) gets proj_rel as src_proj and the identity as ref_proj (same product)
    ref_volume = feat.unsqueeze(2).repeat(1, 1, w["D"], 1, 1)
    volume_sum = ref_volume
    volume_sq_sum = ref_volume ** 2
    for j in range(K):
        warped = F_.homo_warping(feat[geo.neighbor_ids[:, j]], geo.proj_rel[:, j].contiguous(), ident, geo.depth_values)
        volume_sum = volume_sum + warped
        volume_sq_sum = volume_sq_sum + warped ** 2
        del warped
    return volume_sq_sum.div_(K + 1).sub_(volume_sum.div_(K + 1).pow_(2))


def fused():
    return ops.plane_sweep_variance_packed(ops.pack_features(feat), geo.neighbor_ids, geo.proj_rel, geo.depth_values, C, H, W)


def function_level_lazy():
    # the same loop with the function-level patch active: homo_warping defers, the loop collapses into the fused kernel
    from mvsdet_amd import lazywarp
    F_.LAZY_WARP = True
    try:
        lazywarp.note_neighbor_ids(geo.neighbor_ids)
        ref_volume = feat.unsqueeze(2).repeat(1, 1, w["D"], 1, 1)
        volume_sum = ref_volume
        volume_sq_sum = ref_volume ** 2
        for j in range(K):
            warped = F_.homo_warping(feat[geo.neighbor_ids[:, j]], geo.proj_rel[:, j].contiguous(), ident, geo.depth_values)
            volume_sum += warped
            volume_sq_sum += warped.pow_(2)
            del warped
        return volume_sq_sum.div_(K + 1).sub_(volume_sum.div_(K + 1).pow_(2))
    finally:
        F_.LAZY_WARP = False


res = {}
for label, fn in (("function-level eager", function_level), ("function-level patched (deferred)", function_level_lazy), ("fused", fused)):
    out = fn()
    torch.cuda.synchronize()
    ts = []
    for _ in range(reps):
        del out
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        out = fn()
        e1.record()
        torch.cuda.synchronize()
        ts.append(e0.elapsed_time(e1))
    res[label] = (float(np.median(ts)), out)
b = res["fused"][1]
for label, (t, o) in res.items():
    print(f"{name}: {label:36s} {t:7.2f} ms   max |diff to fused| {float((o - b).abs().max()):.2e}")
