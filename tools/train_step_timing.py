#!/usr/bin/env python3
"""Forward + backward timing of the hot-path operators at the reference-true shape (training configuration)."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import bench
from mvsdet_amd import ops
from mvsdet_amd.hotpath import MVSDetHotPath
w = bench.WORKLOADS["scannet_ref_40v_12d_60x80"]
dev = torch.device("cuda:0")
hp = MVSDetHotPath(bench.N_VOXELS, bench.VOXEL_SIZE, list(w["near_far"]), w["D"])
s = bench.SceneInputs(w, 0, dev)
geo = hp.prepare_scene(s.meta, dev)
feat = s.features.clone().requires_grad_(True)
logits = s.cost_logits.clone().requires_grad_(True)
def ev():
    return torch.cuda.Event(enable_timing=True)
def timed(fn, reps=5):
    fn(); torch.cuda.synchronize()
    ts = []
    for _ in range(reps):
        a, b = ev(), ev(); a.record(); fn(); b.record(); torch.cuda.synchronize(); ts.append(a.elapsed_time(b))
    return min(ts)
var = ops.plane_sweep_variance(feat, geo.neighbor_ids, geo.proj_rel, geo.depth_values)
g = torch.randn_like(var)
print("stage1 fwd ms", timed(lambda: ops.plane_sweep_variance(feat, geo.neighbor_ids, geo.proj_rel, geo.depth_values)))
print("stage1 bwd ms", timed(lambda: ops.plane_sweep_variance_backward(feat.detach(), geo.neighbor_ids, geo.proj_rel, geo.depth_values, g)))
del var, g
def full():
    feat.grad = None; logits.grad = None
    out = hp.forward_scene(feat, s.meta, cost_logits=logits, geo=geo)
    (out["volume"].sum() + out["variance"].mean()).backward()
print("scene fwd+bwd ms (a1..a10, stand-in logits)", timed(full, 3))
