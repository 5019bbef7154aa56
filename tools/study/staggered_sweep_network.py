#!/usr/bin/env python3
"""Sweep + cost network of a scene at the reference-true shape: (1) one sweep of all views, then the network (two halves of the views
on two streams, as shipped); (2) staggered -- sweep of the first half, then its network on this stream while the second half's sweep
(store-bound) and network run on a second stream beside the first half's convolutions (matrix-core-bound).  GPU box."""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch  # noqa: E402

import bench  # noqa: E402
from mvsdet_amd import ops  # noqa: E402
from mvsdet_amd.costreg import CostRegNet3DGS  # noqa: E402
from mvsdet_amd.hotpath import MVSDetHotPath  # noqa: E402

dev = torch.device("cuda:0")
w = bench.WORKLOADS["scannet_ref_40v_12d_60x80"]
torch.manual_seed(0)
net = CostRegNet3DGS(w["C"]).to(dev).eval()
hp = MVSDetHotPath(bench.N_VOXELS, bench.VOXEL_SIZE, list(w["near_far"]), w["D"], topk=3)
s = bench.SceneInputs(w, 0, dev)
geo = hp.prepare_scene(s.meta, dev)
N, C, H, W = w["N"], w["C"], w["H"], w["W"]
side = torch.cuda.Stream(device=dev)


def shipped():
    packed = ops.pack_features(s.features)
    var = ops.plane_sweep_variance_shard(packed, geo.neighbor_ids, geo.proj_rel, geo.depth_values, N, 0, C, H, W)
    return net(var)


def staggered(cut=20):
    cur = torch.cuda.current_stream(dev)
    packed = ops.pack_features(s.features)
    va = ops.plane_sweep_variance_shard(packed, geo.neighbor_ids[:cut], geo.proj_rel[:cut], geo.depth_values[:cut], N, 0, C, H, W)
    side.wait_stream(cur)            # packed is ready and the first half's sweep is through
    with torch.cuda.stream(side):
        vb = ops.plane_sweep_variance_shard(packed, geo.neighbor_ids[cut:], geo.proj_rel[cut:], geo.depth_values[cut:], N, cut, C, H, W)
        lb = net._forward_chain(vb)
    la = net._forward_chain(va)
    cur.wait_stream(side)
    lb.record_stream(cur)
    packed.record_stream(side)
    return torch.cat((la, lb), 0)


def timed(fn, reps=10):
    for _ in range(3):
        fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps):
        out = fn()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / reps, out


with torch.no_grad():
    for _ in range(2):
        t1, o1 = timed(shipped)
        t2, o2 = timed(staggered)
        t3, o3 = timed(lambda: staggered(16))
        print(f"sweep, then network on two halves: {t1:.3f} ms   staggered 20 + 20: {t2:.3f} ms (equal bits: {bool(torch.equal(o1, o2))})   16 + 24: {t3:.3f} ms", flush=True)
