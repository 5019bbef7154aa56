#!/usr/bin/env python3
"""Where the host time of one scene goes at the reference-true shape (GPU work there is only ~1.4 ms)."""
import cProfile, os, pstats, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import bench
from mvsdet_amd import ops
from mvsdet_amd.hotpath import MVSDetHotPath
w = bench.WORKLOADS["scannet_ref_40v_12d_60x80"]
dev = torch.device("cuda:0")
hp = MVSDetHotPath(bench.N_VOXELS, bench.VOXEL_SIZE, list(w["near_far"]), w["D"])
s = bench.SceneInputs(w, 0, dev)
def step():
    geo = hp.prepare_scene(s.meta, dev)
    packed = ops.pack_features(s.features)
    var = ops.plane_sweep_variance_packed(packed, geo.neighbor_ids, geo.proj_rel, geo.depth_values, w["C"], w["H"], w["W"])
    prob, off, ed, en, ei, avg = hp.depth_distribution(s.cost_logits)
    return hp.lift(s.features, packed, geo, ed, en)
for _ in range(5): step()
torch.cuda.synchronize()
t = time.perf_counter()
for _ in range(50): step()
torch.cuda.synchronize()
print("ms/scene", (time.perf_counter() - t) / 50 * 1e3)
pr = cProfile.Profile(); pr.enable()
for _ in range(50): step()
torch.cuda.synchronize()
pr.disable(); pstats.Stats(pr).sort_stats("cumtime").print_stats(22)
