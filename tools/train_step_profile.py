#!/usr/bin/env python3
"""Operators (with input shapes) of one hot-path training step at the reference-true shape, by device time (torch profiler)."""
import argparse, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from torch.profiler import profile, ProfilerActivity
import bench
from mvsdet_amd.hotpath import MVSDetHotPath
w = bench.WORKLOADS["scannet_ref_40v_12d_60x80"]
dev = torch.device("cuda:0")
net = bench.PointwiseCostReg(w["C"]).to(dev)
opt = torch.optim.SGD(net.parameters(), lr=1e-4)
hp = MVSDetHotPath(bench.N_VOXELS, bench.VOXEL_SIZE, list(w["near_far"]), w["D"], topk=3, cost_regularization=net)
s = bench.SceneInputs(w, 0, dev)
geo = hp.prepare_scene(s.meta, dev)
def step():
    feat = s.features.detach().requires_grad_(True)
    out = hp.forward_scene(feat, s.meta, geo=geo)
    loss = out["volume"].square().mean() + out["depth_coding"].mean() + out["est_densities"].mean()
    opt.zero_grad(set_to_none=True)
    loss.backward()
    opt.step()
for _ in range(2):
    step()
torch.cuda.synchronize()
with profile(activities=[ProfilerActivity.CPU, ProfilerActivity.CUDA], record_shapes=True) as prof:
    step(); torch.cuda.synchronize()
print(prof.key_averages(group_by_input_shape=True).table(sort_by="self_device_time_total", row_limit=18, max_name_column_width=60, max_shapes_column_width=70))
