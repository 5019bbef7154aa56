#!/usr/bin/env python3
"""Backward of the plane sweep with the LDS gradient images in doubles (bwd_groups-1=0) and in 64-bit fixed point (bwd_groups-1=1):
time of the whole call (pack, memset, geometry, bound, kernel, unpack) at a bench workload, the difference between the two
results, and -- fixed point only -- whether two runs agree bit for bit."""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa: E402

import bench  # noqa: E402
from mvsdet_amd import _lib, ops  # noqa: E402
from mvsdet_amd.hotpath import MVSDetHotPath  # noqa: E402

names = sys.argv[1:] or ["scannet_ref_40v_12d_60x80"]
dev = torch.device("cuda:0")
for name in names:
    w = bench.WORKLOADS[name]
    hp = MVSDetHotPath(bench.N_VOXELS, bench.VOXEL_SIZE, list(w["near_far"]), w["D"])
    s = bench.SceneInputs(w, 0, dev)
    geo = hp.prepare_scene(s.meta, dev)
    g = torch.randn((w["N"], w["C"], w["D"], w["H"], w["W"]), device=dev)
    g *= torch.exp(3.0 * torch.randn_like(g[:, :, :1]))   # heavy-tailed magnitudes: the bound lies far above the typical gradient
    out = {}
    for fixed in (0, 1, 0, 1):   # groups - 1
        _lib.set_option("bwd_groups", fixed + 1)
        ts = []
        for _ in range(6):
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            r = ops.plane_sweep_variance_backward(s.features, geo.neighbor_ids, geo.proj_rel, geo.depth_values, g)
            e1.record()
            torch.cuda.synchronize()
            ts.append(e0.elapsed_time(e1))
        print(f"{name} bwd_groups-1={fixed}: min {min(ts):.3f} ms  median {sorted(ts)[len(ts) // 2]:.3f} ms", flush=True)
        out[fixed] = r
    d = (out[0].double() - out[1].double()).abs().max().item()
    print(f"   max |one group - two groups| = {d:.3e}   max |grad| = {out[0].abs().max().item():.3e}   ratio {d / out[0].abs().max().item():.2e}", flush=True)
_lib.set_option("bwd_groups", 0)
