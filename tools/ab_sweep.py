#!/usr/bin/env python3
"""Interleaved A/B timing of plane-sweep kernel variants in ONE process (cdna guide rule 24).
Usage: python tools/ab_sweep.py [workload] [rounds] -- variants are env settings read per call by the library."""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np  # noqa: E402
import torch  # noqa: E402

import bench  # noqa: E402
from mvsdet_amd import ops  # noqa: E402
from mvsdet_amd.hotpath import MVSDetHotPath  # noqa: E402

name = sys.argv[1] if len(sys.argv) > 1 else "scannet_40v_64d_120x160"
rounds = int(sys.argv[2]) if len(sys.argv) > 2 else 5
VARIANTS = [dict(x.split("=") for x in v.split(",")) for v in os.environ.get(
    "AB_VARIANTS", "MVSDET_SWEEP_TW=32,MVSDET_SWEEP_NT=1;MVSDET_SWEEP_TW=32,MVSDET_SWEEP_NT=0;"
                   "MVSDET_SWEEP_TW=16,MVSDET_SWEEP_NT=1;MVSDET_SWEEP_TW=16,MVSDET_SWEEP_NT=0").split(";")]
w = bench.WORKLOADS[name]
dev = torch.device("cuda:0")
hp = MVSDetHotPath(bench.N_VOXELS, bench.VOXEL_SIZE, list(w["near_far"]), w["D"])
s = bench.SceneInputs(w, 0, dev)
geo = hp.prepare_scene(s.meta, dev)
packed = ops.pack_features(s.features)
var = ops.plane_sweep_variance_packed(packed, geo.neighbor_ids, geo.proj_rel, geo.depth_values, w["C"], w["H"], w["W"])
ref = var.clone() if var.numel() < 2 ** 31 else None
del var
torch.cuda.synchronize()
times = {i: [] for i in range(len(VARIANTS))}
for r in range(rounds + 1):
    for i, v in enumerate(VARIANTS):
        os.environ.update(v)
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        var = ops.plane_sweep_variance_packed(packed, geo.neighbor_ids, geo.proj_rel, geo.depth_values, w["C"], w["H"], w["W"])
        e1.record()
        torch.cuda.synchronize()
        if r > 0:
            times[i].append(e0.elapsed_time(e1))
        if ref is not None and r == 0:
            assert torch.equal(var, ref), f"variant {v} changes the result"
        del var
b = bench.sweep_bytes_per_cv(w) * w["N"]
for i, v in enumerate(VARIANTS):
    t = np.array(times[i])
    print(f"{name} {v}: median {np.median(t):.3f} ms min {t.min():.3f} ms -> {b / (np.median(t) * 1e-3) / 1e9:.0f} GB/s")
