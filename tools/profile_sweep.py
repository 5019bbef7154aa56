#!/usr/bin/env python3
"""Runs pack + plane-sweep variance a few times on one synthetic scene: the target of rocprofv3 runs
(kernel trace, then one --pmc set per run).  Usage: python tools/profile_sweep.py [workload] [reps]"""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa: E402

import bench  # noqa: E402
from mvsdet_amd import ops  # noqa: E402
from mvsdet_amd.hotpath import MVSDetHotPath  # noqa: E402

name = sys.argv[1] if len(sys.argv) > 1 else "scannet_40v_64d_120x160"
reps = int(sys.argv[2]) if len(sys.argv) > 2 else 3
w = bench.WORKLOADS[name]
dev = torch.device("cuda:0")
hp = MVSDetHotPath(bench.N_VOXELS, bench.VOXEL_SIZE, list(w["near_far"]), w["D"])
s = bench.SceneInputs(w, 0, dev)
geo = hp.prepare_scene(s.meta, dev)
ts = []
if w.get("half"):   # BASELINE configs[4] as worded: fp16 feature maps in, fp16 cost volume out, ONE chunk of reference views per rep
    feats = s.features.half() if s.features.dtype != torch.float16 else s.features
    for i in range(reps):
        packed = ops.pack_features(feats)
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for first, var in hp.cost_volume_chunks(packed, geo, w["C"], w["H"], w["W"], w["chunk"], half_out=True):
            del var
            break
        e1.record()
        torch.cuda.synchronize()
        ts.append(e0.elapsed_time(e1))
    b = bench.sweep_bytes_per_cv(w) * w["chunk"]
    print(f"{name} (one {w['chunk']}-view chunk, fp16 storage) sweep ms {ts} -> {b / (min(ts) * 1e-3) / 1e9:.1f} GB/s algorithmic")
    sys.exit(0)
for i in range(reps):
    packed = ops.pack_features(s.features)
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    var = ops.plane_sweep_variance_packed(packed, geo.neighbor_ids, geo.proj_rel, geo.depth_values, w["C"], w["H"], w["W"])
    e1.record()
    torch.cuda.synchronize()
    ts.append(e0.elapsed_time(e1))
    del var
b = bench.sweep_bytes_per_cv(w) * w["N"]
print(f"{name} tile={os.environ.get('MVSDET_SWEEP_TILE', 'default')} sweep ms {ts} -> {b / (min(ts) * 1e-3) / 1e9:.1f} GB/s algorithmic")
