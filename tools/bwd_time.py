#!/usr/bin/env python3
"""Time of the backward sweep's whole operator at a bench workload: the target of tools/ab_libs.sh runs over what-if builds of
the kernel (a part of its work removed: the RESULT is then wrong, the time tells what that part costs).
python tools/bwd_time.py [workload] [option=value ...]"""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa: E402

import bench  # noqa: E402
from mvsdet_amd import _lib, ops  # noqa: E402
from mvsdet_amd.hotpath import MVSDetHotPath  # noqa: E402

name = sys.argv[1] if len(sys.argv) > 1 else "scannet_ref_40v_12d_60x80"
for kv in sys.argv[2:]:
    _lib.set_option(kv.split("=")[0], int(kv.split("=")[1]))
w = bench.WORKLOADS[name]
dev = torch.device("cuda:0")
hp = MVSDetHotPath(bench.N_VOXELS, bench.VOXEL_SIZE, list(w["near_far"]), w["D"])
s = bench.SceneInputs(w, 0, dev)
geo = hp.prepare_scene(s.meta, dev)
g = torch.randn((w["N"], w["C"], w["D"], w["H"], w["W"]), device=dev)
ts = []
for _ in range(10):
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    ops.plane_sweep_variance_backward(s.features, geo.neighbor_ids, geo.proj_rel, geo.depth_values, g)
    e1.record()
    torch.cuda.synchronize()
    ts.append(e0.elapsed_time(e1))
print(f"{name} backward operator: min {min(ts):.3f} ms median {sorted(ts)[5]:.3f} ms", flush=True)
