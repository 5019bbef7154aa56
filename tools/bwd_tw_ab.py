#!/usr/bin/env python3
"""Backward sweep at a bench workload under option settings given as name=value[,value] (swept as a product), e.g.
python tools/bwd_tw_ab.py scannet_ref_40v_12d_60x80 sweep_tw=0,32 bwd_groups=1,2"""
import itertools
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa: E402

import bench  # noqa: E402
from mvsdet_amd import _lib, ops  # noqa: E402
from mvsdet_amd.hotpath import MVSDetHotPath  # noqa: E402

name = sys.argv[1]
sweeps = [(kv.split("=")[0], [int(v) for v in kv.split("=")[1].split(",")]) for kv in sys.argv[2:]]
w = bench.WORKLOADS[name]
dev = torch.device("cuda:0")
hp = MVSDetHotPath(bench.N_VOXELS, bench.VOXEL_SIZE, list(w["near_far"]), w["D"])
s = bench.SceneInputs(w, 0, dev)
geo = hp.prepare_scene(s.meta, dev)
g = torch.randn((w["N"], w["C"], w["D"], w["H"], w["W"]), device=dev)
ref = None
for rnd in range(2):
    for combo in itertools.product(*[v for _, v in sweeps]):
        for (k, _), v in zip(sweeps, combo):
            _lib.set_option(k, v)
        ts = []
        for _ in range(8):
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            r = ops.plane_sweep_variance_backward(s.features, geo.neighbor_ids, geo.proj_rel, geo.depth_values, g)
            e1.record()
            torch.cuda.synchronize()
            ts.append(e0.elapsed_time(e1))
        if ref is None:
            ref = r
        d = float((r.double() - ref.double()).abs().max() / ref.abs().max())
        print(f"{name} {dict(zip([k for k, _ in sweeps], combo))}: min {min(ts):.3f} ms median {sorted(ts)[4]:.3f} ms   rel diff to first {d:.1e}", flush=True)
