#!/usr/bin/env python3
"""Backward of the plane sweep with the software-pipelined plane loop (option bwd_pipe = 1) against round 4's kernel (0): time
of the whole operator (pack, memset, geometry, kernel, unpack) at bench workloads, alternating, and the difference between the
two results (sums of the same terms; only the order of atomics differs).  GPU box: python tools/bwd_pipe_ab.py [workload ...]"""
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
    out = {}
    for pipe in (0, 1, 0, 1):
        _lib.set_option("bwd_pipe", pipe)
        ts = []
        for _ in range(8):
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            r = ops.plane_sweep_variance_backward(s.features, geo.neighbor_ids, geo.proj_rel, geo.depth_values, g)
            e1.record()
            torch.cuda.synchronize()
            ts.append(e0.elapsed_time(e1))
        print(f"{name} bwd_pipe={pipe}: min {min(ts):.3f} ms  median {sorted(ts)[len(ts) // 2]:.3f} ms", flush=True)
        out[pipe] = r
    d = (out[0].double() - out[1].double()).abs().max().item()
    print(f"   max |round-4 kernel - pipelined| = {d:.3e}   max |grad| = {out[0].abs().max().item():.3e}   ratio {d / out[0].abs().max().item():.2e}", flush=True)
_lib.set_option("bwd_pipe", 1)
