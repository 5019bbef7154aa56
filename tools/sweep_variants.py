#!/usr/bin/env python3
"""Times the sampling-table kernel and the slab kernel of the plane sweep under several option sets.
Usage: python tools/sweep_variants.py [workload] ["sweep_boxcap=256;sweep_tw=16;..."] [reps]"""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np  # noqa: E402
import torch  # noqa: E402

import bench  # noqa: E402
from mvsdet_amd import _lib, ops  # noqa: E402
from mvsdet_amd.hotpath import MVSDetHotPath  # noqa: E402

name = sys.argv[1] if len(sys.argv) > 1 else "scannet_40v_64d_120x160"
variants = sys.argv[2] if len(sys.argv) > 2 else ";"
reps = int(sys.argv[3]) if len(sys.argv) > 3 else 5
w = bench.WORKLOADS[name]
dev = torch.device("cuda:0")
hp = MVSDetHotPath(bench.N_VOXELS, bench.VOXEL_SIZE, list(w["near_far"]), w["D"])
s = bench.SceneInputs(w, 0, dev)
geo = hp.prepare_scene(s.meta, dev)
packed = ops.pack_features(s.features)
nbytes = bench.sweep_bytes_per_cv(w) * w["N"]
defaults = {k: _lib.get_option(k) for k in ("sweep_tw", "sweep_boxcap", "sweep_xcd")}
ref = None
for v in variants.split(";"):
    opts = dict(defaults)
    for kv in filter(None, v.split(",")):
        k, val = kv.split("=")
        opts[k] = int(val)
    for k, val in opts.items():
        _lib.set_option(k, val)
    tt, ts = [], []
    for i in range(reps):
        e = [torch.cuda.Event(enable_timing=True) for _ in range(3)]
        e[0].record()
        table = ops.plane_sweep_table(geo.proj_rel, geo.depth_values, w["H"], w["W"])
        e[1].record()
        var = ops.plane_sweep_variance_tabled(packed, geo.neighbor_ids, table, w["C"], w["D"], w["H"], w["W"])
        e[2].record()
        torch.cuda.synchronize()
        tt.append(e[0].elapsed_time(e[1]))
        ts.append(e[1].elapsed_time(e[2]))
        chk = float(var[::7, ::13].double().sum())
        if ref is None:
            ref = chk
        same = chk == ref
        del var, table
    print(f"{name} [{v}] table {min(tt):.3f} ms  slab {min(ts):.3f} ms (median {np.median(ts):.3f}) -> "
          f"{nbytes / (min(ts) * 1e-3) / 1e9:.0f} GB/s = {nbytes / (min(ts) * 1e-3) / 8e12:.3f} of 8 TB/s; same bits as first: {same}",
          flush=True)
