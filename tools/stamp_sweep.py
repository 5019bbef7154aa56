#!/usr/bin/env python3
"""Diagnostic: where a wave of the plane-sweep slab kernel spends its cycles (s_memtime stamps, separate build path)."""
import ctypes, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import bench
from mvsdet_amd import _lib, ops
from mvsdet_amd.hotpath import MVSDetHotPath
name = sys.argv[1] if len(sys.argv) > 1 else "scannet_40v_64d_120x160"
w = bench.WORKLOADS[name]
dev = torch.device("cuda:0")
hp = MVSDetHotPath(bench.N_VOXELS, bench.VOXEL_SIZE, list(w["near_far"]), w["D"])
s = bench.SceneInputs(w, 0, dev)
geo = hp.prepare_scene(s.meta, dev)
packed = ops.pack_features(s.features)
lib = ctypes.CDLL(_lib.LIB_PATH)
buf = torch.zeros(65536 * 4 * 8, dtype=torch.int64, device=dev)
for stamped in (False, True, True):
    lib.mvsdet_debug_set_stamp_buffer(ctypes.c_void_p(buf.data_ptr() if stamped else 0))
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    var = ops.plane_sweep_variance_packed(packed, geo.neighbor_ids, geo.proj_rel, geo.depth_values, w["C"], w["H"], w["W"])
    e1.record(); torch.cuda.synchronize()
    print("stamped" if stamped else "plain", e0.elapsed_time(e1), "ms")
    del var
lib.mvsdet_debug_set_stamp_buffer(None)
t = buf.view(65536, 4, 8).double()
t = t[t.sum(dim=(1, 2)) > 0]
per = t.mean(dim=(0, 1)) / w["D"]
names = ["boxes+prefetch wait+barrier", "dma0 issue+decode", "box0 wait+barrier", "taps0+barrier", "dma1 issue+wait+barrier", "taps1+barrier", "variance->tile+barrier", "tile->global stores"]
tot = per.sum().item()
for n_, v in zip(names, per.tolist()):
    print(f"{n_:30s} {v:9.0f} cycles/plane  {100 * v / tot:5.1f} %")
print("total", tot, "cycles per plane per wave; blocks sampled", t.shape[0])
