#!/usr/bin/env python3
"""Interleaved A/B timing of the plane sweep of TWO builds of libmvsdet_hip.so in one process (cdna guide rule 24):
   python tools/ab_lib.py A.so B.so [C.so ...] [workload] [rounds]
Only the entry points both builds are certain to share are bound (pack + packed sweep); results must agree bit for bit."""
import ctypes
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np  # noqa: E402
import torch  # noqa: E402

import bench  # noqa: E402
from mvsdet_amd.hotpath import MVSDetHotPath  # noqa: E402

paths = [a for a in sys.argv[1:] if a.endswith(".so")]
rest = [a for a in sys.argv[1:] if not a.endswith(".so")]
name = rest[0] if rest else "scannet_40v_64d_120x160"
rounds = int(rest[1]) if len(rest) > 1 else 6
vp, i, sz = ctypes.c_void_p, ctypes.c_int, ctypes.c_size_t
libs = []
for p in paths:
    lib = ctypes.CDLL(os.path.abspath(p))
    lib.mvsdet_packed_bytes.restype = sz
    lib.mvsdet_packed_bytes.argtypes = [i, i, i, i]
    lib.mvsdet_plane_sweep_scratch_bytes.restype = sz
    lib.mvsdet_plane_sweep_scratch_bytes.argtypes = [i, i, i, i, i]
    lib.mvsdet_pack_features_f32.argtypes = [vp, ctypes.POINTER(ctypes.c_int64), vp, i, i, i, i, vp]
    lib.mvsdet_plane_sweep_variance_packed_f32.argtypes = [vp, vp, vp, vp, vp, vp, sz, i, i, i, i, i, i, vp]
    libs.append(lib)

w = bench.WORKLOADS[name]
dev = torch.device("cuda:0")
hp = MVSDetHotPath(bench.N_VOXELS, bench.VOXEL_SIZE, list(w["near_far"]), w["D"])
s = bench.SceneInputs(w, 0, dev)
geo = hp.prepare_scene(s.meta, dev)
N, C, H, W, D = w["N"], w["C"], w["H"], w["W"], w["D"]
K = geo.neighbor_ids.shape[1]
feat = s.features.contiguous()
stream = ctypes.c_void_p(torch.cuda.current_stream().cuda_stream)
packed = torch.empty(libs[0].mvsdet_packed_bytes(N, C, H, W) // 4, device=dev)
strides = (ctypes.c_int64 * 4)(*feat.stride())
assert libs[0].mvsdet_pack_features_f32(feat.data_ptr(), strides, packed.data_ptr(), N, C, H, W, stream) == 0
sb = libs[0].mvsdet_plane_sweep_scratch_bytes(N, K, D, H, W)
scratch = torch.empty(sb // 4, device=dev)
var = torch.empty((N, C, D, H, W), device=dev)
nbr, proj, depth = geo.neighbor_ids.contiguous(), geo.proj_rel.contiguous(), geo.depth_values.contiguous()


def run(lib):
    rc = lib.mvsdet_plane_sweep_variance_packed_f32(packed.data_ptr(), nbr.data_ptr(), proj.data_ptr(), depth.data_ptr(),
                                                    var.data_ptr(), scratch.data_ptr(), sb, N, K, C, D, H, W, stream)
    assert rc == 0, rc


sums = []
times = [[] for _ in libs]
for r in range(rounds + 1):
    for k, lib in enumerate(libs):
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        run(lib)
        e1.record()
        torch.cuda.synchronize()
        if r == 0:
            sums.append(var.view(-1)[:: 4099].double().sum().item())
        else:
            times[k].append(e0.elapsed_time(e1))
assert len(set(sums)) == 1, f"builds disagree: {sums}"
b = bench.sweep_bytes_per_cv(w) * N
for k, p in enumerate(paths):
    t = np.array(times[k])
    print(f"{name} {p}: median {np.median(t):.3f} ms min {t.min():.3f} ms -> {b / (np.median(t) * 1e-3) / 1e9:.0f} GB/s")
