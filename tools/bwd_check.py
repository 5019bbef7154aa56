import sys, os
sys.path.insert(0, ".")
import torch
import bench
from mvsdet_amd import _lib, ops
from mvsdet_amd.hotpath import MVSDetHotPath
w = bench.WORKLOADS["scannet_ref_40v_12d_60x80"]
dev = torch.device("cuda:0")
hp = MVSDetHotPath(bench.N_VOXELS, bench.VOXEL_SIZE, list(w["near_far"]), w["D"])
s = bench.SceneInputs(w, 0, dev)
geo = hp.prepare_scene(s.meta, dev)
torch.manual_seed(1)
g = torch.randn((w["N"], w["C"], w["D"], w["H"], w["W"]), device=dev)
r = ops.plane_sweep_variance_backward(s.features, geo.neighbor_ids, geo.proj_rel, geo.depth_values, g)
print(os.environ.get("MVSDET_HIP_LIB", "default").split("/")[-1], "checksum", float(r.double().abs().sum()), float(r.double().pow(2).sum()))
