"""Host-side mirror of the reference's Python interface for the hot path.

Same names, argument order, shapes and return arity as projects/NeRF-Det/nerfdet/ of Pixie8888/MVSDet
(mvs_models/module.py and mvsdet.py; file:line cited per function), so the reference's call sites -- and
tests written against them -- read the same.  Heavy tensors go to the HIP operators in ops.py.

Geometry stays on the host.  The camera matrices of a scene are a few hundred floats that arrive as numpy
arrays in img_meta (mvsdet.py:419-420).  The sampling positions are sensitive to the rounding of the 4x4
inverse at the 1e-4 px level (DESIGN.md "tolerance budget"), so `relative_projection` evaluates
`src_proj @ inverse(ref_proj)` with the very ATen-CPU ops the reference uses (module.py:116) and ships the
result to the device; nothing per-pixel ever runs on the CPU.
"""
from __future__ import annotations

from typing import Sequence, Tuple

import numpy as np
import torch
from torch import Tensor

from . import ops


# ------------------------------------------------------------------------------------------- geometry (host)
# Set by integration.patch_reference: homo_warping then returns a deferred volume (lazywarp.LazyVolume) instead of
# launching the warp kernel.  Off for direct callers of this module.
LAZY_WARP = False
LAZY_ANY_DEVICE = False   # tests only: defer on CPU tensors too (the fused launches are stubbed there)


# (neighbour projection tensor, reference projection tensor) -> proj_rel on the CPU, filled by `collect_proj_for_scene`:
# the patched MVSDet.collect_proj evaluates the k products of a scene ONCE, from one device-to-host copy of the cameras;
# `homo_warping` then finds its matrix by the identity of the two tensors it is handed (the entries keep them alive, so
# an id is never reused while it is in the table).
_SCENE_PROJ: dict = {}


def amp_fp32(*tensors):
    """`--amp` (tools/train.py:24-28; mmengine's AmpOptimWrapper = torch.autocast around the forward pass): while autocast is on,
    low-precision floating tensors become float32 -- the hot path computes in float32, as torch's own autocast rule for
    `grid_sampler` / `softmax` has the reference do at these very calls.  Outside autocast nothing is touched: a float16 tensor
    handed to the float32 operators then raises their TypeError (the fp16-storage sweep is a separate, explicit entry)."""
    if not torch.is_autocast_enabled("cuda"):
        return tensors if len(tensors) != 1 else tensors[0]
    out = tuple(t.float() if isinstance(t, Tensor) and t.is_floating_point() and t.dtype != torch.float32 else t for t in tensors)
    return out if len(out) != 1 else out[0]


def relative_projection(src_proj: Tensor, ref_proj: Tensor) -> Tensor:
    """module.py:116  proj = src_proj @ inverse(ref_proj), fp32, evaluated with ATen-CPU; result on CPU."""
    hit = _SCENE_PROJ.get((id(src_proj), id(ref_proj)))
    if hit is not None and hit[0] is src_proj and hit[1] is ref_proj:
        return hit[2]
    return torch.matmul(src_proj.detach().float().cpu(), torch.inverse(ref_proj.detach().float().cpu()))


def collect_proj_for_scene(w2c: Tensor, intr: Tensor, neighbor_ids: Tensor):
    """MVSDet.collect_proj (mvsdet.py:249-264) for the patched reference: same return values on the callers' device, and
    -- from ONE host copy of `w2c` / `intr` -- the k matrices `nei_proj @ inverse(ref_proj)` (module.py:116) that the k
    `homo_warping` calls of the scene will ask for, so those cost no further synchronisation and no repeated inverse."""
    dev = w2c.device
    w2c_h, intr_h, ids_h = w2c.detach().float().cpu(), intr.detach().float().cpu(), neighbor_ids.detach().cpu()
    proj_h, nei_h = collect_proj(w2c_h, intr_h, ids_h)
    inv_ref = torch.inverse(proj_h)
    proj = proj_h.to(dev, non_blocking=True) if dev.type != "cpu" else proj_h
    nei = tuple((n.to(dev, non_blocking=True) if dev.type != "cpu" else n) for n in nei_h)
    _SCENE_PROJ.clear()                      # one scene at a time: the reference's loop is sequential
    for n_dev, n_host in zip(nei, nei_h):
        _SCENE_PROJ[(id(n_dev), id(proj))] = (n_dev, proj, torch.matmul(n_host, inv_ref))
    return proj, nei


def knn(x: Tensor, ref: Tensor, k: int, maskself: bool = False) -> Tensor:
    """mvsdet.py:43-64.  x (B,3,Ns), ref (B,3,Nr) -> indices (B,Ns,k) of the k nearest `ref` columns
    (largest negative squared distance first)."""
    cross = -2 * torch.matmul(x.transpose(2, 1), ref)
    x_sq = torch.sum(x ** 2, dim=1, keepdim=True)
    r_sq = torch.sum(ref ** 2, dim=1, keepdim=True)
    neg_dist = -r_sq - cross - x_sq.transpose(2, 1)
    if maskself:
        if x.shape != ref.shape:
            raise AssertionError("maskself needs x and ref of the same shape")
        diag = torch.arange(x_sq.shape[2])
        neg_dist[:, diag, diag] = -100000
    return neg_dist.topk(k=k, dim=-1)[1]


def get_nearest_pose_ids(tar_pose: Tensor, ref_poses: Tensor, num_select: int, maskself: bool = False,
                         angular_dist_method: str = "dist", scene_center=(0, 0, 0)) -> Tensor:
    """mvsdet.py:67-104 ('dist' method, the only one the detector uses).  c2w poses (N,4,4) -> (N,k) int64."""
    if angular_dist_method != "dist":
        raise NotImplementedError("only angular_dist_method='dist' is on the MVSDet path (mvsdet.py:434)")
    num_select = min(num_select, len(ref_poses) - 1)
    tar = tar_pose[:, :3, 3].unsqueeze(0).transpose(2, 1)
    ref = ref_poses[:, :3, 3].unsqueeze(0).transpose(2, 1)
    ids = knn(tar, ref, k=num_select, maskself=maskself)[0]
    if LAZY_WARP:
        from . import lazywarp
        lazywarp.note_neighbor_ids(ids)
    return ids


def collect_proj(w2c: Tensor, intr: Tensor, neighbor_ids: Tensor):
    """MVSDet.collect_proj, mvsdet.py:249-264 -> (proj (N,4,4), tuple of k (N,4,4) neighbour projections)."""
    if intr.dim() == 2:
        intr = intr.unsqueeze(0).repeat(w2c.shape[0], 1, 1)
    proj = torch.matmul(intr, w2c)
    n, k = neighbor_ids.shape
    nei = proj[neighbor_ids.reshape(-1)].view(n, k, 4, 4)
    return proj, torch.unbind(nei, dim=1)


def get_points(n_voxels: Tensor, voxel_size: Tensor, origin: Tensor) -> Tensor:
    """mvsdet.py:1316-1327 -> voxel corner coordinates (3,X,Y,Z)."""
    with torch.no_grad():
        grid = torch.stack(torch.meshgrid([torch.arange(int(n_voxels[0])), torch.arange(int(n_voxels[1])),
                                           torch.arange(int(n_voxels[2]))], indexing="ij"))
        new_origin = origin - n_voxels / 2. * voxel_size
        return grid * voxel_size.view(3, 1, 1, 1) + new_origin.view(3, 1, 1, 1)


def compute_projection(img_meta: dict, stride: int, angles=None) -> Tensor:
    """MVSDet._compute_projection, mvsdet.py:1124-1156 (angles=None) -> (N,3,4) voxel->feature-pixel matrices."""
    if angles is not None:
        raise NotImplementedError("predicted-angle extrinsics (SUNRGBDTotal) are outside the MVSDet configs")
    ratio = img_meta["ori_shape"][0] / (img_meta["img_shape"][0] / stride)
    extr = torch.tensor(np.array(img_meta["lidar2img"]["extrinsic"]))
    intr = torch.tensor(np.array(img_meta["lidar2img"]["intrinsic"]))
    if intr.dim() == 2:  # ScanNet: one K per scene
        K = intr[:3, :3].clone()
        K[:2] /= ratio
        return torch.stack([K @ e[:3] for e in extr])
    K = intr[:, :3, :3].clone()  # ARKitScenes: one K per view
    K[:, :2] /= ratio
    return torch.stack([K[i] @ extr[i][:3] for i in range(len(extr))])


# ------------------------------------------------------------------------------------------- a3
def homo_warping(src_fea: Tensor, src_proj: Tensor, ref_proj: Tensor, depth_values: Tensor) -> Tensor:
    """mvs_models/module.py:105-146.  src_fea (B,C,H,W), src_proj/ref_proj (B,4,4), depth_values (B,D)
    -> warped (B,C,D,H,W)."""
    if depth_values.dim() != 2:
        raise NotImplementedError("per-pixel depth_values (B,D,H,W) (module.py:130-133) is unused by MVSDet")
    src_fea, depth_values = amp_fp32(src_fea, depth_values)
    proj = relative_projection(src_proj, ref_proj).to(src_fea.device)
    depth_values = depth_values.to(src_fea.device)
    if LAZY_WARP and (src_fea.is_cuda or LAZY_ANY_DEVICE) and src_fea.dtype == torch.float32:
        # inside the patched reference: defer, so that its variance loop collapses into the fused kernel (lazywarp.py)
        from . import lazywarp
        return lazywarp.lazy_homo_warp(src_fea, proj, depth_values)
    return ops.homo_warp(src_fea, proj, depth_values)


# ------------------------------------------------------------------------------------------- a9
def backproject_Weigh(features: Tensor, points: Tensor, projection: Tensor, depth: Tensor, voxel_size: Sequence[float],
                      prob: Tensor, gt_depth=None, save_dir=None, img_meta=None, depth_mean=None):
    """mvsdet.py:1372-1492.  features (N,C,h,w); points (3,X,Y,Z); projection (N,3,4);
    depth, prob (N, h*w, 1, J) -> (volume (N,C,X,Y,Z), valid (N,1,X,Y,Z) bool, gap_all, rmse)."""
    if gt_depth is not None:
        raise NotImplementedError("the gt_depth debug branch (mvsdet.py:1435-1481) is outside the hot path")
    features, points, projection, depth, prob = amp_fp32(features, points, projection, depth, prob)
    n, c, h, w = features.shape
    nx, ny, nz = points.shape[-3:]
    j = depth.shape[-1] * depth.shape[-2]
    # (N, h*w, 1, J) is a view of (N,J,h,w): hand the kernel that view, no copy (mvsdet.py:1393-1395)
    est_depth = depth.reshape(n, h, w, j).permute(0, 3, 1, 2)
    est_dens = prob.reshape(n, h, w, j).permute(0, 3, 1, 2)
    if LAZY_WARP and (features.is_cuda or LAZY_ANY_DEVICE) and features.dtype == torch.float32:
        # inside the patched reference: `volume.sum(dim=0)` / `valid.sum(dim=0)` (mvsdet.py:509-511) then cost one launch
        # of the fused lifting kernel instead of a (N,C,X,Y,Z) volume and a reduction over it (lazywarp.py)
        from . import lazywarp
        volume, valid = lazywarp.lazy_backproject(features, points, projection, est_depth, est_dens, float(voxel_size[-1]))
        return volume, valid, torch.tensor(1.), torch.tensor(1.)
    volume, valid = ops.backproject_weigh(features, points, projection, est_depth, est_dens, float(voxel_size[-1]))
    volume = volume.view(n, c, nx, ny, nz)
    valid = valid.view(n, 1, nx, ny, nz)
    return volume, valid, torch.tensor(1.), torch.tensor(1.)
