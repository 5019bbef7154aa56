"""Seeded synthetic ScanNet / ARKitScenes shaped scenes.

There is no dataset on the GPU box, so benchmarks and parity tests run on
scenes built here.  A scene carries exactly what the reference's
``MVSDet.extract_feat`` reads from ``img_meta`` (mvsdet.py:407-428):

    img_meta['lidar2img'] = {'extrinsic': [N x (4,4) w2c], 'intrinsic': (4,4) | [N x (4,4)],
                             'origin': (3,)}
    img_meta['img_shape'] = (4*Hf - 1, 4*Wf)      # resized, un-padded (239, 320 on ScanNet)
    img_meta['ori_shape'] = (968, 1296)

Cameras stand inside a 6 m x 6 m room at 1.0-1.6 m height, walk in a slow arc
(consecutive yaw steps <= 15 degrees, pitch within +-20 degrees) so that the two
nearest cameras of every view overlap with it, as ScanNet trajectories do.
All randomness comes from ``numpy.random.default_rng(seed)`` (PCG64, identical
on every platform) so fixtures generated in one container reproduce elsewhere.
"""
from __future__ import annotations

import numpy as np
import torch

SCANNET_K = np.array([[1170.0, 0.0, 648.0, 0.0],
                      [0.0, 1170.0, 484.0, 0.0],
                      [0.0, 0.0, 1.0, 0.0],
                      [0.0, 0.0, 0.0, 1.0]], dtype=np.float32)
ORI_SHAPE = (968, 1296)


def _look_at_c2w(pos, yaw, pitch):
    """OpenCV camera (x right, y down, z forward) -> world (z up)."""
    fwd = np.array([np.cos(yaw) * np.cos(pitch), np.sin(yaw) * np.cos(pitch), np.sin(pitch)])
    up = np.array([0.0, 0.0, 1.0])
    right = np.cross(fwd, up)
    right /= np.linalg.norm(right)
    down = np.cross(fwd, right)
    c2w = np.eye(4)
    c2w[:3, 0] = right
    c2w[:3, 1] = down
    c2w[:3, 2] = fwd
    c2w[:3, 3] = pos
    return c2w


def make_cameras(n_views: int, seed: int = 0, per_view_intrinsics: bool = False):
    """Returns (w2c (N,4,4) float32, K (4,4) or (N,4,4) float32)."""
    rng = np.random.default_rng(seed)
    yaw = rng.uniform(0, 2 * np.pi)
    pos = np.array([rng.uniform(-1.0, 1.0), rng.uniform(-1.0, 1.0), rng.uniform(1.0, 1.6)])
    w2c = []
    for _ in range(n_views):
        yaw = yaw + np.deg2rad(rng.uniform(4.0, 15.0))
        pitch = np.deg2rad(rng.uniform(-20.0, 5.0))
        step = rng.uniform(0.05, 0.30)
        pos = pos + step * np.array([np.cos(yaw + np.pi / 2), np.sin(yaw + np.pi / 2), 0.0])
        pos[:2] = np.clip(pos[:2], -2.5, 2.5)
        pos[2] = np.clip(pos[2] + rng.uniform(-0.05, 0.05), 1.0, 1.6)
        c2w = _look_at_c2w(pos.copy(), yaw, pitch)
        w2c.append(np.linalg.inv(c2w))
    w2c = np.stack(w2c).astype(np.float32)
    if per_view_intrinsics:
        K = np.repeat(SCANNET_K[None], n_views, 0).copy()
        jit = rng.uniform(0.95, 1.05, size=(n_views, 2)).astype(np.float32)
        K[:, 0, 0] *= jit[:, 0]
        K[:, 1, 1] *= jit[:, 0]
        K[:, 0, 2] *= jit[:, 1]
        K[:, 1, 2] *= jit[:, 1]
    else:
        K = SCANNET_K.copy()
    return w2c, K


def make_img_meta(n_views: int, feat_hw=(60, 80), seed: int = 0, per_view_intrinsics: bool = False,
                  origin=(0.0, 0.0, 0.5)):
    """The dict the reference reads in extract_feat (mvsdet.py:407-428)."""
    w2c, K = make_cameras(n_views, seed, per_view_intrinsics)
    hf, wf = feat_hw
    meta = {
        "lidar2img": {
            "extrinsic": [w2c[i] for i in range(n_views)],
            "intrinsic": [K[i] for i in range(n_views)] if per_view_intrinsics else K,
            "origin": np.asarray(origin, dtype=np.float32),
        },
        "img_shape": (4 * hf - 1, 4 * wf),
        "ori_shape": ORI_SHAPE,
    }
    return meta


def make_features(n_views: int, channels: int, feat_hw=(60, 80), seed: int = 0, device="cpu",
                  dtype=torch.float32):
    """feat ~ N(0,1), (N,C,Hf,Wf).  CPU generator for parity tests (same values on every
    machine); device generator for full-size benchmark inputs."""
    hf, wf = feat_hw
    g = torch.Generator(device=device)
    g.manual_seed(1000 + seed)
    return torch.randn((n_views, channels, hf, wf), generator=g, device=device, dtype=dtype)


def make_cost_logits(n_views: int, n_depth: int, feat_hw=(60, 80), seed: int = 0, device="cpu",
                     sharp: float = 3.0):
    """Stand-in for CostRegNet_3DGS output (N,2,D,Hf,Wf): cost logits and offset logits.
    ``sharp`` scales the cost logits so the soft-max has a clear ranking (no top-k ties)."""
    hf, wf = feat_hw
    g = torch.Generator(device=device)
    g.manual_seed(2000 + seed)
    out = torch.randn((n_views, 2, n_depth, hf, wf), generator=g, device=device)
    out[:, 0] *= sharp
    return out


class Scene:
    """Bundle of one synthetic scene."""

    def __init__(self, n_views, channels, n_depth, feat_hw=(60, 80), seed=0, device="cpu",
                 per_view_intrinsics=False, near_far=(0.2, 5.0)):
        self.n_views, self.channels, self.n_depth = n_views, channels, n_depth
        self.feat_hw = feat_hw
        self.near_far = near_far
        self.img_meta = make_img_meta(n_views, feat_hw, seed, per_view_intrinsics)
        self.features = make_features(n_views, channels, feat_hw, seed, device)
        self.cost_logits = make_cost_logits(n_views, n_depth, feat_hw, seed, device)
