"""ctypes front-end of the CPU oracle + numpy restatement of the host-side geometry.

TEST INFRASTRUCTURE (see planesweep_oracle.c).  Imported only by tests/, bench.py's
cpu_baseline leg and __graft_entry__.smoke().  Arrays in, arrays out (numpy, fp32).

Reference citations are relative to projects/NeRF-Det/nerfdet/ of Pixie8888/MVSDet.
"""
from __future__ import annotations

import ctypes
import os
import subprocess

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
_LIB_PATH = os.path.join(_HERE, "libplanesweep_oracle.so")
_lib = None

_f = ctypes.POINTER(ctypes.c_float)
_i64 = ctypes.POINTER(ctypes.c_int64)
_i32 = ctypes.POINTER(ctypes.c_int32)
_u8 = ctypes.POINTER(ctypes.c_uint8)


def build(force: bool = False) -> str:
    """gcc-compile the oracle next to its source (no-op when up to date)."""
    src = os.path.join(_HERE, "planesweep_oracle.c")
    if force or not os.path.exists(_LIB_PATH) or os.path.getmtime(_LIB_PATH) < os.path.getmtime(src):
        subprocess.check_call(["make", "-C", _HERE, "-B", "libplanesweep_oracle.so"], stdout=subprocess.DEVNULL)
    return _LIB_PATH


def lib():
    global _lib
    if _lib is None:
        if not os.path.exists(_LIB_PATH):
            build()
        _lib = ctypes.CDLL(_LIB_PATH)
        _lib.orc_num_threads.restype = ctypes.c_int
    return _lib


def num_threads() -> int:
    return lib().orc_num_threads()


def set_num_threads(n: int):
    lib().orc_set_num_threads(int(n))


def _np(a, dtype=np.float32):
    if hasattr(a, "detach"):
        a = a.detach().cpu().numpy()
    return np.ascontiguousarray(a, dtype=dtype)


def _p(a, t):
    return a.ctypes.data_as(t) if a is not None else None


# --------------------------------------------------------------------------- host-side geometry (numpy)
def depth_planes(near, far, D):
    """mvsdet.py:221-225."""
    interval = (far - near) / D
    dv = np.arange(near, far, interval, dtype=np.float32)
    assert len(dv) == D
    return dv


def knn_neighbors(c2w, k, maskself=True):
    """knn / get_nearest_pose_ids, mvsdet.py:43-104 (angular_dist_method='dist').
    Ties go to the lower view index."""
    c2w = _np(c2w)
    n = c2w.shape[0]
    k = min(k, n - 1)
    x = c2w[:, :3, 3].T.copy()  # (3,N)
    inner = np.float32(-2) * (x.T @ x)
    xx = np.sum(x ** 2, axis=0, keepdims=True)  # (1,N)
    pd = -xx - inner - xx.T
    if maskself:
        pd[np.arange(n), np.arange(n)] = -100000
    return np.argsort(-pd, axis=-1, kind="stable")[:, :k].astype(np.int64)


def feat_intrinsics(intrinsic, img_shape, ori_shape, stride=4):
    """mvsdet.py:423-428: K_feat = K; K_feat[:2] /= ratio (ratio from the un-padded height)."""
    K = _np(intrinsic).copy()
    ratio = ori_shape[0] / (img_shape[0] / stride)
    if K.ndim == 2:
        K[:2] /= np.float32(ratio)
    else:
        K[:, :2] /= np.float32(ratio)
    return K


def relative_projections(extrinsic, K_feat, nbr):
    """collect_proj (mvsdet.py:249-264) followed by module.py:116:
    proj[n,j] = (K_feat @ w2c)[nbr[n,j]] @ inverse((K_feat @ w2c)[n]), all fp32."""
    w2c = _np(extrinsic)
    K = _np(K_feat)
    if K.ndim == 2:
        K = np.repeat(K[None], w2c.shape[0], 0)
    proj = np.matmul(K, w2c).astype(np.float32)
    inv = np.linalg.inv(proj).astype(np.float32)
    return np.matmul(proj[nbr], inv[:, None]).astype(np.float32), proj


def compute_projection(extrinsic, intrinsic, img_shape, ori_shape, stride=4):
    """_compute_projection, mvsdet.py:1124-1156 (angles=None): (N,3,4) = K'[:3,:3] @ E[:3]."""
    E = _np(extrinsic)
    K = feat_intrinsics(intrinsic, img_shape, ori_shape, stride)
    if K.ndim == 2:
        K = np.repeat(K[None], E.shape[0], 0)
    return np.matmul(K[:, :3, :3], E[:, :3]).astype(np.float32)


def get_points(n_voxels, voxel_size, origin):
    """get_points, mvsdet.py:1316-1327: voxel *corner* coordinates, (3,X,Y,Z) fp32."""
    nv = np.asarray(n_voxels, dtype=np.int64)
    vs = np.asarray(voxel_size, dtype=np.float32)
    og = np.asarray(origin, dtype=np.float32)
    idx = np.stack(np.meshgrid(np.arange(nv[0]), np.arange(nv[1]), np.arange(nv[2]), indexing="ij"))
    new_origin = og - (nv.astype(np.float32) / np.float32(2.0)) * vs
    pts = idx.astype(np.float32) * vs.reshape(3, 1, 1, 1)
    return (pts + new_origin.reshape(3, 1, 1, 1)).astype(np.float32)


# --------------------------------------------------------------------------- C kernels
def homo_warp(src, proj, depth):
    src, proj, depth = _np(src), _np(proj), _np(depth)
    B, C, H, W = src.shape
    D = depth.shape[1]
    out = np.empty((B, C, D, H, W), np.float32)
    lib().orc_homo_warp(_p(src, _f), _p(proj, _f), _p(depth, _f), _p(out, _f), B, C, D, H, W)
    return out


def plane_sweep_variance(feat, nbr, proj, depth, mode=0):
    feat, proj, depth = _np(feat), _np(proj), _np(depth)
    nbr = _np(nbr, np.int64)
    N, C, H, W = feat.shape
    K = nbr.shape[1]
    D = depth.shape[1]
    assert proj.shape == (N, K, 4, 4) and depth.shape[0] == N
    out = np.empty((N, C, D, H, W), np.float32)
    rc = lib().orc_plane_sweep_variance(_p(feat, _f), _p(nbr, _i64), _p(proj, _f), _p(depth, _f), _p(out, _f),
                                        N, K, C, D, H, W, mode)
    if rc:
        raise ValueError(f"orc_plane_sweep_variance rc={rc}")
    return out


def plane_sweep_variance_bwd(feat, nbr, proj, depth, g):
    feat, proj, depth, g = _np(feat), _np(proj), _np(depth), _np(g)
    nbr = _np(nbr, np.int64)
    N, C, H, W = feat.shape
    K, D = nbr.shape[1], depth.shape[1]
    out = np.empty_like(feat)
    rc = lib().orc_plane_sweep_variance_bwd(_p(feat, _f), _p(nbr, _i64), _p(proj, _f), _p(depth, _f), _p(g, _f),
                                            _p(out, _f), N, K, C, D, H, W)
    if rc:
        raise ValueError(f"orc_plane_sweep_variance_bwd rc={rc}")
    return out


def depth_prob_topk(cost_reg, off_logit, near, interval, topk=3):
    cost_reg, off_logit = _np(cost_reg), _np(off_logit)
    N, D, H, W = cost_reg.shape
    prob = np.empty_like(cost_reg)
    off = np.empty_like(cost_reg)
    est_depth = np.empty((N, topk, H, W), np.float32)
    est_dens = np.empty((N, topk, H, W), np.float32)
    est_idx = np.empty((N, topk, H, W), np.int32)
    avg = np.empty((N, H, W), np.float32)
    rc = lib().orc_depth_prob_topk(_p(cost_reg, _f), _p(off_logit, _f), _p(prob, _f), _p(off, _f), _p(est_depth, _f),
                                   _p(est_dens, _f), _p(est_idx, _i32), _p(avg, _f), N, D, H, W, topk,
                                   ctypes.c_float(np.float32(near)), ctypes.c_float(np.float32(interval)))
    if rc:
        raise ValueError(f"orc_depth_prob_topk rc={rc}")
    return dict(prob=prob, off=off, est_depth=est_depth, est_dens=est_dens, est_idx=est_idx, avg_depth=avg)


def depth_prob_topk_bwd(prob, off, est_idx, g_prob, g_depth, g_dens, g_avg, near, interval):
    prob, off = _np(prob), _np(off)
    est_idx = _np(est_idx, np.int32)
    N, D, H, W = prob.shape
    topk = est_idx.shape[1]
    gs = [None if g is None else _np(g) for g in (g_prob, g_depth, g_dens, g_avg)]
    g_cost = np.empty_like(prob)
    g_off = np.empty_like(prob)
    lib().orc_depth_prob_topk_bwd(_p(prob, _f), _p(off, _f), _p(est_idx, _i32), _p(gs[0], _f), _p(gs[1], _f),
                                  _p(gs[2], _f), _p(gs[3], _f), _p(g_cost, _f), _p(g_off, _f), N, D, H, W, topk,
                                  ctypes.c_float(np.float32(near)), ctypes.c_float(np.float32(interval)))
    return g_cost, g_off


def _strides_elems(a):
    return np.array([s // a.itemsize for s in a.strides], dtype=np.int64)


def _stage3_args(features, points, projection, est_depth, est_dens):
    # features may be a non-contiguous crop: keep its strides (as the reference does, mvsdet.py:499)
    if hasattr(features, "detach"):
        features = features.detach().cpu().numpy()
    features = np.asarray(features, dtype=np.float32)
    fs = _strides_elems(features)
    points = _np(points).reshape(3, -1)
    projection = _np(projection)
    est_depth = _np(est_depth)
    est_dens = _np(est_dens)
    assert est_depth.shape == est_dens.shape
    ds = _strides_elems(est_depth)
    return features, fs, points, projection, est_depth, est_dens, ds


def backproject_weigh(features, points, projection, est_depth, est_dens, vz, want_index=False):
    """est_depth/est_dens: (N,J,h,w) (the reference's (N,h*w,1,J) is this tensor transposed)."""
    features, fs, points, projection, est_depth, est_dens, ds = _stage3_args(features, points, projection, est_depth, est_dens)
    N, C, h, w = features.shape
    V = points.shape[1]
    J = est_depth.shape[1]
    volume = np.empty((N, C, V), np.float32)
    valid = np.empty((N, V), np.uint8)
    xi = np.empty((N, V), np.int32) if want_index else None
    yi = np.empty((N, V), np.int32) if want_index else None
    z = np.empty((N, V), np.float32) if want_index else None
    lib().orc_backproject_weigh(features.ctypes.data_as(_f), _p(fs, _i64), _p(points, _f), _p(projection, _f),
                                _p(est_depth, _f), _p(est_dens, _f), _p(ds, _i64), _p(volume, _f), _p(valid, _u8),
                                _p(xi, _i32), _p(yi, _i32), _p(z, _f), N, C, h, w, V, J, ctypes.c_float(np.float32(vz)))
    out = dict(volume=volume, valid=valid.astype(bool))
    if want_index:
        out.update(x=xi, y=yi, z=z)
    return out


def backproject_weigh_mean(features, points, projection, est_depth, est_dens, vz):
    features, fs, points, projection, est_depth, est_dens, ds = _stage3_args(features, points, projection, est_depth, est_dens)
    N, C, h, w = features.shape
    V = points.shape[1]
    J = est_depth.shape[1]
    mean = np.empty((C, V), np.float32)
    count = np.empty((V,), np.int32)
    lib().orc_backproject_weigh_mean(features.ctypes.data_as(_f), _p(fs, _i64), _p(points, _f), _p(projection, _f),
                                     _p(est_depth, _f), _p(est_dens, _f), _p(ds, _i64), _p(mean, _f), _p(count, _i32),
                                     N, C, h, w, V, J, ctypes.c_float(np.float32(vz)))
    return dict(volume_mean=mean, valid_count=count)


def backproject_weigh_bwd(features, points, projection, est_depth, est_dens, vz, g):
    features, fs, points, projection, est_depth, est_dens, ds = _stage3_args(features, points, projection, est_depth, est_dens)
    N, C, h, w = features.shape
    V = points.shape[1]
    J = est_depth.shape[1]
    g = _np(g).reshape(N, C, V)
    gfeat = np.empty((N, C, h, w), np.float32)
    gdens = np.empty((N, J, h, w), np.float32)
    rc = lib().orc_backproject_weigh_bwd(features.ctypes.data_as(_f), _p(fs, _i64), _p(points, _f), _p(projection, _f),
                                         _p(est_depth, _f), _p(est_dens, _f), _p(ds, _i64), _p(g, _f), _p(gfeat, _f),
                                         _p(gdens, _f), N, C, h, w, V, J, ctypes.c_float(np.float32(vz)))
    if rc:
        raise ValueError(f"orc_backproject_weigh_bwd rc={rc}")
    return gfeat, gdens
