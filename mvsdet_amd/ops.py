"""PyTorch-ROCm custom operators over the C ABI (torch.ops.mvsdet_amd.*).

Every operator enqueues hand-written gfx950 kernels of libmvsdet_hip.so on the current HIP stream
through ctypes; torch supplies device memory, streams and autograd plumbing only.  Operators raise if
their inputs are not float32 tensors on a ROCm device -- there is no CPU implementation here.
"""
from __future__ import annotations

from typing import Optional, Tuple

import torch
from torch import Tensor

from . import _lib

_NS = "mvsdet_amd"


def _req(t: Tensor, name: str, dtype=torch.float32, dim=None):
    if not t.is_cuda:
        raise RuntimeError(f"mvsdet_amd: `{name}` must live on a ROCm device (got {t.device}); "
                           "this package has no CPU path")
    if t.dtype != dtype:
        raise TypeError(f"mvsdet_amd: `{name}` must be {dtype} (got {t.dtype})")
    if dim is not None and t.dim() != dim:
        raise ValueError(f"mvsdet_amd: `{name}` must be {dim}-D (got shape {tuple(t.shape)})")


def _stream(t: Tensor):
    return _lib.current_stream(t.device)


# ------------------------------------------------------------------------------------------- pack
@torch.library.custom_op(f"{_NS}::pack_features", mutates_args=(), device_types="cuda")
def pack_features(feat: Tensor) -> Tensor:
    """(N,C,H,W) float32 or float16 (any strides) -> packed channel-last maps, flat float32
    (see include/mvsdet_hip.h; the fp16 -> fp32 conversion is exact)."""
    half = feat.dtype == torch.float16
    _req(feat, "feat", dtype=torch.float16 if half else torch.float32, dim=4)
    N, C, H, W = feat.shape
    lib = _lib.load()
    out = torch.empty(lib.mvsdet_packed_bytes(N, C, H, W) // 4, dtype=torch.float32, device=feat.device)
    fn = lib.mvsdet_pack_features_f16 if half else lib.mvsdet_pack_features_f32
    with torch.cuda.device(feat.device):
        _lib.check(fn(_lib.ptr(feat), _lib.strides4(feat), _lib.ptr(out), N, C, H, W, _stream(feat)), "pack_features")
    return out


@pack_features.register_fake
def _(feat):
    N, C, H, W = feat.shape
    return feat.new_empty(N * H * W * 32 * ((C + 31) // 32), dtype=torch.float32)


# ------------------------------------------------------------------------------------------- a3
@torch.library.custom_op(f"{_NS}::homo_warp", mutates_args=(), device_types="cuda")
def homo_warp(src: Tensor, proj: Tensor, depth: Tensor) -> Tensor:
    """a3 (mvs_models/module.py:105): src (B,C,H,W), proj (B,4,4) = src_proj @ inv(ref_proj), depth (B,D)."""
    _req(src, "src", dim=4)
    _req(proj, "proj", dim=3)
    _req(depth, "depth", dim=2)
    B, C, H, W = src.shape
    D = depth.shape[1]
    if proj.shape != (B, 4, 4) or depth.shape[0] != B:
        raise ValueError(f"homo_warp: proj {tuple(proj.shape)} / depth {tuple(depth.shape)} do not match B={B}")
    src, proj, depth = src.contiguous(), proj.contiguous(), depth.contiguous()
    out = torch.empty((B, C, D, H, W), dtype=torch.float32, device=src.device)
    with torch.cuda.device(src.device):
        _lib.check(_lib.load().mvsdet_homo_warp_f32(_lib.ptr(src), _lib.ptr(proj), _lib.ptr(depth), _lib.ptr(out),
                                                    B, C, D, H, W, _stream(src)), "homo_warp")
    return out


@homo_warp.register_fake
def _(src, proj, depth):
    B, C, H, W = src.shape
    return src.new_empty((B, C, depth.shape[1], H, W))


# ------------------------------------------------------------------------------------------- a3+a4
def ray_depth(intr: Tensor, est_depth: Optional[Tensor], h: int, w: int):
    """mvsdet.py:1158-1216 + :494: intr (N,5) {fx,fy,cx,cy,skew} at feature level -> depth_scale (N,h*w,1) and, when
    est_depth (N,J,H,W) is given, est_ray_depth (N,h*w,1,J) in the layout extract_feat hands to the Gaussian adapter."""
    _req(intr, "intr", dim=2)
    N = intr.shape[0]
    intr = intr.contiguous()
    scale = torch.empty((N, h * w, 1), dtype=torch.float32, device=intr.device)
    ray = None
    J, H, W = 0, h, w
    if est_depth is not None:
        _req(est_depth, "est_depth", dim=4)
        if est_depth.shape[0] != N:
            raise ValueError("ray_depth: est_depth and intr disagree on the number of views")
        est_depth = est_depth.contiguous()
        J, H, W = est_depth.shape[1:]
        ray = torch.empty((N, J, h * w), dtype=torch.float32, device=intr.device)
    with torch.cuda.device(intr.device):
        _lib.check(_lib.load().mvsdet_ray_depth_f32(_lib.ptr(intr), _lib.ptr(est_depth), _lib.ptr(scale), _lib.ptr(ray), N, J, H, W,
                                                    h, w, _stream(intr)), "ray_depth")
    return scale, (None if ray is None else ray.transpose(2, 1).unsqueeze(2))


def validate_neighbors(nbr: Tensor, n_src: int) -> None:
    """Host check of neighbour view ids (include/mvsdet_hip.h: mvsdet_validate_neighbors): the reference's index gather
    (mvsdet.py:440) raises on an id outside [0, N); the kernels only clamp.  `nbr` must be a CPU tensor."""
    if nbr.device.type != "cpu":
        raise ValueError("validate_neighbors: the ids are checked on their host copy")
    t = nbr.to(torch.int64).contiguous()
    m, k = (t.shape[0], t.shape[1]) if t.dim() == 2 else (t.numel(), 1)
    import ctypes
    p = ctypes.cast(t.data_ptr(), ctypes.POINTER(ctypes.c_int64)) if t.numel() else None
    _lib.check(_lib.load().mvsdet_validate_neighbors(p, int(m), int(k), int(n_src)), "validate_neighbors")


def _check_sweep(feat, nbr, proj, depth):
    _req(feat, "feat", dim=4)
    _req(nbr, "nbr", dtype=torch.int64, dim=2)
    _req(proj, "proj", dim=4)
    _req(depth, "depth", dim=2)
    N, C, H, W = feat.shape
    K = nbr.shape[1]
    if nbr.shape[0] != N or proj.shape != (N, K, 4, 4) or depth.shape[0] != N:
        raise ValueError(f"plane_sweep_variance: nbr {tuple(nbr.shape)}, proj {tuple(proj.shape)}, depth "
                         f"{tuple(depth.shape)} do not match N={N}")
    return N, K, C, depth.shape[1], H, W


@torch.library.custom_op(f"{_NS}::plane_sweep_variance_packed", mutates_args=(), device_types="cuda")
def plane_sweep_variance_packed(packed: Tensor, nbr: Tensor, proj: Tensor, depth: Tensor, C: int, H: int,
                                W: int) -> Tensor:
    """a3+a4 on already packed maps (mvsdet.py:439-467) -> (N,C,D,H,W)."""
    _req(packed, "packed", dim=1)
    _req(nbr, "nbr", dtype=torch.int64, dim=2)
    _req(proj, "proj", dim=4)
    _req(depth, "depth", dim=2)
    N, K = nbr.shape
    D = depth.shape[1]
    lib = _lib.load()
    if packed.numel() * 4 != lib.mvsdet_packed_bytes(N, C, H, W):
        raise ValueError("plane_sweep_variance_packed: packed buffer does not match (N,C,H,W)")
    if proj.shape != (N, K, 4, 4) or depth.shape[0] != N:
        raise ValueError("plane_sweep_variance_packed: proj/depth shape mismatch")
    nbr, proj, depth = nbr.contiguous(), proj.contiguous(), depth.contiguous()
    out = torch.empty((N, C, D, H, W), dtype=torch.float32, device=packed.device)
    # channel-independent sampling table (8 B per view, neighbour, plane, pixel), built by the op's first kernel
    sbytes = lib.mvsdet_plane_sweep_scratch_bytes(N, K, D, H, W)
    scratch = torch.empty(max(sbytes // 4, 4), dtype=torch.float32, device=packed.device)
    with torch.cuda.device(packed.device):
        _lib.check(lib.mvsdet_plane_sweep_variance_packed_f32(_lib.ptr(packed), _lib.ptr(nbr), _lib.ptr(proj),
                                                              _lib.ptr(depth), _lib.ptr(out), _lib.ptr(scratch), sbytes,
                                                              N, K, C, D, H, W, _stream(packed)),
                   "plane_sweep_variance_packed")
    return out


@plane_sweep_variance_packed.register_fake
def _(packed, nbr, proj, depth, C, H, W):
    return packed.new_empty((nbr.shape[0], C, depth.shape[1], H, W))


@torch.library.custom_op(f"{_NS}::plane_sweep_variance_shard", mutates_args=(), device_types="cuda")
def plane_sweep_variance_shard(packed: Tensor, nbr: Tensor, proj: Tensor, depth: Tensor, n_src: int, ref_first: int,
                               C: int, H: int, W: int, half_out: bool = False) -> Tensor:
    """a3+a4 for the reference views ref_first .. ref_first+M-1 of a scene whose n_src views are all in `packed`
    (intra-scene view sharding).  nbr (M,K) holds global view ids -> (M,C,D,H,W), bit-identical to the
    same rows of plane_sweep_variance_packed.  half_out=True stores the same fp32 result rounded to float16
    (BASELINE configs[4]).  Forward only."""
    _req(packed, "packed", dim=1)
    _req(nbr, "nbr", dtype=torch.int64, dim=2)
    _req(proj, "proj", dim=4)
    _req(depth, "depth", dim=2)
    M, K = nbr.shape
    D = depth.shape[1]
    lib = _lib.load()
    if packed.numel() * 4 != lib.mvsdet_packed_bytes(n_src, C, H, W):
        raise ValueError("plane_sweep_variance_shard: packed buffer does not match (n_src,C,H,W)")
    if proj.shape != (M, K, 4, 4) or depth.shape[0] != M:
        raise ValueError("plane_sweep_variance_shard: proj/depth shape mismatch")
    if M < 1 or ref_first < 0 or ref_first + M > n_src:
        raise ValueError(f"plane_sweep_variance_shard: views [{ref_first},{ref_first + M}) outside the {n_src} packed views")
    nbr, proj, depth = nbr.contiguous(), proj.contiguous(), depth.contiguous()
    out = torch.empty((M, C, D, H, W), dtype=torch.float16 if half_out else torch.float32, device=packed.device)
    sbytes = lib.mvsdet_plane_sweep_scratch_bytes(M, K, D, H, W)
    scratch = torch.empty(max(sbytes // 4, 4), dtype=torch.float32, device=packed.device)
    fn = lib.mvsdet_plane_sweep_variance_shard_f16 if half_out else lib.mvsdet_plane_sweep_variance_shard_f32
    with torch.cuda.device(packed.device):
        _lib.check(fn(_lib.ptr(packed), _lib.ptr(nbr), _lib.ptr(proj),
                      _lib.ptr(depth), _lib.ptr(out), _lib.ptr(scratch), sbytes,
                      n_src, ref_first, M, K, C, D, H, W, _stream(packed)),
                   "plane_sweep_variance_shard")
    return out


@plane_sweep_variance_shard.register_fake
def _(packed, nbr, proj, depth, n_src, ref_first, C, H, W, half_out=False):
    return packed.new_empty((nbr.shape[0], C, depth.shape[1], H, W), dtype=torch.float16 if half_out else torch.float32)


@torch.library.custom_op(f"{_NS}::plane_sweep_table", mutates_args=(), device_types="cuda")
def plane_sweep_table(proj: Tensor, depth: Tensor, H: int, W: int) -> Tensor:
    """Channel-independent sampling table of a scene (8 B per view, neighbour, plane, pixel): proj (N,K,4,4),
    depth (N,D) -> opaque flat buffer for plane_sweep_variance_tabled."""
    _req(proj, "proj", dim=4)
    _req(depth, "depth", dim=2)
    N, K = proj.shape[:2]
    D = depth.shape[1]
    lib = _lib.load()
    sbytes = lib.mvsdet_plane_sweep_scratch_bytes(N, K, D, H, W)
    table = torch.empty(max(sbytes // 4, 4), dtype=torch.float32, device=proj.device)
    proj, depth = proj.contiguous(), depth.contiguous()
    with torch.cuda.device(proj.device):
        _lib.check(lib.mvsdet_plane_sweep_table_f32(_lib.ptr(proj), _lib.ptr(depth), _lib.ptr(table), sbytes, N, K, D, H, W,
                                                    _stream(proj)), "plane_sweep_table")
    return table


@plane_sweep_table.register_fake
def _(proj, depth, H, W):
    # same size as the real op: the scratch-size query is a host-only function of the shape
    sbytes = _lib.load().mvsdet_plane_sweep_scratch_bytes(proj.shape[0], proj.shape[1], depth.shape[1], H, W)
    return proj.new_empty(max(sbytes // 4, 4))


@torch.library.custom_op(f"{_NS}::plane_sweep_variance_tabled", mutates_args=(), device_types="cuda")
def plane_sweep_variance_tabled(packed: Tensor, nbr: Tensor, table: Tensor, C: int, D: int, H: int, W: int) -> Tensor:
    """The per-channel half of the sweep on a table from plane_sweep_table -> (N,C,D,H,W)."""
    _req(packed, "packed", dim=1)
    _req(nbr, "nbr", dtype=torch.int64, dim=2)
    _req(table, "table", dim=1)
    N, K = nbr.shape
    nbr = nbr.contiguous()
    out = torch.empty((N, C, D, H, W), dtype=torch.float32, device=packed.device)
    with torch.cuda.device(packed.device):
        _lib.check(_lib.load().mvsdet_plane_sweep_variance_tabled_f32(_lib.ptr(packed), _lib.ptr(nbr), _lib.ptr(table),
                                                                      table.numel() * 4, _lib.ptr(out), N, K, C, D, H, W,
                                                                      _stream(packed)), "plane_sweep_variance_tabled")
    return out


@plane_sweep_variance_tabled.register_fake
def _(packed, nbr, table, C, D, H, W):
    return packed.new_empty((nbr.shape[0], C, D, H, W))


def sweep_row_pitch(W: int) -> int:
    """Row pitch (elements) of a cost volume whose rows start on 128-byte lines: W rounded up to a multiple of 32."""
    return (int(W) + 31) // 32 * 32


def plane_sweep_table_pitched(proj: Tensor, depth: Tensor, H: int, W: int, w_pitch: int) -> Tensor:
    """plane_sweep_table for a cost volume with row pitch `w_pitch` (plane_sweep_variance_tabled_pitched)."""
    _req(proj, "proj", dim=4)
    _req(depth, "depth", dim=2)
    N, K = proj.shape[:2]
    D = depth.shape[1]
    lib = _lib.load()
    sbytes = lib.mvsdet_plane_sweep_scratch_bytes(N, K, D, H, W)
    table = torch.empty(max(sbytes // 4, 4), dtype=torch.float32, device=proj.device)
    proj, depth = proj.contiguous(), depth.contiguous()
    with torch.cuda.device(proj.device):
        _lib.check(lib.mvsdet_plane_sweep_table_pitched_f32(_lib.ptr(proj), _lib.ptr(depth), _lib.ptr(table), sbytes, N, K, D, H,
                                                            W, int(w_pitch), _stream(proj)), "plane_sweep_table_pitched")
    return table


def plane_sweep_variance_tabled_pitched(packed: Tensor, nbr: Tensor, table: Tensor, C: int, D: int, H: int, W: int,
                                        w_pitch: int) -> Tensor:
    """The per-channel half of the sweep into a row-PITCHED volume: returns the (N,C,D,H,W) view of an (N,C,D,H,w_pitch)
    buffer -- the values of plane_sweep_variance_tabled bit for bit, rows `w_pitch` elements apart (every row on a 128-byte
    line when w_pitch % 32 == 0: whole-line stores from 32x4 tiles for the 80-wide maps of the shipped configs).
    Forward only; the pad columns are never written."""
    _req(packed, "packed", dim=1)
    _req(nbr, "nbr", dtype=torch.int64, dim=2)
    _req(table, "table", dim=1)
    if w_pitch < W:
        raise ValueError(f"plane_sweep_variance_tabled_pitched: pitch {w_pitch} < W={W}")
    N, K = nbr.shape
    nbr = nbr.contiguous()
    buf = torch.empty((N, C, D, H, int(w_pitch)), dtype=torch.float32, device=packed.device)
    with torch.cuda.device(packed.device):
        _lib.check(_lib.load().mvsdet_plane_sweep_variance_tabled_pitched_f32(
            _lib.ptr(packed), _lib.ptr(nbr), _lib.ptr(table), table.numel() * 4, _lib.ptr(buf), N, K, C, D, H, W, int(w_pitch),
            _stream(packed)), "plane_sweep_variance_tabled_pitched")
    return buf[..., :W]


@torch.library.custom_op(f"{_NS}::plane_sweep_variance", mutates_args=(), device_types="cuda")
def plane_sweep_variance(feat: Tensor, nbr: Tensor, proj: Tensor, depth: Tensor) -> Tensor:
    """a3+a4 (mvsdet.py:439-467): feat (N,C,H,W), nbr (N,K) int64, proj (N,K,4,4), depth (N,D) -> (N,C,D,H,W)."""
    N, K, C, D, H, W = _check_sweep(feat, nbr, proj, depth)
    packed = pack_features(feat)
    return plane_sweep_variance_packed(packed, nbr, proj, depth, C, H, W)


@plane_sweep_variance.register_fake
def _(feat, nbr, proj, depth):
    N, C, H, W = feat.shape
    return feat.new_empty((N, C, depth.shape[1], H, W))


@torch.library.custom_op(f"{_NS}::plane_sweep_variance_backward", mutates_args=(), device_types="cuda")
def plane_sweep_variance_backward(feat: Tensor, nbr: Tensor, proj: Tensor, depth: Tensor, grad: Tensor) -> Tensor:
    N, K, C, D, H, W = _check_sweep(feat, nbr, proj, depth)
    _req(grad, "grad", dim=5)
    if grad.shape != (N, C, D, H, W):
        raise ValueError("plane_sweep_variance_backward: grad shape mismatch")
    feat, nbr, proj, depth, grad = feat.contiguous(), nbr.contiguous(), proj.contiguous(), depth.contiguous(), grad.contiguous()
    lib = _lib.load()
    wbytes = lib.mvsdet_plane_sweep_bwd_workspace_bytes(N, K, C, D, H, W)
    ws = torch.empty(wbytes // 4, dtype=torch.float32, device=feat.device)
    gfeat = torch.empty_like(feat)
    with torch.cuda.device(feat.device):
        _lib.check(lib.mvsdet_plane_sweep_variance_bwd_f32(_lib.ptr(feat), _lib.ptr(nbr), _lib.ptr(proj), _lib.ptr(depth),
                                                           _lib.ptr(grad), _lib.ptr(gfeat), _lib.ptr(ws), wbytes, N, K, C,
                                                           D, H, W, _stream(feat)), "plane_sweep_variance_backward")
    return gfeat


@plane_sweep_variance_backward.register_fake
def _(feat, nbr, proj, depth, grad):
    return torch.empty_like(feat)


def _sweep_setup(ctx, inputs, output):
    feat, nbr, proj, depth = inputs
    ctx.save_for_backward(feat, nbr, proj, depth)


def _sweep_bwd(ctx, grad):
    feat, nbr, proj, depth = ctx.saved_tensors
    return plane_sweep_variance_backward(feat, nbr, proj, depth, grad), None, None, None


plane_sweep_variance.register_autograd(_sweep_bwd, setup_context=_sweep_setup)


@torch.library.custom_op(f"{_NS}::plane_sweep_variance_keep", mutates_args=(), device_types="cuda")
def plane_sweep_variance_keep(feat: Tensor, nbr: Tensor, proj: Tensor, depth: Tensor) -> Tuple[Tensor, Tensor, Tensor]:
    """a3+a4 for a TRAINING step: (variance (N,C,D,H,W), the packed maps, the sweep geometry) -- what the forward pass makes on its
    way is handed out instead of dropped: the lifting reads the packed maps, and the backward pass takes both
    (`plane_sweep_variance_backward_packed`) instead of packing the features and building the geometry a second time."""
    N, K, C, D, H, W = _check_sweep(feat, nbr, proj, depth)
    packed = pack_features(feat)
    nbr, proj, depth = nbr.contiguous(), proj.contiguous(), depth.contiguous()
    lib = _lib.load()
    out = torch.empty((N, C, D, H, W), dtype=torch.float32, device=feat.device)
    sbytes = lib.mvsdet_plane_sweep_scratch_bytes(N, K, D, H, W)
    table = torch.empty(max(sbytes // 4, 4), dtype=torch.float32, device=feat.device)
    with torch.cuda.device(feat.device):
        _lib.check(lib.mvsdet_plane_sweep_variance_packed_f32(_lib.ptr(packed), _lib.ptr(nbr), _lib.ptr(proj), _lib.ptr(depth),
                                                              _lib.ptr(out), _lib.ptr(table), sbytes, N, K, C, D, H, W,
                                                              _stream(feat)), "plane_sweep_variance_keep")
    return out, packed, table


@plane_sweep_variance_keep.register_fake
def _(feat, nbr, proj, depth):
    N, C, H, W = feat.shape
    sbytes = int(_lib.load().mvsdet_plane_sweep_scratch_bytes(N, nbr.shape[1], depth.shape[1], H, W))   # a host-side size function
    return (feat.new_empty((N, C, depth.shape[1], H, W)), feat.new_empty((N * ((C + 31) // 32) * H * W * 32,)),
            feat.new_empty((max(sbytes // 4, 4),)))


@torch.library.custom_op(f"{_NS}::plane_sweep_variance_backward_packed", mutates_args=(), device_types="cuda")
def plane_sweep_variance_backward_packed(packed: Tensor, nbr: Tensor, table: Tensor, grad: Tensor) -> Tensor:
    """dL/dfeat (N,C,H,W) from dL/dvar (N,C,D,H,W), the packed maps and the sweep geometry of the forward pass."""
    _req(grad, "grad", dim=5)
    _req(nbr, "nbr", dtype=torch.int64, dim=2)
    N, C, D, H, W = grad.shape
    K = nbr.shape[1]
    lib = _lib.load()
    if nbr.shape[0] != N or packed.numel() * 4 != lib.mvsdet_packed_bytes(N, C, H, W):
        raise ValueError("plane_sweep_variance_backward_packed: packed maps / neighbour ids do not match the gradient")
    grad, nbr = grad.contiguous(), nbr.contiguous()
    pb = (int(lib.mvsdet_packed_bytes(N, C, H, W)) + 255) // 256 * 256
    ws = torch.empty(pb // 4, dtype=torch.float32, device=grad.device)
    gfeat = torch.empty((N, C, H, W), dtype=torch.float32, device=grad.device)
    with torch.cuda.device(grad.device):
        _lib.check(lib.mvsdet_plane_sweep_variance_bwd_packed_f32(_lib.ptr(packed), _lib.ptr(nbr), _lib.ptr(table), table.numel() * 4,
                                                                  _lib.ptr(grad), _lib.ptr(gfeat), _lib.ptr(ws), pb, N, K, C, D, H, W,
                                                                  _stream(grad)), "plane_sweep_variance_backward_packed")
    return gfeat


@plane_sweep_variance_backward_packed.register_fake
def _(packed, nbr, table, grad):
    N, C, D, H, W = grad.shape
    return grad.new_empty((N, C, H, W))


def _sweep_keep_setup(ctx, inputs, output):
    ctx.save_for_backward(output[1], inputs[1], output[2])
    ctx.mark_non_differentiable(output[1], output[2])


def _sweep_keep_bwd(ctx, grad, g_packed, g_table):
    packed, nbr, table = ctx.saved_tensors
    return plane_sweep_variance_backward_packed(packed, nbr, table, grad), None, None, None


plane_sweep_variance_keep.register_autograd(_sweep_keep_bwd, setup_context=_sweep_keep_setup)


# ------------------------------------------------------------------------------------------- a5-a7
@torch.library.custom_op(f"{_NS}::depth_prob_topk", mutates_args=(), device_types="cuda")
def depth_prob_topk(cost_reg: Tensor, off_logit: Tensor, near: float, interval: float,
                    topk: int) -> Tuple[Tensor, Tensor, Tensor, Tensor, Tensor, Tensor]:
    """a5-a7 (mvsdet.py:470-475, 266-283, 298-317) -> prob, off (N,D,H,W); est_depth, est_dens (N,topk,H,W);
    est_idx (N,topk,H,W) int32; avg_depth (N,H,W)."""
    _req(cost_reg, "cost_reg", dim=4)
    _req(off_logit, "off_logit", dim=4)
    if cost_reg.shape != off_logit.shape:
        raise ValueError("depth_prob_topk: cost_reg / off_logit shape mismatch")
    N, D, H, W = cost_reg.shape
    # the slices of one (N, 2, D, H, W) tensor (the network's output) are read in place: dense (D, H, W) blocks, one stride
    # between the views; anything else is made contiguous first
    dense = (W * H * D, W * H, W, 1)
    if not (cost_reg.stride()[1:] == dense[1:] == off_logit.stride()[1:] and cost_reg.stride(0) == off_logit.stride(0) >= dense[0]):
        cost_reg, off_logit = cost_reg.contiguous(), off_logit.contiguous()
    view_stride = cost_reg.stride(0) if N > 1 else dense[0]
    dev = cost_reg.device
    prob = torch.empty((N, D, H, W), dtype=torch.float32, device=dev)
    off = torch.empty((N, D, H, W), dtype=torch.float32, device=dev)
    est_depth = torch.empty((N, topk, H, W), dtype=torch.float32, device=dev)
    est_dens = torch.empty((N, topk, H, W), dtype=torch.float32, device=dev)
    est_idx = torch.empty((N, topk, H, W), dtype=torch.int32, device=dev)
    avg = torch.empty((N, H, W), dtype=torch.float32, device=dev)
    with torch.cuda.device(dev):
        _lib.check(_lib.load().mvsdet_depth_prob_topk_strided_f32(_lib.ptr(cost_reg), _lib.ptr(off_logit), view_stride,
                                                                  _lib.ptr(prob), _lib.ptr(off), _lib.ptr(est_depth),
                                                                  _lib.ptr(est_dens), _lib.ptr(est_idx), _lib.ptr(avg), N, D, H,
                                                                  W, topk, near, interval, _stream(cost_reg)), "depth_prob_topk")
    return prob, off, est_depth, est_dens, est_idx, avg


@depth_prob_topk.register_fake
def _(cost_reg, off_logit, near, interval, topk):
    N, D, H, W = cost_reg.shape
    e = cost_reg.new_empty
    return (e((N, D, H, W)), e((N, D, H, W)), e((N, topk, H, W)), e((N, topk, H, W)),
            cost_reg.new_empty((N, topk, H, W), dtype=torch.int32), e((N, H, W)))


@torch.library.custom_op(f"{_NS}::sample_depth_prob", mutates_args=(), device_types="cuda")
def sample_depth_prob(prob: Tensor, off: Tensor, near: float, interval: float,
                      topk: int) -> Tuple[Tensor, Tensor, Tensor, Tensor]:
    """a6+a7 on ready-made probabilities (mvsdet.py:266-283, 298-317) -> est_depth, est_dens (N,topk,H,W),
    est_idx int32, avg_depth (N,H,W)."""
    _req(prob, "prob", dim=4)
    _req(off, "off", dim=4)
    if prob.shape != off.shape:
        raise ValueError("sample_depth_prob: prob / off shape mismatch")
    N, D, H, W = prob.shape
    prob, off = prob.contiguous(), off.contiguous()
    dev = prob.device
    est_depth = torch.empty((N, topk, H, W), dtype=torch.float32, device=dev)
    est_dens = torch.empty((N, topk, H, W), dtype=torch.float32, device=dev)
    est_idx = torch.empty((N, topk, H, W), dtype=torch.int32, device=dev)
    avg = torch.empty((N, H, W), dtype=torch.float32, device=dev)
    with torch.cuda.device(dev):
        _lib.check(_lib.load().mvsdet_sample_depth_prob_f32(_lib.ptr(prob), _lib.ptr(off), _lib.ptr(est_depth),
                                                            _lib.ptr(est_dens), _lib.ptr(est_idx), _lib.ptr(avg), N, D, H,
                                                            W, topk, near, interval, _stream(prob)), "sample_depth_prob")
    return est_depth, est_dens, est_idx, avg


@sample_depth_prob.register_fake
def _(prob, off, near, interval, topk):
    N, D, H, W = prob.shape
    e = prob.new_empty
    return e((N, topk, H, W)), e((N, topk, H, W)), prob.new_empty((N, topk, H, W), dtype=torch.int32), e((N, H, W))


def _sdp_setup(ctx, inputs, output):
    prob, off, near, interval, topk = inputs
    ctx.save_for_backward(prob, off, output[2])
    ctx.near, ctx.interval = near, interval


def _sdp_bwd(ctx, g_depth, g_dens, g_idx, g_avg):
    # gradients w.r.t. prob and off directly (no softmax / sigmoid Jacobian), in plain tensor ops:
    # est_dens = prob[idx]; est_depth = idx*iv + near + off[idx]*iv; avg = sum_d prob_d * depth_d
    prob, off, est_idx = ctx.saved_tensors
    iv, near = ctx.interval, ctx.near
    idx = est_idx.long()
    gp = torch.zeros_like(prob)
    go = torch.zeros_like(prob)
    if g_dens is not None:
        gp.scatter_add_(1, idx, g_dens)
    if g_depth is not None:
        go.scatter_add_(1, idx, g_depth * iv)
    if g_avg is not None:
        d = torch.arange(prob.shape[1], device=prob.device, dtype=torch.float32).view(1, -1, 1, 1)
        gp += g_avg.unsqueeze(1) * ((d * iv + near) + off * iv)
        go += g_avg.unsqueeze(1) * prob * iv
    return gp, go, None, None, None


sample_depth_prob.register_autograd(_sdp_bwd, setup_context=_sdp_setup)


@torch.library.custom_op(f"{_NS}::depth_prob_topk_backward", mutates_args=(), device_types="cuda")
def depth_prob_topk_backward(prob: Tensor, off: Tensor, est_idx: Tensor, g_prob: Tensor, g_depth: Tensor,
                             g_dens: Tensor, g_avg: Tensor, near: float, interval: float) -> Tuple[Tensor, Tensor]:
    N, D, H, W = prob.shape
    topk = est_idx.shape[1]
    g_cost = torch.empty_like(prob)
    g_off = torch.empty_like(prob)
    g_prob, g_depth, g_dens, g_avg = g_prob.contiguous(), g_depth.contiguous(), g_dens.contiguous(), g_avg.contiguous()
    with torch.cuda.device(prob.device):
        _lib.check(_lib.load().mvsdet_depth_prob_topk_bwd_f32(
            _lib.ptr(prob), _lib.ptr(off), _lib.ptr(est_idx), _lib.ptr(g_prob), _lib.ptr(g_depth), _lib.ptr(g_dens),
            _lib.ptr(g_avg), _lib.ptr(g_cost), _lib.ptr(g_off), N, D, H, W, topk, near, interval, _stream(prob)),
            "depth_prob_topk_backward")
    return g_cost, g_off


@depth_prob_topk_backward.register_fake
def _(prob, off, est_idx, g_prob, g_depth, g_dens, g_avg, near, interval):
    return torch.empty_like(prob), torch.empty_like(prob)


def _dp_setup(ctx, inputs, output):
    cost_reg, off_logit, near, interval, topk = inputs
    prob, off, est_depth, est_dens, est_idx, avg = output
    ctx.save_for_backward(prob, off, est_idx)
    ctx.near, ctx.interval = near, interval


def _dp_bwd(ctx, g_prob, g_off, g_depth, g_dens, g_idx, g_avg):
    prob, off, est_idx = ctx.saved_tensors

    def z(g, like_shape):
        return torch.zeros(like_shape, dtype=torch.float32, device=prob.device) if g is None else g

    N, D, H, W = prob.shape
    k = est_idx.shape[1]
    g_cost, g_offl = depth_prob_topk_backward(prob, off, est_idx, z(g_prob, prob.shape), z(g_depth, (N, k, H, W)),
                                              z(g_dens, (N, k, H, W)), z(g_avg, (N, H, W)), ctx.near, ctx.interval)
    if g_off is not None:  # direct gradient on the sigmoid output
        g_offl = g_offl + g_off * off * (1.0 - off)
    return g_cost, g_offl, None, None, None


depth_prob_topk.register_autograd(_dp_bwd, setup_context=_dp_setup)


# ------------------------------------------------------------------------------------------- a9
def _check_stage3(features, points, projection, est_depth, est_dens):
    _req(features, "features", dim=4)
    _req(points, "points")
    _req(projection, "projection", dim=3)
    _req(est_depth, "est_depth", dim=4)
    _req(est_dens, "est_dens", dim=4)
    N, C, h, w = features.shape
    if points.shape[0] != 3:
        raise ValueError("backproject_weigh: points must be (3, ...)")
    V = points.numel() // 3
    J = est_depth.shape[1]
    if projection.shape != (N, 3, 4):
        raise ValueError(f"backproject_weigh: projection {tuple(projection.shape)} != ({N},3,4)")
    if est_depth.shape != (N, J, h, w) or est_dens.shape != (N, J, h, w):
        raise ValueError(f"backproject_weigh: est_depth/est_dens must be ({N},J,{h},{w}), got "
                         f"{tuple(est_depth.shape)} / {tuple(est_dens.shape)}")
    if est_depth.stride() != est_dens.stride():
        est_dens = est_dens.contiguous()
        est_depth = est_depth.contiguous()
    return N, C, h, w, V, J, est_depth, est_dens


@torch.library.custom_op(f"{_NS}::backproject_weigh", mutates_args=(), device_types="cuda")
def backproject_weigh(features: Tensor, points: Tensor, projection: Tensor, est_depth: Tensor, est_dens: Tensor,
                      vz: float) -> Tuple[Tensor, Tensor]:
    """a9 (mvsdet.py:1372): features (N,C,h,w) any strides; points (3,X,Y,Z); projection (N,3,4);
    est_depth/est_dens (N,J,h,w) any strides -> volume (N,C,V) fp32, valid (N,V) bool."""
    N, C, h, w, V, J, est_depth, est_dens = _check_stage3(features, points, projection, est_depth, est_dens)
    points, projection = points.contiguous(), projection.contiguous()
    dev = features.device
    volume = torch.empty((N, C, V), dtype=torch.float32, device=dev)
    valid = torch.empty((N, V), dtype=torch.uint8, device=dev)
    with torch.cuda.device(dev):
        _lib.check(_lib.load().mvsdet_backproject_weigh_f32(
            _lib.ptr(features), _lib.strides4(features), _lib.ptr(points), _lib.ptr(projection), _lib.ptr(est_depth),
            _lib.ptr(est_dens), _lib.strides4(est_depth), _lib.ptr(volume), _lib.ptr(valid), None, None,
            N, C, h, w, V, J, vz, _stream(features)), "backproject_weigh")
    return volume, valid.bool()


@backproject_weigh.register_fake
def _(features, points, projection, est_depth, est_dens, vz):
    N, C = features.shape[:2]
    V = points.numel() // 3
    return features.new_empty((N, C, V)), features.new_empty((N, V), dtype=torch.bool)


@torch.library.custom_op(f"{_NS}::backproject_weigh_backward", mutates_args=(), device_types="cuda")
def backproject_weigh_backward(features: Tensor, points: Tensor, projection: Tensor, est_depth: Tensor,
                               est_dens: Tensor, vz: float, grad: Tensor) -> Tuple[Tensor, Tensor]:
    N, C, h, w, V, J, est_depth, est_dens = _check_stage3(features, points, projection, est_depth, est_dens)
    points, projection, grad = points.contiguous(), projection.contiguous(), grad.contiguous()
    dev = features.device
    gfeat = torch.empty((N, C, h, w), dtype=torch.float32, device=dev)
    gdens = torch.empty((N, J, h, w), dtype=torch.float32, device=dev)
    with torch.cuda.device(dev):
        _lib.check(_lib.load().mvsdet_backproject_weigh_bwd_f32(
            _lib.ptr(features), _lib.strides4(features), _lib.ptr(points), _lib.ptr(projection), _lib.ptr(est_depth),
            _lib.ptr(est_dens), _lib.strides4(est_depth), _lib.ptr(grad), _lib.ptr(gfeat), _lib.ptr(gdens),
            N, C, h, w, V, J, vz, _stream(features)), "backproject_weigh_backward")
    return gfeat, gdens


@backproject_weigh_backward.register_fake
def _(features, points, projection, est_depth, est_dens, vz, grad):
    N, C, h, w = features.shape
    return features.new_empty((N, C, h, w)), features.new_empty((N, est_depth.shape[1], h, w))


def _bp_setup(ctx, inputs, output):
    features, points, projection, est_depth, est_dens, vz = inputs
    ctx.save_for_backward(features, points, projection, est_depth, est_dens)
    ctx.vz = vz


def _bp_bwd(ctx, g_volume, g_valid):
    features, points, projection, est_depth, est_dens = ctx.saved_tensors
    gfeat, gdens = backproject_weigh_backward(features, points, projection, est_depth, est_dens, ctx.vz, g_volume)
    return gfeat, None, None, None, gdens, None


backproject_weigh.register_autograd(_bp_bwd, setup_context=_bp_setup)


# ------------------------------------------------------------------------------------------- a9+a10
@torch.library.custom_op(f"{_NS}::backproject_weigh_mean", mutates_args=(), device_types="cuda")
def backproject_weigh_mean(features: Tensor, packed: Tensor, points: Tensor, projection: Tensor, est_depth: Tensor,
                           est_dens: Tensor, H: int, W: int, vz: float) -> Tuple[Tensor, Tensor]:
    """a9+a10 fused (mvsdet.py:1372 + 511-515).  `features` is the (N,C,h,w) crop view (used for shapes and by
    the backward pass), `packed` = pack_features(full (N,C,H,W) maps).  -> mean (C,V) fp32, count (V) int32."""
    N, C, h, w, V, J, est_depth, est_dens = _check_stage3(features, points, projection, est_depth, est_dens)
    _req(packed, "packed", dim=1)
    lib = _lib.load()
    if packed.numel() * 4 != lib.mvsdet_packed_bytes(N, C, H, W):
        raise ValueError("backproject_weigh_mean: packed buffer does not match (N,C,H,W)")
    points, projection = points.contiguous(), projection.contiguous()
    dev = features.device
    mean = torch.empty((C, V), dtype=torch.float32, device=dev)
    count = torch.empty((V,), dtype=torch.int32, device=dev)
    with torch.cuda.device(dev):
        _lib.check(lib.mvsdet_backproject_weigh_mean_packed_f32(
            _lib.ptr(packed), _lib.ptr(points), _lib.ptr(projection), _lib.ptr(est_depth), _lib.ptr(est_dens),
            _lib.strides4(est_depth), _lib.ptr(mean), _lib.ptr(count), N, C, H, W, h, w, V, J, vz, _stream(features)),
            "backproject_weigh_mean")
    return mean, count


@backproject_weigh_mean.register_fake
def _(features, packed, points, projection, est_depth, est_dens, H, W, vz):
    C = features.shape[1]
    V = points.numel() // 3
    return features.new_empty((C, V)), features.new_empty((V,), dtype=torch.int32)


@torch.library.custom_op(f"{_NS}::backproject_weigh_sum_shard", mutates_args=(), device_types="cuda")
def backproject_weigh_sum_shard(packed: Tensor, points: Tensor, projection: Tensor, est_depth: Tensor, est_dens: Tensor,
                                n_src: int, ref_first: int, C: int, H: int, W: int, vz: float) -> Tuple[Tensor, Tensor]:
    """a9 summed over the M views ref_first .. ref_first+M-1 only, NOT divided (the a10 division happens after the
    ranks' sums and counts are all-reduced).  projection (M,3,4), est_depth / est_dens (M,J,h,w) are the shard's;
    `packed` holds all n_src views.  -> sum (C,V) fp32, count (V) int32.  Forward only."""
    _req(packed, "packed", dim=1)
    _req(points, "points")
    _req(projection, "projection", dim=3)
    _req(est_depth, "est_depth", dim=4)
    _req(est_dens, "est_dens", dim=4)
    M, J, h, w = est_depth.shape
    V = points.numel() // 3
    lib = _lib.load()
    if packed.numel() * 4 != lib.mvsdet_packed_bytes(n_src, C, H, W):
        raise ValueError("backproject_weigh_sum_shard: packed buffer does not match (n_src,C,H,W)")
    if points.shape[0] != 3 or projection.shape != (M, 3, 4) or est_dens.shape != est_depth.shape:
        raise ValueError("backproject_weigh_sum_shard: points / projection / est_dens shape mismatch")
    if M < 1 or ref_first < 0 or ref_first + M > n_src:
        raise ValueError(f"backproject_weigh_sum_shard: views [{ref_first},{ref_first + M}) outside the {n_src} packed views")
    if est_depth.stride() != est_dens.stride():
        est_depth, est_dens = est_depth.contiguous(), est_dens.contiguous()
    points, projection = points.contiguous(), projection.contiguous()
    dev = packed.device
    total = torch.empty((C, V), dtype=torch.float32, device=dev)
    count = torch.empty((V,), dtype=torch.int32, device=dev)
    view_floats = lib.mvsdet_packed_bytes(1, C, H, W) // 4
    with torch.cuda.device(dev):
        _lib.check(lib.mvsdet_backproject_weigh_sum_packed_f32(
            _lib.ptr(packed[view_floats * ref_first:]), _lib.ptr(points), _lib.ptr(projection), _lib.ptr(est_depth),
            _lib.ptr(est_dens), _lib.strides4(est_depth), _lib.ptr(total), _lib.ptr(count), M, C, H, W, h, w, V, J, vz,
            _stream(packed)), "backproject_weigh_sum_shard")
    return total, count


@backproject_weigh_sum_shard.register_fake
def _(packed, points, projection, est_depth, est_dens, n_src, ref_first, C, H, W, vz):
    V = points.numel() // 3
    return packed.new_empty((C, V)), packed.new_empty((V,), dtype=torch.int32)


@torch.library.custom_op(f"{_NS}::backproject_weigh_mean_backward", mutates_args=(), device_types="cuda")
def backproject_weigh_mean_backward(features: Tensor, points: Tensor, projection: Tensor, est_depth: Tensor,
                                    est_dens: Tensor, count: Tensor, vz: float, grad: Tensor) -> Tuple[Tensor, Tensor]:
    N, C, h, w, V, J, est_depth, est_dens = _check_stage3(features, points, projection, est_depth, est_dens)
    points, projection, grad = points.contiguous(), projection.contiguous(), grad.contiguous()
    dev = features.device
    gfeat = torch.empty((N, C, h, w), dtype=torch.float32, device=dev)
    gdens = torch.empty((N, J, h, w), dtype=torch.float32, device=dev)
    with torch.cuda.device(dev):
        _lib.check(_lib.load().mvsdet_backproject_weigh_mean_bwd_f32(
            _lib.ptr(features), _lib.strides4(features), _lib.ptr(points), _lib.ptr(projection), _lib.ptr(est_depth),
            _lib.ptr(est_dens), _lib.strides4(est_depth), _lib.ptr(count), _lib.ptr(grad), _lib.ptr(gfeat),
            _lib.ptr(gdens), N, C, h, w, V, J, vz, _stream(features)), "backproject_weigh_mean_backward")
    return gfeat, gdens


@backproject_weigh_mean_backward.register_fake
def _(features, points, projection, est_depth, est_dens, count, vz, grad):
    N, C, h, w = features.shape
    return features.new_empty((N, C, h, w)), features.new_empty((N, est_depth.shape[1], h, w))


def _bpm_setup(ctx, inputs, output):
    features, packed, points, projection, est_depth, est_dens, H, W, vz = inputs
    mean, count = output
    ctx.save_for_backward(features, points, projection, est_depth, est_dens, count)
    ctx.vz = vz


def _bpm_bwd(ctx, g_mean, g_count):
    features, points, projection, est_depth, est_dens, count = ctx.saved_tensors
    gfeat, gdens = backproject_weigh_mean_backward(features, points, projection, est_depth, est_dens, count, ctx.vz,
                                                   g_mean)
    return gfeat, None, None, None, None, gdens, None, None, None


backproject_weigh_mean.register_autograd(_bpm_bwd, setup_context=_bpm_setup)


# ------------------------------------------------------------------------------------------- cost network on bf16x3 (f-1)
def split_bf16(t: Tensor):
    """fp32 -> (hi, mid) bfloat16 pieces: hi = bf16(t), mid = bf16(t - hi), both round-to-nearest-even; t - hi is exact."""
    hi = t.to(torch.bfloat16)
    mid = (t - hi.to(torch.float32)).to(torch.bfloat16)
    return hi, mid


def _tap_table(order: int):
    """The 3x3x3 tap of each half of the 14 tap pairs (-1 = empty), as csrc/costreg_bf16.hip:tap_table builds it."""
    if order == 0:
        return [i if i < 27 else -1 for i in range(28)]
    t = []
    for pi in range(8):
        pd, ph, pw = pi >> 2, (pi >> 1) & 1, pi & 1
        nt = 1 << (pd + ph + pw)
        for j in range(nt):
            bits, jw, jh, jd = j, 0, 0, 0
            if pw:
                jw, bits = bits & 1, bits >> 1
            if ph:
                jh, bits = bits & 1, bits >> 1
            if pd:
                jd = bits & 1
            t.append(((2 * jd if pd else 1) * 3 + (2 * jh if ph else 1)) * 3 + (2 * jw if pw else 1))
        if nt == 1:
            t.append(-1)
    return t


# MFMA row m of a row group carries output channel ROW_CHANNEL[m] (csrc/costreg_bf16.hip: bf_mfma_row_channel): a lane's
# accumulator registers 8q..8q+7 are then eight consecutive channels, i.e. one 16-byte unit of the SCL form
ROW_CHANNEL = [8 * ((m >> 4) * 2 + ((m >> 2) & 1)) + (m & 3) + 4 * ((m >> 3) & 1) for m in range(32)]


def get_option(name: str) -> int:
    """A tuning option of the library (mvsdet_get_option)."""
    import ctypes
    v = ctypes.c_int(0)
    _lib.check(_lib.load().mvsdet_get_option(name.encode(), ctypes.byref(v)), "get_option")
    return int(v.value)


def split_conv_weight(weight: Tensor, order: int = 0) -> Tensor:
    """Conv3d weight (Cout = 64*m, Cin, 3,3,3) fp32 -> the layout the bf16x3 kernels stream into LDS (include/mvsdet_hip.h):
    [Cout/64][ceil(Cin/8)][14 tap pairs][2 row groups][2 pieces][64 lanes][8 channels] bf16, lane = 32*(half of the pair) +
    MFMA row m (which carries output ROW_CHANNEL[m] of its group of 32); empty halves and the channels beyond Cin are zero.
    order 0: stride-1 convolution (pair p = taps 2p, 2p+1);
    1: stride-2 convolution (pairs grouped by the parity class of the input voxel); 2: `weight` is a ConvTranspose3d weight
    (Cin, Cout = 64*m, 3,3,3), pairs grouped by the parity class of the output voxel.  On a ROCm device this is ONE small
    kernel (run per call: in-place weight updates are always seen); on the CPU the same layout from torch ops (tests)."""
    if order == 2:
        cin, cout = weight.shape[:2]
    else:
        cout, cin = weight.shape[:2]
    if cout % 64 or tuple(weight.shape[2:]) != (3, 3, 3) or order not in (0, 1, 2, 3):
        raise ValueError(f"split_conv_weight: weight {tuple(weight.shape)} (order {order}) has no 64*m output channels / 3x3x3 taps")
    c8 = (cin + 7) // 8
    if weight.is_cuda:
        w = weight.detach().to(torch.float32).contiguous()
        out = torch.empty((cout // 64, c8, 14, 2, 2, 64, 8), dtype=torch.bfloat16, device=w.device)
        with torch.cuda.device(w.device):
            _lib.check(_lib.load().mvsdet_split_conv_weight_ordered(_lib.ptr(w), _lib.ptr(out), cout, cin, order, _stream(w)),
                       "split_conv_weight")
        return out
    w = weight.detach().to(torch.float32)
    w = (w.transpose(0, 1) if order == 2 else w).reshape(cout, cin, 27)
    taps = torch.tensor(_tap_table(0 if order == 3 else order))
    w = torch.where(taps.view(1, 1, 28) >= 0, w[:, :, taps.clamp(min=0)], torch.zeros(()))   # (Cout, Cin, 28) in pair order
    w = torch.nn.functional.pad(w, (0, 0, 0, c8 * 8 - cin))                                  # channels -> 8*c8
    pieces = torch.stack(split_bf16(w), 0)                                                  # (piece, Cout, C, 28)
    if order == 3 or (order == 0 and get_option("conv_mfma16")):
        # the 16x16x32 form: channel = ob*64 + 32q + 8mh + 4b + ml (row group rg = 2q + b, row m = 4mh + ml), tap = 4ks + kg
        # (piece, ob, q, mh, b, ml, c8, j, ks, kg) -> (ob, c8, ks, q, b, piece, kg, mh, ml, j)
        pieces = pieces.reshape(2, cout // 64, 2, 4, 2, 4, c8, 8, 7, 4).permute(1, 6, 8, 2, 4, 0, 9, 3, 5, 7)
        return pieces.contiguous().reshape(cout // 64, c8, 14, 2, 2, 64, 8)
    # (piece, ob, a, m, c8, j, p, h) with row m <- channel ROW_CHANNEL[m]  -> (ob, c8, p, a, piece, h, m, j)
    pieces = pieces.reshape(2, cout // 64, 2, 32, c8, 8, 14, 2)[:, :, :, torch.tensor(ROW_CHANNEL)].permute(1, 4, 6, 2, 0, 7, 3, 5)
    return pieces.contiguous().reshape(cout // 64, c8, 14, 2, 2, 64, 8)


def split_conv_weights(items) -> list:
    """`split_conv_weight` of several (weight, order) pairs in ONE launch (up to 8 per launch): a network's layers, cut anew on
    every forward call so that an in-place weight update -- also one through `.data`, which bumps no version counter -- is seen."""
    import ctypes
    items = list(items)
    outs = []
    lib = _lib.load()
    for lo in range(0, len(items), 8):
        chunk = items[lo:lo + 8]
        ws, cins, couts, orders, res = [], [], [], [], []
        for weight, order in chunk:
            cin, cout = (weight.shape[:2] if order == 2 else weight.shape[1::-1])
            if cout % 64 or tuple(weight.shape[2:]) != (3, 3, 3) or order not in (0, 1, 2) or not weight.is_cuda:
                raise ValueError(f"split_conv_weights: weight {tuple(weight.shape)} (order {order})")
            ws.append(weight.detach().to(torch.float32).contiguous())
            res.append(torch.empty((cout // 64, (cin + 7) // 8, 14, 2, 2, 64, 8), dtype=torch.bfloat16, device=weight.device))
            cins.append(int(cin)); couts.append(int(cout)); orders.append(int(order))
        n = len(chunk)
        wp = (ctypes.c_void_p * n)(*[w.data_ptr() for w in ws])
        op = (ctypes.c_void_p * n)(*[r.data_ptr() for r in res])
        ia = lambda v: (ctypes.c_int * n)(*v)   # noqa: E731
        with torch.cuda.device(ws[0].device):
            _lib.check(lib.mvsdet_split_conv_weights_batched(wp, op, ia(couts), ia(cins), ia(orders), n, _stream(ws[0])),
                       "split_conv_weights_batched")
        outs.extend(res)
    return outs


class SclTensor:
    """An activation in the split channel-last form the bf16x3 convolutions read (include/mvsdet_hip.h): `data` is the
    flat bfloat16 buffer [2][N][ceil(C/8)][Dp][Hp][Wp][8] with a zero border; `shape` the logical (N,C,D,H,W)."""

    def __init__(self, data: Tensor, shape, padded):
        self.data, self.shape, self.padded = data, tuple(int(v) for v in shape), tuple(int(v) for v in padded)

    def pieces(self):
        """(hi, mid) as (N, C8*8, D, H, W) bfloat16 tensors (interior only) -- for tests."""
        n, c, d, h, w = self.shape
        c8 = (c + 7) // 8
        dp, hp, wp = self.padded
        v = self.data.view(2, n, c8, dp, hp, wp, 8)[:, :, :, 1:d + 1, 1:h + 1, 1:w + 1]
        v = v.permute(0, 1, 2, 6, 3, 4, 5).reshape(2, n, c8 * 8, d, h, w)
        return v[0], v[1]

    def border_is_zero(self) -> bool:
        """Everything outside the interior voxels (tests: a producing kernel must not write there)."""
        n, c, d, h, w = self.shape
        dp, hp, wp = self.padded
        v = self.data.view(2, n, (c + 7) // 8, dp, hp, wp, 8).clone()
        v[:, :, :, 1:d + 1, 1:h + 1, 1:w + 1] = 0
        return not bool(v.view(torch.int16).any())


class PsclTensor:
    """The parity-split SCL form (include/mvsdet_hip.h): flat bfloat16 buffer [2][8 classes][N][ceil(C/8)][cDp][cHp][cWp][8];
    class = 4*(d&1) + 2*(h&1) + (w&1), voxel (d,h,w) at index (d//2 + 1, h//2 + 1, w//2 + 1) of its class."""

    def __init__(self, data: Tensor, shape, padded):
        self.data, self.shape, self.padded = data, tuple(int(v) for v in shape), tuple(int(v) for v in padded)

    def pieces(self):
        """(hi, mid) as (N, C8*8, D, H, W) bfloat16 tensors, and whether everything else in the buffer is zero -- for tests."""
        n, c, d, h, w = self.shape
        c8 = (c + 7) // 8
        dp, hp, wp = self.padded
        v = self.data.view(2, 8, n, c8, dp, hp, wp, 8)
        out = torch.zeros((2, n, c8, d, h, w, 8), dtype=torch.bfloat16, device=self.data.device)
        rest = v.clone()
        for cls in range(8):
            pd, ph, pw = cls >> 2, (cls >> 1) & 1, cls & 1
            nd, nh, nw = (d - pd + 1) // 2, (h - ph + 1) // 2, (w - pw + 1) // 2
            out[:, :, :, pd::2, ph::2, pw::2] = v[:, cls, :, :, 1:nd + 1, 1:nh + 1, 1:nw + 1]
            rest[:, cls, :, :, 1:nd + 1, 1:nh + 1, 1:nw + 1] = 0
        out = out.permute(0, 1, 2, 6, 3, 4, 5).reshape(2, n, c8 * 8, d, h, w)
        return out[0], out[1], not bool(rest.view(torch.int16).any())


def scl_geometry(N: int, C: int, D: int, H: int, W: int):
    """(bytes, (Dp, Hp, Wp)) of the SCL form of an (N,C,D,H,W) activation."""
    import ctypes
    dp, hp, wp = ctypes.c_int(0), ctypes.c_int(0), ctypes.c_int(0)
    nbytes = _lib.load().mvsdet_scl_bytes(N, C, D, H, W, ctypes.byref(dp), ctypes.byref(hp), ctypes.byref(wp))
    return int(nbytes), (dp.value, hp.value, wp.value)


def pscl_geometry(N: int, C: int, D: int, H: int, W: int):
    """(bytes, (cDp, cHp, cWp)) of the parity-split SCL form of an (N,C,D,H,W) activation."""
    import ctypes
    dp, hp, wp = ctypes.c_int(0), ctypes.c_int(0), ctypes.c_int(0)
    nbytes = _lib.load().mvsdet_pscl_bytes(N, C, D, H, W, ctypes.byref(dp), ctypes.byref(hp), ctypes.byref(wp))
    return int(nbytes), (dp.value, hp.value, wp.value)


def scl_empty(shape, device) -> SclTensor:
    """A zeroed SCL buffer for an (N,C,D,H,W) result: a producing convolution fills the interior, the border stays zero."""
    nbytes, padded = scl_geometry(*[int(v) for v in shape])
    return SclTensor(torch.zeros((nbytes // 2,), dtype=torch.bfloat16, device=device), shape, padded)


def pscl_empty(shape, device) -> PsclTensor:
    nbytes, padded = pscl_geometry(*[int(v) for v in shape])
    return PsclTensor(torch.zeros((nbytes // 2,), dtype=torch.bfloat16, device=device), shape, padded)


def pscl_from_tensor(x: Tensor) -> PsclTensor:
    """(N,C,D,H,W) fp32 -> PsclTensor with torch operators (tests and tools: in the product the form is written by the
    producing convolution's epilogue)."""
    n, c, d, h, w = x.shape
    c8 = (c + 7) // 8
    out = pscl_empty(x.shape, x.device)
    dp, hp, wp = out.padded
    v = out.data.view(2, 8, n, c8, dp, hp, wp, 8)
    xp = torch.nn.functional.pad(x, (0, 0, 0, 0, 0, 0, 0, c8 * 8 - c))
    for piece, t in enumerate(split_bf16(xp)):
        t = t.view(n, c8, 8, d, h, w).permute(0, 1, 3, 4, 5, 2)           # (n, c8, d, h, w, 8)
        for cls in range(8):
            pd, ph, pw = cls >> 2, (cls >> 1) & 1, cls & 1
            sub = t[:, :, pd::2, ph::2, pw::2]
            v[piece, cls, :, :, 1:sub.shape[2] + 1, 1:sub.shape[3] + 1, 1:sub.shape[4] + 1] = sub
    return out


def scl_pack(x: Tensor, out: Optional[SclTensor] = None) -> SclTensor:
    """(N,C,D,H,W) fp32 -> SclTensor.  x may be any view whose last dimension has stride 1 (a row-pitched cost volume is
    read in place).  `out`: a buffer of the same shape to refill (its border is already zero); without one the result is a
    new buffer from the caching allocator whose border the packing kernel writes itself (no clearing pass, nothing kept)."""
    import ctypes
    _req(x, "x", dim=5)
    if x.stride(4) != 1 or min(x.stride()) < 0:
        x = x.contiguous()
    N, C, D, H, W = x.shape
    nbytes, padded = scl_geometry(N, C, D, H, W)
    fresh = out is None or out.shape != tuple(x.shape) or out.data.device != x.device
    if fresh:
        out = SclTensor(torch.empty((nbytes // 2,), dtype=torch.bfloat16, device=x.device), x.shape, padded)
    xstr = (ctypes.c_int64 * 4)(*[int(v) for v in x.stride()[:4]])
    with torch.cuda.device(x.device):
        _lib.check(_lib.load().mvsdet_scl_pack_f32(_lib.ptr(x), xstr, _lib.ptr(out.data), N, C, D, H, W, 2 if fresh else 0, _stream(x)),
                   "scl_pack")
    return out


def gemm_split_weight(wmat: Tensor) -> Tensor:
    """(M, K) fp32 row-major -> the fragment-ordered bf16 pieces the neck's GEMM-shaped layers stream from L2
    ([M/32][K/16][2 pieces][64 lanes][8] bf16; M % 128 == 0, K % 32 == 0: include/mvsdet_hip.h)."""
    _req(wmat, "wmat", dim=2)
    M, K = wmat.shape
    lib = _lib.load()
    nbytes = int(lib.mvsdet_gemm_split_weight_bytes(int(M), int(K)))
    if nbytes == 0:
        raise ValueError(f"gemm_split_weight: ({M}, {K}) is not (128 m, 32 k)")
    w = wmat.detach().contiguous()
    out = torch.empty((nbytes // 2,), dtype=torch.bfloat16, device=w.device)
    with torch.cuda.device(w.device):
        _lib.check(lib.mvsdet_gemm_split_weight(_lib.ptr(w), _lib.ptr(out), int(M), int(K), _stream(w)), "gemm_split_weight")
    return out


def _check_wsplit(name: str, wsplit: Tensor, rows: int, cin: int, like: Tensor):
    """A split weight is read as [rows/32][cin/16][2][64][8] bf16 with no further check in the kernel: one made for another
    (rows, Cin) would be read out of bounds."""
    need = int(_lib.load().mvsdet_gemm_split_weight_bytes(int(rows), int(cin)))
    if need == 0:
        raise ValueError(f"{name}: ({rows}, {cin}) is not a (128 m, 32 k) layer")
    if not isinstance(wsplit, Tensor) or wsplit.dtype != torch.bfloat16 or not wsplit.is_contiguous() or wsplit.device != like.device:
        raise TypeError(f"{name}: wsplit must be the contiguous bfloat16 tensor gemm_split_weight made, on the input's device")
    if wsplit.numel() * 2 != need:
        raise ValueError(f"{name}: wsplit holds {wsplit.numel() * 2} bytes, a ({rows}, {cin}) layer's split weight {need}")


def gemm_layer_ok(rows: int, cin: int) -> bool:
    """Whether the neck's GEMM kernels take a layer with this many matrix rows (Cout, or 8 Cout for the transposed one) and Cin."""
    return rows % 128 == 0 and cin % 32 == 0


def conv3d_k1_s2_bf16x3(x: Tensor, wsplit: Tensor, bias: Tensor, cout: int, relu: bool = False) -> Tensor:
    """Conv3d(kernel 1, stride 2) + bias [+ ReLU] of an (N,Cin,D,H,W) tensor with even D, H, W: one GEMM over the sub-sampled voxels,
    the sub-sampling in its gather (imvoxel_neck.py:196-217 `downsample` with the eval-mode BatchNorm folded in)."""
    _req(x, "x", dim=5)
    _req(bias, "bias", dim=1)
    x = x.contiguous()
    N, Cin, D, H, W = x.shape
    if D % 2 or H % 2 or W % 2 or bias.numel() != cout:
        raise ValueError("conv3d_k1_s2_bf16x3: D, H, W must be even and bias have Cout elements")
    _check_wsplit("conv3d_k1_s2_bf16x3", wsplit, cout, Cin, x)
    out = torch.empty((N, cout, D // 2, H // 2, W // 2), dtype=torch.float32, device=x.device)
    with torch.cuda.device(x.device):
        _lib.check(_lib.load().mvsdet_conv3d_k1_s2_bf16x3(_lib.ptr(x), _lib.ptr(wsplit), _lib.ptr(bias.contiguous()), _lib.ptr(out), N, Cin,
                                                          int(cout), D, H, W, int(relu), _stream(x)), "conv3d_k1_s2_bf16x3")
    return out


def convT3d_k2_s2_bf16x3(x: Tensor, wsplit: Tensor, bias: Tensor, cout: int, relu: bool = True) -> Tensor:
    """ConvTranspose3d(kernel 2, stride 2) + bias [+ ReLU]: (N,Cin,D,H,W) -> (N,Cout,2D,2H,2W), the 2x2x2 interleave written by the
    GEMM's epilogue (imvoxel_neck.py:166-180).  `wsplit` = gemm_split_weight of the (8 Cout, Cin) matrix with rows 8 o + 4 p + 2 q + r."""
    _req(x, "x", dim=5)
    _req(bias, "bias", dim=1)
    x = x.contiguous()
    N, Cin, D, H, W = x.shape
    if bias.numel() != cout:
        raise ValueError("convT3d_k2_s2_bf16x3: bias must have Cout elements")
    _check_wsplit("convT3d_k2_s2_bf16x3", wsplit, 8 * cout, Cin, x)
    out = torch.empty((N, cout, 2 * D, 2 * H, 2 * W), dtype=torch.float32, device=x.device)
    with torch.cuda.device(x.device):
        _lib.check(_lib.load().mvsdet_convT3d_k2_s2_bf16x3(_lib.ptr(x), _lib.ptr(wsplit), _lib.ptr(bias.contiguous()), _lib.ptr(out), N, Cin,
                                                           int(cout), D, H, W, int(relu), _stream(x)), "convT3d_k2_s2_bf16x3")
    return out


def _check_affine(name, scale, shift, Cout):
    if (scale is None) != (shift is None):
        raise ValueError(f"{name}: scale and shift come together")
    if scale is not None:
        _req(scale, "scale", dim=1)
        _req(shift, "shift", dim=1)
        if scale.numel() != Cout or shift.numel() != Cout:
            raise ValueError(f"{name}: scale / shift must have {Cout} elements")
        return scale.contiguous(), shift.contiguous()
    return None, None


def _conv_outputs(name, outputs, shape, dev, scl_out, pscl_out, allow_pscl=True):
    """The output buffers a convolution was asked for: `outputs` is a sequence out of "f32", "scl", "pscl" (order = order of
    the returned values); `scl_out` / `pscl_out` are buffers of the same shape to refill (their borders are already zero)."""
    if isinstance(outputs, str):
        outputs = (outputs,)
    bad = [o for o in outputs if o not in ("f32", "scl") + (("pscl",) if allow_pscl else ())]
    if bad or not outputs:
        raise ValueError(f"{name}: outputs {outputs!r}")
    res = {}
    if "f32" in outputs:
        res["f32"] = torch.empty(shape, dtype=torch.float32, device=dev)
    if "scl" in outputs:
        ok = scl_out is not None and scl_out.shape == tuple(shape) and scl_out.data.device == dev
        res["scl"] = scl_out if ok else scl_empty(shape, dev)
    if "pscl" in outputs:
        ok = pscl_out is not None and pscl_out.shape == tuple(shape) and pscl_out.data.device == dev
        res["pscl"] = pscl_out if ok else pscl_empty(shape, dev)
    return outputs, res


def _ret(outputs, res):
    vals = tuple(res[o] for o in outputs)
    return vals[0] if len(vals) == 1 else vals


def conv3d_k3_bf16x3(x, weight_split: Tensor, scale: Optional[Tensor], shift: Optional[Tensor], relu: bool,
                     residual: Optional[Tensor] = None, outputs=("f32",), scl_out: Optional[SclTensor] = None,
                     pscl_out: Optional[PsclTensor] = None):
    """Conv3d(Cin -> Cout = 64*m, kernel 3, stride 1, padding 1, no bias) [+ affine] [+ residual] [+ ReLU] on the bf16
    matrix cores with three-term split operands (csrc/costreg_bf16.hip).
    x: the fp32 (N,Cin,D,H,W) tensor itself (any view with w stride 1: cut into pieces inside the kernel, no extra pass) or
    its SclTensor form (scl_pack, or a producing layer's "scl" output) -- identical results.
    outputs: any of "f32" (the (N,Cout,D,H,W) tensor), "scl" (SclTensor: already cut for a stride-1 / transposed consumer),
    "pscl" (PsclTensor: for a stride-2 consumer); one value or a tuple in that order is returned."""
    import ctypes
    scl = isinstance(x, SclTensor)
    if scl:
        N, Cin, D, H, W = x.shape
        dev = x.data.device
    else:
        _req(x, "x", dim=5)
        if x.stride(4) != 1 or min(x.stride()) < 0:
            x = x.contiguous()
        N, Cin, D, H, W = x.shape
        dev = x.device
    if weight_split.dtype != torch.bfloat16 or weight_split.dim() != 7 or tuple(weight_split.shape[1:]) != ((Cin + 7) // 8, 14, 2, 2, 64, 8):
        raise ValueError(f"conv3d_k3_bf16x3: weight_split {tuple(weight_split.shape)} does not match Cin={Cin}")
    Cout = weight_split.shape[0] * 64
    scale, shift = _check_affine("conv3d_k3_bf16x3", scale, shift, Cout)
    outputs, res = _conv_outputs("conv3d_k3_bf16x3", outputs, (N, Cout, D, H, W), dev, scl_out, pscl_out)
    if residual is not None:
        _req(residual, "residual", dim=5)
        if tuple(residual.shape) != (N, Cout, D, H, W):
            raise ValueError(f"conv3d_k3_bf16x3: residual {tuple(residual.shape)} != output {(N, Cout, D, H, W)}")
        residual = residual.contiguous()
    weight_split = weight_split.contiguous()
    lib = _lib.load()
    # small volumes: partial sums of the input-channel splits (0 bytes: the grid fills the chip unsplit); fp32 output only
    wbytes = lib.mvsdet_conv3d_k3_bf16x3_workspace_bytes(N, Cin, Cout, D, H, W) if tuple(outputs) == ("f32",) else 0
    ws = torch.empty((wbytes // 4,), dtype=torch.float32, device=dev) if wbytes else None
    xstr = None if scl else (ctypes.c_int64 * 4)(*[int(v) for v in x.stride()[:4]])
    with torch.cuda.device(dev):
        _lib.check(lib.mvsdet_conv3d_k3_bf16x3_io(_lib.ptr(x.data) if scl else None, None if scl else _lib.ptr(x), xstr,
                                                  _lib.ptr(weight_split), _lib.ptr(scale), _lib.ptr(shift), _lib.ptr(residual),
                                                  _lib.ptr(res.get("f32")), _lib.ptr(res["scl"].data) if "scl" in res else None,
                                                  _lib.ptr(res["pscl"].data) if "pscl" in res else None, _lib.ptr(ws), wbytes,
                                                  N, Cin, Cout, D, H, W, int(relu), _lib.current_stream(dev)), "conv3d_k3_bf16x3")
    return _ret(outputs, res)


def split_conv_weight_mx(weight: Tensor) -> Tensor:
    """(Cout = 64 m, Cin, 3,3,3) fp32 -> the operand image of `conv3d_k3_fp16mx`: per (block of 64 outputs, 8 input channels, sub-stage)
    32 x 64 sixteen-byte units -- fp16 fragments of the weights and block-scaled e2m3 fragments of their fp16 roundings and of the
    remainders (csrc/costreg_mx.h).  A few hundred thousand elements: cut on every call, like the bf16x3 pieces."""
    _req(weight, "weight", dim=5)
    Cout, Cin = weight.shape[:2]
    lib = _lib.load()
    nbytes = int(lib.mvsdet_split_conv_weight_mx_bytes(int(Cout), int(Cin)))
    if nbytes == 0 or tuple(weight.shape[2:]) != (3, 3, 3):
        raise ValueError(f"split_conv_weight_mx: weight {tuple(weight.shape)} is not (64 m, Cin, 3, 3, 3)")
    w = weight.detach().contiguous()
    out = torch.empty((Cout // 64, (Cin + 7) // 8, 2, 32, 64, 4), dtype=torch.int32, device=w.device)
    with torch.cuda.device(w.device):
        _lib.check(lib.mvsdet_split_conv_weight_mx(_lib.ptr(w), _lib.ptr(out), int(Cout), int(Cin), _stream(w)), "split_conv_weight_mx")
    return out


def conv3d_k3_fp16mx(x: Tensor, weight_split_mx: Tensor, scale: Optional[Tensor], shift: Optional[Tensor], relu: bool,
                     outputs=("f32",), scl_out: Optional[SclTensor] = None, pscl_out: Optional[PsclTensor] = None):
    """Conv3d(Cin -> Cout = 64 m, kernel 3, stride 1, padding 1, no bias) [+ affine] [+ ReLU] of the fp32 (N,Cin,D,H,W) tensor read in
    place (any view with w stride 1), on ONE fp16 and TWO block-scaled FP6 products per fp32-equivalent product (csrc/costreg_mx.h:
    v_mfma_f32_16x16x32_f16 + v_mfma_scale_f32_16x16x128_f8f6f4) instead of bf16x3's three: mvsnet.py:76, the layer that reads the
    variance volume.  weight_split_mx: `split_conv_weight_mx`.  outputs as for `conv3d_k3_bf16x3`."""
    import ctypes
    _req(x, "x", dim=5)
    if x.stride(4) != 1 or min(x.stride()) < 0:
        x = x.contiguous()
    N, Cin, D, H, W = x.shape
    dev = x.device
    if (weight_split_mx.dtype != torch.int32 or weight_split_mx.dim() != 6 or not weight_split_mx.is_contiguous()
            or tuple(weight_split_mx.shape[1:]) != ((Cin + 7) // 8, 2, 32, 64, 4) or weight_split_mx.device != dev):
        raise ValueError(f"conv3d_k3_fp16mx: weight_split_mx {tuple(weight_split_mx.shape)} {weight_split_mx.dtype} does not match Cin={Cin}")
    Cout = weight_split_mx.shape[0] * 64
    scale, shift = _check_affine("conv3d_k3_fp16mx", scale, shift, Cout)
    outputs, res = _conv_outputs("conv3d_k3_fp16mx", outputs, (N, Cout, D, H, W), dev, scl_out, pscl_out)
    xstr = (ctypes.c_int64 * 4)(*[int(v) for v in x.stride()[:4]])
    with torch.cuda.device(dev):
        _lib.check(_lib.load().mvsdet_conv3d_k3_fp16mx_f32in(_lib.ptr(x), xstr, _lib.ptr(weight_split_mx), _lib.ptr(scale), _lib.ptr(shift),
                                                             _lib.ptr(res.get("f32")), _lib.ptr(res["scl"].data) if "scl" in res else None,
                                                             _lib.ptr(res["pscl"].data) if "pscl" in res else None, N, Cin, Cout, D, H, W,
                                                             int(relu), _lib.current_stream(dev)), "conv3d_k3_fp16mx")
    return _ret(outputs, res)


def conv3d_k3_bf16x3_stats(x, weight_split: Tensor, pivot: Optional[Tensor] = None) -> Tuple[Tensor, Tensor]:
    """`conv3d_k3_bf16x3(x, weight_split, None, None, False)` in front of a training-mode BatchNorm (module.py:26-37): the raw
    fp32 output and, from the kernel's epilogue, the per-channel partial sums of the outputs and of their squares -- a float64
    tensor (Cout, parts, 2), one entry per block of the grid -- that `bn3d_relu_train(..., parts=)` finishes: the BatchNorm then
    reads the tensor once.  pivot (Cout floats, e.g. the BatchNorm's running mean): the sums are those of (value - pivot_c), which
    keeps fp32 lane sums from cancelling when a channel's mean is far from zero; pass the same vector to `bn3d_relu_train`.
    Needs the 16x16x32 form of the kernel (library option conv_mfma16, the default)."""
    import ctypes
    scl = isinstance(x, SclTensor)
    if scl:
        N, Cin, D, H, W = x.shape
        dev = x.data.device
    else:
        _req(x, "x", dim=5)
        if x.stride(4) != 1 or min(x.stride()) < 0:
            x = x.contiguous()
        N, Cin, D, H, W = x.shape
        dev = x.device
    if weight_split.dtype != torch.bfloat16 or weight_split.dim() != 7 or tuple(weight_split.shape[1:]) != ((Cin + 7) // 8, 14, 2, 2, 64, 8):
        raise ValueError(f"conv3d_k3_bf16x3_stats: weight_split {tuple(weight_split.shape)} does not match Cin={Cin}")
    Cout = weight_split.shape[0] * 64
    weight_split = weight_split.contiguous()
    if pivot is not None:
        if pivot.dtype != torch.float32 or pivot.numel() != Cout or pivot.device != dev:
            raise ValueError(f"conv3d_k3_bf16x3_stats: pivot must be {Cout} fp32 values on {dev}")
        pivot = pivot.detach().contiguous()
    lib = _lib.load()
    parts = int(lib.mvsdet_conv3d_k3_bf16x3_stats_parts(N, D, H, W, 0 if scl else 1))
    out = torch.empty((N, Cout, D, H, W), dtype=torch.float32, device=dev)
    stats = torch.empty((Cout, parts, 2), dtype=torch.float64, device=dev)
    xstr = None if scl else (ctypes.c_int64 * 4)(*[int(v) for v in x.stride()[:4]])
    with torch.cuda.device(dev):
        _lib.check(lib.mvsdet_conv3d_k3_bf16x3_stats(_lib.ptr(x.data) if scl else None, None if scl else _lib.ptr(x), xstr,
                                                     _lib.ptr(weight_split), _lib.ptr(out), _lib.ptr(stats), stats.numel() * 8,
                                                     _lib.ptr(pivot), N, Cin, Cout, D, H, W, _lib.current_stream(dev)),
                   "conv3d_k3_bf16x3_stats")
    return out, stats


def conv3d_k3_s2_bf16x3(x, weight_split: Tensor, scale: Optional[Tensor], shift: Optional[Tensor], relu: bool,
                        outputs=("f32",), scl_out: Optional[SclTensor] = None):
    """Conv3d(Cin -> Cout = 64*m, kernel 3, stride 2, padding 1, no bias) [+ affine] [+ ReLU] (mvsnet.py:77,79) on the bf16
    matrix cores, three-term split: x (N,Cin,D,H,W) fp32 (w stride 1) or its PsclTensor form (a producing layer's "pscl"
    output: the class tiles then arrive by LDS-DMA) -> (N,Cout,(D-1)//2+1,(H-1)//2+1,(W-1)//2+1) as "f32" and / or "scl";
    weight_split = split_conv_weight(weight, order=1)."""
    import ctypes
    pin = isinstance(x, PsclTensor)
    if pin:
        N, Cin, D, H, W = x.shape
        dev = x.data.device
    else:
        _req(x, "x", dim=5)
        if x.stride(4) != 1 or min(x.stride()) < 0:
            x = x.contiguous()
        N, Cin, D, H, W = x.shape
        dev = x.device
    if weight_split.dtype != torch.bfloat16 or weight_split.dim() != 7 or tuple(weight_split.shape[1:]) != ((Cin + 7) // 8, 14, 2, 2, 64, 8):
        raise ValueError(f"conv3d_k3_s2_bf16x3: weight_split {tuple(weight_split.shape)} does not match Cin={Cin}")
    Cout = weight_split.shape[0] * 64
    scale, shift = _check_affine("conv3d_k3_s2_bf16x3", scale, shift, Cout)
    oshape = (N, Cout, (D - 1) // 2 + 1, (H - 1) // 2 + 1, (W - 1) // 2 + 1)
    outputs, res = _conv_outputs("conv3d_k3_s2_bf16x3", outputs, oshape, dev, scl_out, None, allow_pscl=False)
    weight_split = weight_split.contiguous()
    xstr = None if pin else (ctypes.c_int64 * 4)(*[int(v) for v in x.stride()[:4]])
    lib = _lib.load()
    wbytes = lib.mvsdet_conv3d_k3_s2_bf16x3_workspace_bytes(N, Cin, Cout, D, H, W) if tuple(outputs) == ("f32",) else 0   # small volumes
    ws = torch.empty((wbytes // 4,), dtype=torch.float32, device=dev) if wbytes else None
    with torch.cuda.device(dev):
        _lib.check(lib.mvsdet_conv3d_k3_s2_bf16x3_io(None if pin else _lib.ptr(x), xstr, _lib.ptr(x.data) if pin else None,
                                                     _lib.ptr(weight_split), _lib.ptr(scale), _lib.ptr(shift), _lib.ptr(res.get("f32")),
                                                     _lib.ptr(res["scl"].data) if "scl" in res else None, _lib.ptr(ws), wbytes, N, Cin,
                                                     Cout, D, H, W, int(relu), _lib.current_stream(dev)), "conv3d_k3_s2_bf16x3")
    return _ret(outputs, res)


def convT3d_k3_s2_bf16x3(x, weight_split: Tensor, scale: Optional[Tensor], shift: Optional[Tensor],
                         residual: Optional[Tensor], relu: bool, outputs=("f32",), scl_out: Optional[SclTensor] = None):
    """ConvTranspose3d(Cin -> Cout = 64*m, kernel 3, stride 2, padding 1, output_padding 1, no bias) [+ affine] [+ ReLU]
    [+ residual, added last] (mvsnet.py:92-100,110-111) on the bf16 matrix cores, three-term split: x (N,Cin,D,H,W) fp32 (or
    its SclTensor) -> (N,Cout,2D,2H,2W) as "f32" and / or "scl"; weight_split = split_conv_weight(weight (Cin,Cout,3,3,3), order=2)."""
    xs = x if isinstance(x, SclTensor) else scl_pack(x)
    N, Cin, D, H, W = xs.shape
    if weight_split.dtype != torch.bfloat16 or weight_split.dim() != 7 or tuple(weight_split.shape[1:]) != ((Cin + 7) // 8, 14, 2, 2, 64, 8):
        raise ValueError(f"convT3d_k3_s2_bf16x3: weight_split {tuple(weight_split.shape)} does not match Cin={Cin}")
    Cout = weight_split.shape[0] * 64
    dev = xs.data.device
    scale, shift = _check_affine("convT3d_k3_s2_bf16x3", scale, shift, Cout)
    oshape = (N, Cout, 2 * D, 2 * H, 2 * W)
    outputs, res = _conv_outputs("convT3d_k3_s2_bf16x3", outputs, oshape, dev, scl_out, None, allow_pscl=False)
    if residual is not None:
        _req(residual, "residual", dim=5)
        if tuple(residual.shape) != oshape:
            raise ValueError(f"convT3d_k3_s2_bf16x3: residual {tuple(residual.shape)} != output {oshape}")
        residual = residual.contiguous()
    weight_split = weight_split.contiguous()
    with torch.cuda.device(dev):
        _lib.check(_lib.load().mvsdet_convT3d_k3_s2_bf16x3_io(_lib.ptr(xs.data), _lib.ptr(weight_split), _lib.ptr(scale), _lib.ptr(shift),
                                                              _lib.ptr(residual), _lib.ptr(res.get("f32")),
                                                              _lib.ptr(res["scl"].data) if "scl" in res else None, N, Cin, Cout, D, H, W,
                                                              int(relu), _lib.current_stream(dev)), "convT3d_k3_s2_bf16x3")
    return _ret(outputs, res)


def convT3d_k3_s2_bf16x3_stats(x, weight_split: Tensor, pivot: Optional[Tensor] = None) -> Optional[Tuple[Tensor, Tensor]]:
    """`convT3d_k3_s2_bf16x3(x, weight_split, None, None, None, False)` in front of a training-mode BatchNorm (mvsnet.py:92-100): the
    raw fp32 output and the per-channel partial sums (float64 (Cout, parts, 2), sums of value - pivot_c and of its square) that
    `bn3d_relu_train(..., parts=, pivot=)` finishes.  None where the shape has no statistics form (the plain call applies)."""
    xs = x if isinstance(x, SclTensor) else scl_pack(x)
    N, Cin, D, H, W = xs.shape
    if weight_split.dtype != torch.bfloat16 or weight_split.dim() != 7 or tuple(weight_split.shape[1:]) != ((Cin + 7) // 8, 14, 2, 2, 64, 8):
        raise ValueError(f"convT3d_k3_s2_bf16x3_stats: weight_split {tuple(weight_split.shape)} does not match Cin={Cin}")
    Cout = weight_split.shape[0] * 64
    dev = xs.data.device
    lib = _lib.load()
    parts = int(lib.mvsdet_convT3d_k3_s2_bf16x3_stats_parts(N, D, H, W))
    if parts == 0:
        return None
    if pivot is not None:
        if pivot.dtype != torch.float32 or pivot.numel() != Cout or pivot.device != dev:
            raise ValueError(f"convT3d_k3_s2_bf16x3_stats: pivot must be {Cout} fp32 values on {dev}")
        pivot = pivot.detach().contiguous()
    weight_split = weight_split.contiguous()
    out = torch.empty((N, Cout, 2 * D, 2 * H, 2 * W), dtype=torch.float32, device=dev)
    stats = torch.empty((Cout, parts, 2), dtype=torch.float64, device=dev)
    with torch.cuda.device(dev):
        _lib.check(lib.mvsdet_convT3d_k3_s2_bf16x3_stats(_lib.ptr(xs.data), _lib.ptr(weight_split), _lib.ptr(out), _lib.ptr(stats),
                                                         stats.numel() * 8, _lib.ptr(pivot), N, Cin, Cout, D, H, W,
                                                         _lib.current_stream(dev)), "convT3d_k3_s2_bf16x3_stats")
    return out, stats


# ------------------------------------------------------------------------------------------- misc
def device_copy(src: Tensor, dst: Tensor):
    """float4 device-to-device copy kernel (bench.py's achievable-HBM yardstick)."""
    _req(src, "src")
    _req(dst, "dst")
    assert src.is_contiguous() and dst.is_contiguous() and src.numel() == dst.numel()
    with torch.cuda.device(src.device):
        _lib.check(_lib.load().mvsdet_copy_f32(_lib.ptr(src), _lib.ptr(dst), src.numel(), _stream(src)), "copy")


# ------------------------------------------------------------------------------------------- cost network head (f-1)
def store_pattern_probe(var: Tensor, W: int, tile_w: int, planes_per_block: int = 0) -> None:
    """Overwrites `var` -- an (N,C,D,H,pitch) fp32 or fp16 buffer -- with the sweep's store stream alone (bench.py: the ceiling
    of the output layout on the box at hand)."""
    if var.dtype not in (torch.float32, torch.float16) or not var.is_cuda or not var.is_contiguous() or var.dim() != 5:
        raise ValueError("store_pattern_probe: var must be a contiguous 5-D fp32 or fp16 CUDA tensor")
    N, C, D, H, pitch = var.shape
    lib = _lib.load()
    fn = lib.mvsdet_store_pattern_probe_f32 if var.dtype == torch.float32 else lib.mvsdet_store_pattern_probe_f16
    with torch.cuda.device(var.device):
        _lib.check(fn(_lib.ptr(var), N, C, D, H, int(W), int(pitch), int(tile_w), int(planes_per_block), _stream(var)),
                   "store_pattern_probe")


def conv3d_k3_cout2_sum(x: Tensor, x2: Optional[Tensor], weight: Tensor, bias: Optional[Tensor]) -> Tensor:
    """Conv3d(Cin -> 2, kernel 3, padding 1, bias) of mvs_models/mvsnet.py:102,112 on x + x2 (N,Cin,D,H,W) -> (N,2,D,H,W): the
    skip addition of mvsnet.py:111 formed while the head stages its input (forward only; x2 None: x alone)."""
    _req(x, "x", dim=5)
    _req(weight, "weight", dim=5)
    N, Cin, D, H, W = x.shape
    if tuple(weight.shape) != (2, Cin, 3, 3, 3):
        raise ValueError(f"conv3d_k3_cout2_sum: weight {tuple(weight.shape)} != (2,{Cin},3,3,3)")
    if x2 is not None:
        _req(x2, "x2", dim=5)
        if x2.shape != x.shape:
            raise ValueError(f"conv3d_k3_cout2_sum: x2 {tuple(x2.shape)} != x {tuple(x.shape)}")
        if W % 4:
            return conv3d_k3_cout2(x + x2, weight, bias)
        x2 = x2.contiguous()
    if bias is not None:
        _req(bias, "bias", dim=1)
        bias = bias.contiguous()
    x, weight = x.contiguous(), weight.contiguous()
    out = torch.empty((N, 2, D, H, W), dtype=torch.float32, device=x.device)
    with torch.cuda.device(x.device):
        _lib.check(_lib.load().mvsdet_conv3d_k3_cout2_sum_f32(_lib.ptr(x), _lib.ptr(x2), _lib.ptr(weight), _lib.ptr(bias), _lib.ptr(out),
                                                              N, Cin, D, H, W, _stream(x)), "conv3d_k3_cout2_sum")
    return out


@torch.library.custom_op(f"{_NS}::conv3d_k3_cout2", mutates_args=(), device_types="cuda")
def conv3d_k3_cout2(x: Tensor, weight: Tensor, bias: Optional[Tensor]) -> Tensor:
    """Conv3d(Cin -> 2, kernel 3, stride 1, padding 1) of mvs_models/mvsnet.py:102 on (N,Cin,D,H,W) -> (N,2,D,H,W).
    Forward only: under autograd use torch's convolution (mvsdet_amd.costreg does)."""
    _req(x, "x", dim=5)
    _req(weight, "weight", dim=5)
    N, Cin, D, H, W = x.shape
    if tuple(weight.shape) != (2, Cin, 3, 3, 3):
        raise ValueError(f"conv3d_k3_cout2: weight {tuple(weight.shape)} != (2,{Cin},3,3,3)")
    if bias is not None:
        _req(bias, "bias", dim=1)
        if bias.numel() != 2:
            raise ValueError("conv3d_k3_cout2: bias must have 2 elements")
        bias = bias.contiguous()
    x, weight = x.contiguous(), weight.contiguous()
    out = torch.empty((N, 2, D, H, W), dtype=torch.float32, device=x.device)
    with torch.cuda.device(x.device):
        _lib.check(_lib.load().mvsdet_conv3d_k3_cout2_f32(_lib.ptr(x), _lib.ptr(weight), _lib.ptr(bias), _lib.ptr(out), N, Cin,
                                                          D, H, W, _stream(x)), "conv3d_k3_cout2")
    return out


@conv3d_k3_cout2.register_fake
def _(x, weight, bias):
    return x.new_empty((x.shape[0], 2) + tuple(x.shape[2:]))


def permute_conv_weight(weight: Tensor) -> Tensor:
    """(Cout,Cin,3,3,3) -> the [c][kd][kh][kw][o] layout conv3d_k3_mfma reads (Cin padded to even with zeros)."""
    cout, cin = weight.shape[:2]
    if cout % 64 or tuple(weight.shape[2:]) != (3, 3, 3):
        raise ValueError(f"conv3d_k3_mfma: weight {tuple(weight.shape)} != (64*m,Cin,3,3,3)")
    w = weight.detach().permute(1, 2, 3, 4, 0).contiguous()
    if cin % 2:
        w = torch.cat([w, w.new_zeros((1, 3, 3, 3, cout))], 0)
    return w



@torch.library.custom_op(f"{_NS}::conv3d_k3_mfma", mutates_args=(), device_types="cuda")
def conv3d_k3_mfma(x: Tensor, weight_perm: Tensor, scale: Optional[Tensor], shift: Optional[Tensor], relu: bool,
                   stride: int = 1, residual: Optional[Tensor] = None) -> Tensor:
    """Conv3d(Cin -> Cout = 64*m, kernel 3, stride 1 or 2, padding 1, no bias) [+ per-channel affine + ReLU] of
    mvs_models/mvsnet.py:76-82 on the fp32 matrix cores: x (N,Cin,D,H,W), weight_perm = permute_conv_weight(weight)
    -> (N,Cout,D',H',W').  Forward only."""
    _req(x, "x", dim=5)
    _req(weight_perm, "weight_perm", dim=5)
    N, Cin, D, H, W = x.shape
    Cout = weight_perm.shape[4]
    if tuple(weight_perm.shape[:4]) != (Cin + Cin % 2, 3, 3, 3) or Cout % 64:
        raise ValueError(f"conv3d_k3_mfma: weight_perm {tuple(weight_perm.shape)} does not match Cin={Cin}")
    if (scale is None) != (shift is None):
        raise ValueError("conv3d_k3_mfma: scale and shift come together")
    if scale is not None:
        _req(scale, "scale", dim=1)
        _req(shift, "shift", dim=1)
        if scale.numel() != Cout or shift.numel() != Cout:
            raise ValueError(f"conv3d_k3_mfma: scale / shift must have {Cout} elements")
        scale, shift = scale.contiguous(), shift.contiguous()
    if stride not in (1, 2):
        raise ValueError("conv3d_k3_mfma: stride must be 1 or 2")
    x, weight_perm = x.contiguous(), weight_perm.contiguous()
    od, oh, ow = (D - 1) // stride + 1, (H - 1) // stride + 1, (W - 1) // stride + 1
    out = torch.empty((N, Cout, od, oh, ow), dtype=torch.float32, device=x.device)
    if residual is not None:   # added between the affine and the ReLU (imvoxel_neck.py:227-229); stride 1 only
        _req(residual, "residual", dim=5)
        if stride != 1 or tuple(residual.shape) != tuple(out.shape):
            raise ValueError(f"conv3d_k3_mfma: residual {tuple(residual.shape)} needs stride 1 and the output shape {tuple(out.shape)}")
        residual = residual.contiguous()
    lib = _lib.load()
    # a small volume (the neck at one scene) is split over the input channels: partial sums in a workspace
    wbytes = lib.mvsdet_conv3d_k3_mfma_workspace_bytes(N, Cin, Cout, D, H, W, stride)
    ws = torch.empty((wbytes // 4,), dtype=torch.float32, device=x.device) if wbytes else None
    with torch.cuda.device(x.device):
        _lib.check(lib.mvsdet_conv3d_k3_mfma_ws_f32(_lib.ptr(x), _lib.ptr(weight_perm), _lib.ptr(scale), _lib.ptr(shift),
                                                    _lib.ptr(residual), _lib.ptr(out), _lib.ptr(ws), wbytes, N, Cin, Cout, D, H, W,
                                                    stride, int(relu), _stream(x)), "conv3d_k3_mfma")
    return out


@conv3d_k3_mfma.register_fake
def _(x, weight_perm, scale, shift, relu, stride=1, residual=None):
    return x.new_empty((x.shape[0], weight_perm.shape[4]) + tuple((s - 1) // stride + 1 for s in x.shape[2:]))



def permute_convT_weight(weight: Tensor) -> Tensor:
    """ConvTranspose3d weight (Cin,Cout,3,3,3) -> the [c][kd][kh][kw][o] layout convT3d_k3_s2_mfma reads."""
    cin, cout = weight.shape[:2]
    if cout % 64 or tuple(weight.shape[2:]) != (3, 3, 3):
        raise ValueError(f"convT3d_k3_s2_mfma: weight {tuple(weight.shape)} != (Cin,64*m,3,3,3)")
    w = weight.detach().permute(0, 2, 3, 4, 1).contiguous()
    if cin % 2:
        w = torch.cat([w, w.new_zeros((1, 3, 3, 3, cout))], 0)
    return w


@torch.library.custom_op(f"{_NS}::convT3d_k3_s2_mfma", mutates_args=(), device_types="cuda")
def convT3d_k3_s2_mfma(x: Tensor, weight_perm: Tensor, scale: Optional[Tensor], shift: Optional[Tensor],
                       residual: Optional[Tensor], relu: bool) -> Tensor:
    """ConvTranspose3d(kernel 3, stride 2, padding 1, output_padding 1, no bias) [+ affine + ReLU] [+ residual] of
    mvs_models/mvsnet.py:92-100 on the fp32 matrix cores: x (N,Cin,D,H,W) -> (N,Cout,2D,2H,2W).  Forward only."""
    _req(x, "x", dim=5)
    _req(weight_perm, "weight_perm", dim=5)
    N, Cin, D, H, W = x.shape
    Cout = weight_perm.shape[4]
    if tuple(weight_perm.shape[:4]) != (Cin + Cin % 2, 3, 3, 3) or Cout % 64:
        raise ValueError(f"convT3d_k3_s2_mfma: weight_perm {tuple(weight_perm.shape)} does not match Cin={Cin}")
    if (scale is None) != (shift is None):
        raise ValueError("convT3d_k3_s2_mfma: scale and shift come together")
    if scale is not None:
        _req(scale, "scale", dim=1)
        _req(shift, "shift", dim=1)
        if scale.numel() != Cout or shift.numel() != Cout:
            raise ValueError(f"convT3d_k3_s2_mfma: scale / shift must have {Cout} elements")
        scale, shift = scale.contiguous(), shift.contiguous()
    oshape = (N, Cout, 2 * D, 2 * H, 2 * W)
    if residual is not None:
        _req(residual, "residual", dim=5)
        if tuple(residual.shape) != oshape:
            raise ValueError(f"convT3d_k3_s2_mfma: residual {tuple(residual.shape)} != {oshape}")
        residual = residual.contiguous()
    x, weight_perm = x.contiguous(), weight_perm.contiguous()
    out = torch.empty(oshape, dtype=torch.float32, device=x.device)
    with torch.cuda.device(x.device):
        _lib.check(_lib.load().mvsdet_convT3d_k3_s2_mfma_f32(_lib.ptr(x), _lib.ptr(weight_perm), _lib.ptr(scale),
                                                             _lib.ptr(shift), _lib.ptr(residual), _lib.ptr(out), N, Cin, Cout,
                                                             D, H, W, int(relu), _stream(x)), "convT3d_k3_s2_mfma")
    return out


@convT3d_k3_s2_mfma.register_fake
def _(x, weight_perm, scale, shift, residual, relu):
    return x.new_empty((x.shape[0], weight_perm.shape[4]) + tuple(2 * s for s in x.shape[2:]))


@torch.library.custom_op(f"{_NS}::conv3d_k3_dw", mutates_args=(), device_types="cuda")
def conv3d_k3_dw(x: Tensor, grad_out: Tensor, nsplit: int = 0, stride: int = 1, bf16x3: bool = False) -> Tensor:
    """Weight gradient of Conv3d(kernel 3, stride 1 or 2, padding 1): x (N,Cin,D,H,W), grad_out (N,Cout,D/s,H/s,W/s) ->
    (Cout,Cin,3,3,3), on the fp32 matrix cores; `nsplit` voxel splits are accumulated separately and summed
    (0 = automatic: up to 128 splits, fewer when Cin*Cout is large so that the partial sums stay below 256 MB).
    With stride 2 and the tensors exchanged (x = grad of the output, grad_out = the input) this is the weight gradient of
    ConvTranspose3d(kernel 3, stride 2, padding 1, output_padding 1) in its own (Cin,Cout,3,3,3) layout.
    bf16x3: the sum runs on the bf16 matrix cores with three-term split operands (csrc/costreg_dw_bf16.hip; stride 2:
    csrc/costreg_dw_s2_bf16.hip, blocks of 64 grad_out x 16 x channels); rows are read as float4: W a multiple of 4 (stride 2: 8)."""
    _req(x, "x", dim=5)
    _req(grad_out, "grad_out", dim=5)
    if stride not in (1, 2):
        raise ValueError(f"conv3d_k3_dw: stride {stride} not in (1, 2)")
    if bf16x3 and x.shape[-1] % (4 * stride):
        raise ValueError(f"conv3d_k3_dw: the bf16x3 kernel reads rows as float4, W={x.shape[-1]} is not a multiple of {4 * stride}")
    N, Cin, D, H, W = x.shape
    Cout = grad_out.shape[1]
    if any(v % stride for v in (D, H, W)) or tuple(grad_out.shape) != (N, Cout, D // stride, H // stride, W // stride):
        raise ValueError(f"conv3d_k3_dw: grad_out {tuple(grad_out.shape)} does not match x {tuple(x.shape)} at stride {stride}")
    x, grad_out = x.contiguous(), grad_out.contiguous()
    lib = _lib.load()
    if nsplit <= 0:
        nsplit = max(8, min(128, (256 << 20) // (Cout * Cin * 108)))
        if bf16x3:   # one block of 12 waves per CU: 256 / (channel blocks) splits, a multiple of 8 (one split = one XCD)
            cblocks = ((Cin + 31) // 32) * ((Cout + 31) // 32) if stride == 1 else ((Cin + 15) // 16) * ((Cout + 63) // 64)
            nsplit = max(8, min(64, (256 // cblocks) // 8 * 8))
    if bf16x3:       # the kernels keep the columns of a split in a 4096-entry table
        cols = N * ((H + 3) // 4) * ((W + 15) // 16) if stride == 1 else N * ((H // 2 + 3) // 4) * ((W // 2 + 7) // 8)
        nsplit = max(nsplit, -(-cols // 4096))
    pbytes = lib.mvsdet_conv3d_k3_dw_partial_bytes(Cin, Cout, nsplit)
    partial = torch.empty((nsplit, Cout, Cin, 27), dtype=torch.float32, device=x.device)
    fn = lib.mvsdet_conv3d_k3_dw_mfma_f32 if stride == 1 else lib.mvsdet_conv3d_k3_s2_dw_mfma_f32
    if bf16x3:
        fn = lib.mvsdet_conv3d_k3_dw_bf16x3 if stride == 1 else lib.mvsdet_conv3d_k3_s2_dw_bf16x3
    with torch.cuda.device(x.device):
        _lib.check(fn(_lib.ptr(x), _lib.ptr(grad_out), _lib.ptr(partial), pbytes, nsplit, N, Cin, Cout, D, H, W, _stream(x)),
                   "conv3d_k3_dw")
    return partial.sum(0).view(Cout, Cin, 3, 3, 3)


@conv3d_k3_dw.register_fake
def _(x, grad_out, nsplit=0, stride=1, bf16x3=False):
    return x.new_empty((grad_out.shape[1], x.shape[1], 3, 3, 3))


@torch.library.custom_op(f"{_NS}::conv3d_k3_cout2_backward", mutates_args=(), device_types="cuda")
def conv3d_k3_cout2_backward(x: Tensor, weight: Tensor, grad_out: Tensor, nsplit: int = 32,
                             bf16x3: bool = False) -> Tuple[Tensor, Tensor]:
    """Gradients of conv3d_k3_cout2 w.r.t. x (N,Cin,D,H,W) and weight (2,Cin,3,3,3) from grad_out (N,2,D,H,W).
    bf16x3: the weight gradient on the bf16 matrix cores with three-term split operands where its kernel applies
    (Cin in {16, 32, 64}, W % 4 == 0), three blocks per CU; otherwise (and by default) the fp32 kernel with `nsplit` voxel splits."""
    _req(x, "x", dim=5)
    _req(weight, "weight", dim=5)
    _req(grad_out, "grad_out", dim=5)
    N, Cin, D, H, W = x.shape
    if tuple(weight.shape) != (2, Cin, 3, 3, 3) or tuple(grad_out.shape) != (N, 2, D, H, W):
        raise ValueError("conv3d_k3_cout2_backward: shape mismatch")
    x, weight, grad_out = x.contiguous(), weight.contiguous(), grad_out.contiguous()
    lib = _lib.load()
    gx = torch.empty_like(x)
    # the library's own shape gate (channel count, W % 4, the LDS of a row stage: W up to ~470 at Cin = 64); wider maps keep the
    # fp32 kernel instead of raising from the entry point
    on_mfma = (bool(bf16x3) and bool(lib.mvsdet_conv3d_k3_cout2_dw_bf16x3_ok(int(Cin), int(W)))
               and x.data_ptr() % 16 == 0 and grad_out.data_ptr() % 16 == 0)
    if on_mfma:
        nsplit = min(768, N * D * H)   # blocks of four waves, a wave takes whole x rows (n, d, h)
    partial = torch.empty((nsplit, 2, Cin, 27), dtype=torch.float32, device=x.device)
    with torch.cuda.device(x.device):
        _lib.check(lib.mvsdet_conv3d_k3_cout2_dx_f32(_lib.ptr(grad_out), _lib.ptr(weight), _lib.ptr(gx), N, Cin, D, H, W,
                                                     _stream(x)), "conv3d_k3_cout2_dx")
        fn = lib.mvsdet_conv3d_k3_cout2_dw_bf16x3 if on_mfma else lib.mvsdet_conv3d_k3_cout2_dw_f32
        _lib.check(fn(_lib.ptr(x), _lib.ptr(grad_out), _lib.ptr(partial), partial.numel() * 4, nsplit, N, Cin, D, H, W,
                      _stream(x)), "conv3d_k3_cout2_dw")
    return gx, partial.sum(0).view(2, Cin, 3, 3, 3)


@conv3d_k3_cout2_backward.register_fake
def _(x, weight, grad_out, nsplit=32, bf16x3=False):
    return torch.empty_like(x), torch.empty_like(weight)


@torch.library.custom_op(f"{_NS}::bn3d_relu_train", mutates_args=(), device_types="cuda")
def bn3d_relu_train(x: Tensor, weight: Optional[Tensor], bias: Optional[Tensor], eps: float,
                    relu: bool, residual: Optional[Tensor] = None, parts: Optional[Tensor] = None,
                    pivot: Optional[Tensor] = None) -> Tuple[Tensor, Tensor, Tensor]:
    """Training-mode BatchNorm3d [+ ReLU] (mvs_models/module.py:26-37) on the batch statistics of x (N,C,D,H,W) fp32 ->
    (out, batch mean, 1/sqrt(biased batch variance + eps)).  The running statistics are the caller's to update (a custom
    operator with an autograd formula must not mutate its inputs): mvsdet_amd.costreg does it from the returned vectors.
    residual (same shape as x): out = [relu](bn(x)) + residual in the same pass (mvsnet.py:109-111, the skip additions).
    parts: float64 (C, n, 2) partial sums of x and of its squares per channel, left by the convolution that produced x
    (`conv3d_k3_bf16x3_stats`): the statistics pass over x is skipped; pivot: the vector that call was given (None = zeros)."""
    _req(x, "x", dim=5)
    if residual is not None:
        _req(residual, "residual", dim=5)
        if residual.shape != x.shape:
            raise ValueError(f"bn3d_relu_train: residual {tuple(residual.shape)} != x {tuple(x.shape)}")
        residual = residual.contiguous()
    N, C = x.shape[:2]
    vol = x[0, 0].numel()
    for t, name in ((weight, "weight"), (bias, "bias")):
        if t is not None:
            _req(t, name, dim=1)
            if t.numel() != C:
                raise ValueError(f"bn3d_relu_train: {name} must have {C} elements")
    x = x.contiguous()
    weight = None if weight is None else weight.contiguous()
    bias = None if bias is None else bias.contiguous()
    out = torch.empty_like(x)
    mean = torch.empty(C, dtype=torch.float32, device=x.device)
    invstd = torch.empty_like(mean)
    lib = _lib.load()
    wb = lib.mvsdet_bn3d_workspace_bytes(C)
    ws = torch.empty(wb // 8, dtype=torch.float64, device=x.device)
    if parts is not None:
        if parts.dtype != torch.float64 or parts.dim() != 3 or parts.shape[0] != C or parts.shape[2] != 2 or parts.device != x.device:
            raise ValueError(f"bn3d_relu_train: parts must be float64 ({C}, n, 2) on {x.device}, got {parts.dtype} {tuple(parts.shape)}")
        parts = parts.contiguous()
        if pivot is not None:
            if pivot.dtype != torch.float32 or pivot.numel() != C or pivot.device != x.device:
                raise ValueError(f"bn3d_relu_train: pivot must be {C} fp32 values on {x.device}")
            pivot = pivot.contiguous()
    elif pivot is not None:
        raise ValueError("bn3d_relu_train: a pivot belongs to partial sums")
    with torch.cuda.device(x.device):
        if parts is not None:
            _lib.check(lib.mvsdet_bn3d_relu_train_fwd_parts_f32(_lib.ptr(x), _lib.ptr(parts), int(parts.shape[1]), _lib.ptr(pivot), _lib.ptr(weight), _lib.ptr(bias),
                                                                _lib.ptr(residual), None, None, _lib.ptr(out), _lib.ptr(mean), _lib.ptr(invstd),
                                                                _lib.ptr(ws), wb, N, C, vol, 0.0, float(eps), int(relu), _stream(x)),
                       "bn3d_relu_train")
        else:
            _lib.check(lib.mvsdet_bn3d_relu_train_fwd_res_f32(_lib.ptr(x), _lib.ptr(weight), _lib.ptr(bias), _lib.ptr(residual), None, None,
                                                              _lib.ptr(out), _lib.ptr(mean), _lib.ptr(invstd), _lib.ptr(ws), wb, N, C, vol,
                                                              0.0, float(eps), int(relu), _stream(x)), "bn3d_relu_train")
    return out, mean, invstd


@bn3d_relu_train.register_fake
def _(x, weight, bias, eps, relu, residual=None, parts=None, pivot=None):
    return torch.empty_like(x), x.new_empty(x.shape[1]), x.new_empty(x.shape[1])


@torch.library.custom_op(f"{_NS}::bn3d_relu_backward", mutates_args=(), device_types="cuda")
def bn3d_relu_backward(x: Tensor, grad_out: Tensor, weight: Optional[Tensor], bias: Optional[Tensor], save_mean: Tensor,
                       save_invstd: Tensor, relu: bool) -> Tuple[Tensor, Tensor, Tensor]:
    """Backward of bn3d_relu_train -> (grad_x, grad_weight, grad_bias)."""
    _req(x, "x", dim=5)
    _req(grad_out, "grad_out", dim=5)
    if grad_out.shape != x.shape:
        raise ValueError("bn3d_relu_backward: grad_out does not match x")
    N, C = x.shape[:2]
    vol = x[0, 0].numel()
    x, grad_out = x.contiguous(), grad_out.contiguous()
    weight = None if weight is None else weight.contiguous()
    bias = None if bias is None else bias.contiguous()
    gx = torch.empty_like(x)
    gw = torch.empty(C, dtype=torch.float32, device=x.device)
    gb = torch.empty_like(gw)
    lib = _lib.load()
    wb = lib.mvsdet_bn3d_workspace_bytes(C)
    ws = torch.empty(wb // 8, dtype=torch.float64, device=x.device)
    with torch.cuda.device(x.device):
        _lib.check(lib.mvsdet_bn3d_relu_bwd_f32(_lib.ptr(x), _lib.ptr(grad_out), _lib.ptr(weight), _lib.ptr(bias), _lib.ptr(save_mean),
                                                _lib.ptr(save_invstd), _lib.ptr(gx), _lib.ptr(gw), _lib.ptr(gb), _lib.ptr(ws), wb, N, C,
                                                vol, int(relu), _stream(x)), "bn3d_relu_backward")
    return gx, gw, gb


@bn3d_relu_backward.register_fake
def _(x, grad_out, weight, bias, save_mean, save_invstd, relu):
    return torch.empty_like(x), x.new_empty(x.shape[1]), x.new_empty(x.shape[1])


def _bn_setup(ctx, inputs, output):
    x, weight, bias, _, relu, residual, _parts, _pivot = inputs
    ctx.save_for_backward(x, weight, bias, output[1], output[2])
    ctx.relu = relu
    ctx.has_residual = residual is not None


def _bn_bwd(ctx, g_out, g_mean, g_invstd):
    x, weight, bias, mean, invstd = ctx.saved_tensors
    gx, gw, gb = bn3d_relu_backward(x, g_out.contiguous(), weight, bias, mean, invstd, ctx.relu)
    # the residual is added after the activation: its gradient is grad_out itself
    return gx, (gw if weight is not None else None), (gb if bias is not None else None), None, None, (g_out if ctx.has_residual else None), None, None


bn3d_relu_train.register_autograd(_bn_bwd, setup_context=_bn_setup)


# ------------------------------------------------------------------------------------------- --amp (tools/train.py:24-28)
# Under torch.autocast the 2-D backbone hands out float16 / bfloat16 maps.  The hot path computes in float32 -- the rule torch's own
# autocast applies to `grid_sampler`, `softmax` and the reductions the reference runs here -- so these operators CAST low-precision
# floating inputs to float32 while autocast is on (and return float32), instead of raising the TypeError a non-float32 tensor gets
# outside autocast.  The fp16-STORAGE sweep (pack_features on float16 maps, plane_sweep_variance_shard(half_out=True)) is an explicit
# opt-in and keeps its dtypes: it has no autocast rule.
AUTOCAST_FP32_OPS = (homo_warp, plane_sweep_variance, plane_sweep_variance_keep, depth_prob_topk, sample_depth_prob,
                     backproject_weigh, backproject_weigh_mean)
for _op in AUTOCAST_FP32_OPS:
    _op.register_autocast("cuda", torch.float32)
del _op

