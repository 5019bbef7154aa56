"""ctypes binding of libmvsdet_hip.so (include/mvsdet_hip.h) -- the stub a maintainer of the reference
would add to call the HIP path from Python.

There is NO fallback: if the library is missing or a call fails, a RuntimeError is raised.
"""
from __future__ import annotations

import ctypes
import os
import subprocess

_HERE = os.path.dirname(os.path.abspath(__file__))
# MVSDET_HIP_LIB: load another build of the same C ABI (kernel A/B runs); the default is the in-tree library
LIB_PATH = os.environ.get("MVSDET_HIP_LIB") or os.path.join(_HERE, "libmvsdet_hip.so")
_lib = None

_vp = ctypes.c_void_p
_i = ctypes.c_int
_f = ctypes.c_float
_sz = ctypes.c_size_t
_i64p = ctypes.POINTER(ctypes.c_int64)

# name -> argtypes (all return int unless listed in _RESTYPE); mirrors include/mvsdet_hip.h one to one
SIGNATURES = {
    "mvsdet_version": [],
    "mvsdet_last_error": [],
    "mvsdet_set_option": [ctypes.c_char_p, _i],
    "mvsdet_get_option": [ctypes.c_char_p, ctypes.POINTER(ctypes.c_int)],
    "mvsdet_validate_neighbors": [_i64p, _i, _i, _i],
    "mvsdet_packed_bytes": [_i, _i, _i, _i],
    "mvsdet_pack_features_f32": [_vp, _i64p, _vp, _i, _i, _i, _i, _vp],
    "mvsdet_pack_features_f16": [_vp, _i64p, _vp, _i, _i, _i, _i, _vp],
    "mvsdet_homo_warp_f32": [_vp, _vp, _vp, _vp, _i, _i, _i, _i, _i, _vp],
    "mvsdet_plane_sweep_scratch_bytes": [_i, _i, _i, _i, _i],
    "mvsdet_plane_sweep_workspace_bytes": [_i, _i, _i, _i, _i, _i],
    "mvsdet_plane_sweep_variance_packed_f32": [_vp, _vp, _vp, _vp, _vp, _vp, _sz, _i, _i, _i, _i, _i, _i, _vp],
    "mvsdet_plane_sweep_variance_shard_f32": [_vp, _vp, _vp, _vp, _vp, _vp, _sz, _i, _i, _i, _i, _i, _i, _i, _i, _vp],
    "mvsdet_plane_sweep_variance_shard_f16": [_vp, _vp, _vp, _vp, _vp, _vp, _sz, _i, _i, _i, _i, _i, _i, _i, _i, _vp],
    "mvsdet_plane_sweep_table_f32": [_vp, _vp, _vp, _sz, _i, _i, _i, _i, _i, _vp],
    "mvsdet_plane_sweep_variance_tabled_f32": [_vp, _vp, _vp, _sz, _vp, _i, _i, _i, _i, _i, _i, _vp],
    "mvsdet_plane_sweep_variance_f32": [_vp, _vp, _vp, _vp, _vp, _vp, _sz, _i, _i, _i, _i, _i, _i, _vp],
    "mvsdet_plane_sweep_bwd_workspace_bytes": [_i, _i, _i, _i, _i, _i],
    "mvsdet_plane_sweep_variance_bwd_f32": [_vp, _vp, _vp, _vp, _vp, _vp, _vp, _sz, _i, _i, _i, _i, _i, _i, _vp],
    "mvsdet_plane_sweep_variance_bwd_packed_f32": [_vp, _vp, _vp, _sz, _vp, _vp, _vp, _sz, _i, _i, _i, _i, _i, _i, _vp],
    "mvsdet_depth_prob_topk_f32": [_vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _i, _i, _i, _i, _i, _f, _f, _vp],
    "mvsdet_depth_prob_topk_strided_f32": [_vp, _vp, ctypes.c_longlong, _vp, _vp, _vp, _vp, _vp, _vp, _i, _i, _i, _i, _i, _f, _f, _vp],
    "mvsdet_sample_depth_prob_f32": [_vp, _vp, _vp, _vp, _vp, _vp, _i, _i, _i, _i, _i, _f, _f, _vp],
    "mvsdet_ray_depth_f32": [_vp, _vp, _vp, _vp, _i, _i, _i, _i, _i, _i, _vp],
    "mvsdet_depth_prob_topk_bwd_f32": [_vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _i, _i, _i, _i, _i, _f, _f, _vp],
    "mvsdet_conv3d_k3_mfma_f32": [_vp, _vp, _vp, _vp, _vp, _i, _i, _i, _i, _i, _i, _i, _vp],
    "mvsdet_conv3d_k3_res_mfma_f32": [_vp, _vp, _vp, _vp, _vp, _vp, _i, _i, _i, _i, _i, _i, _i, _vp],
    "mvsdet_convT3d_k3_s2_mfma_f32": [_vp, _vp, _vp, _vp, _vp, _vp, _i, _i, _i, _i, _i, _i, _i, _vp],
    "mvsdet_conv3d_k3_s2_mfma_f32": [_vp, _vp, _vp, _vp, _vp, _i, _i, _i, _i, _i, _i, _i, _vp],
    "mvsdet_plane_sweep_tile_shape": [_i, _i, _i, _i, _vp, _vp, _vp],
    "mvsdet_conv3d_k3_mfma_workspace_bytes": [_i, _i, _i, _i, _i, _i, _i],
    "mvsdet_bn3d_workspace_bytes": [_i],
    "mvsdet_bn3d_relu_train_fwd_f32": [_vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _sz, _i, _i, ctypes.c_longlong, _f, _f, _i, _vp],
    "mvsdet_bn3d_relu_train_fwd_res_f32": [_vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _sz, _i, _i, ctypes.c_longlong, _f, _f, _i, _vp],
    "mvsdet_bn3d_relu_train_fwd_parts_f32": [_vp, _vp, _sz, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _sz, _i, _i, ctypes.c_longlong, _f, _f, _i, _vp],
    "mvsdet_bn3d_relu_bwd_f32": [_vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _sz, _i, _i, ctypes.c_longlong, _i, _vp],
    "mvsdet_conv3d_k3_mfma_ws_f32": [_vp, _vp, _vp, _vp, _vp, _vp, _vp, _sz, _i, _i, _i, _i, _i, _i, _i, _i, _vp],
    "mvsdet_conv3d_k3_dw_partial_bytes": [_i, _i, _i],
    "mvsdet_conv3d_k3_dw_mfma_f32": [_vp, _vp, _vp, _sz, _i, _i, _i, _i, _i, _i, _i, _vp],
    "mvsdet_conv3d_k3_dw_bf16x3": [_vp, _vp, _vp, _sz, _i, _i, _i, _i, _i, _i, _i, _vp],
    "mvsdet_conv3d_k3_s2_dw_mfma_f32": [_vp, _vp, _vp, _sz, _i, _i, _i, _i, _i, _i, _i, _vp],
    "mvsdet_conv3d_k3_s2_dw_bf16x3": [_vp, _vp, _vp, _sz, _i, _i, _i, _i, _i, _i, _i, _vp],
    "mvsdet_conv3d_k3_cout2_dx_f32": [_vp, _vp, _vp, _i, _i, _i, _i, _i, _vp],
    "mvsdet_conv3d_k3_cout2_dw_f32": [_vp, _vp, _vp, _sz, _i, _i, _i, _i, _i, _i, _vp],
    "mvsdet_conv3d_k3_cout2_dw_bf16x3": [_vp, _vp, _vp, _sz, _i, _i, _i, _i, _i, _i, _vp],
    "mvsdet_conv3d_k3_cout2_dw_bf16x3_ok": [_i, _i],
    "mvsdet_gemm_split_weight_bytes": [_i, _i],
    "mvsdet_gemm_split_weight": [_vp, _vp, _i, _i, _vp],
    "mvsdet_conv3d_k1_s2_bf16x3": [_vp, _vp, _vp, _vp, _i, _i, _i, _i, _i, _i, _i, _vp],
    "mvsdet_convT3d_k2_s2_bf16x3": [_vp, _vp, _vp, _vp, _i, _i, _i, _i, _i, _i, _i, _vp],
    "mvsdet_conv3d_k3_cout2_f32": [_vp, _vp, _vp, _vp, _i, _i, _i, _i, _i, _vp],
    "mvsdet_conv3d_k3_cout2_sum_f32": [_vp, _vp, _vp, _vp, _vp, _i, _i, _i, _i, _i, _vp],
    "mvsdet_backproject_weigh_f32": [_vp, _i64p, _vp, _vp, _vp, _vp, _i64p, _vp, _vp, _vp, _vp,
                                     _i, _i, _i, _i, _i, _i, _f, _vp],
    "mvsdet_backproject_weigh_mean_packed_f32": [_vp, _vp, _vp, _vp, _vp, _i64p, _vp, _vp,
                                                 _i, _i, _i, _i, _i, _i, _i, _i, _f, _vp],
    "mvsdet_backproject_weigh_sum_packed_f32": [_vp, _vp, _vp, _vp, _vp, _i64p, _vp, _vp,
                                                _i, _i, _i, _i, _i, _i, _i, _i, _f, _vp],
    "mvsdet_backproject_weigh_bwd_f32": [_vp, _i64p, _vp, _vp, _vp, _vp, _i64p, _vp, _vp, _vp,
                                         _i, _i, _i, _i, _i, _i, _f, _vp],
    "mvsdet_backproject_weigh_mean_bwd_f32": [_vp, _i64p, _vp, _vp, _vp, _vp, _i64p, _vp, _vp, _vp, _vp,
                                              _i, _i, _i, _i, _i, _i, _f, _vp],
    "mvsdet_copy_f32": [_vp, _vp, _sz, _vp],
    "mvsdet_store_pattern_probe_f32": [_vp, _i, _i, _i, _i, _i, _i, _i, _i, _vp],
    "mvsdet_store_pattern_probe_f16": [_vp, _i, _i, _i, _i, _i, _i, _i, _i, _vp],
    "mvsdet_scl_bytes": [_i, _i, _i, _i, _i, _vp, _vp, _vp],
    "mvsdet_pscl_bytes": [_i, _i, _i, _i, _i, _vp, _vp, _vp],
    "mvsdet_conv3d_k3_bf16x3_io": [_vp, _vp, _i64p, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _sz, _i, _i, _i, _i, _i, _i, _i, _vp],
    "mvsdet_conv3d_k3_bf16x3_stats_parts": [_i, _i, _i, _i, _i],
    "mvsdet_conv3d_k3_bf16x3_stats": [_vp, _vp, _i64p, _vp, _vp, _vp, _sz, _vp, _i, _i, _i, _i, _i, _i, _vp],
    "mvsdet_convT3d_k3_s2_bf16x3_stats_parts": [_i, _i, _i, _i],
    "mvsdet_convT3d_k3_s2_bf16x3_stats": [_vp, _vp, _vp, _vp, _sz, _vp, _i, _i, _i, _i, _i, _i, _vp],
    "mvsdet_conv3d_k3_s2_bf16x3_io": [_vp, _i64p, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _sz, _i, _i, _i, _i, _i, _i, _i, _vp],
    "mvsdet_convT3d_k3_s2_bf16x3_io": [_vp, _vp, _vp, _vp, _vp, _vp, _vp, _i, _i, _i, _i, _i, _i, _i, _vp],
    "mvsdet_split_conv_weight_bytes": [_i, _i],
    "mvsdet_split_conv_weight": [_vp, _vp, _i, _i, _vp],
    "mvsdet_split_conv_weight_ordered": [_vp, _vp, _i, _i, _i, _vp],
    "mvsdet_split_conv_weights_batched": [_vp, _vp, _vp, _vp, _vp, _i, _vp],
    "mvsdet_convT3d_k3_s2_bf16x3": [_vp, _vp, _vp, _vp, _vp, _vp, _i, _i, _i, _i, _i, _i, _i, _vp],
    "mvsdet_conv3d_k3_s2_bf16x3_f32in": [_vp, _i64p, _vp, _vp, _vp, _vp, _i, _i, _i, _i, _i, _i, _i, _vp],
    "mvsdet_scl_pack_f32": [_vp, _i64p, _vp, _i, _i, _i, _i, _i, _i, _vp],
    "mvsdet_plane_sweep_table_pitched_f32": [_vp, _vp, _vp, _sz, _i, _i, _i, _i, _i, _i, _vp],
    "mvsdet_plane_sweep_variance_tabled_pitched_f32": [_vp, _vp, _vp, _sz, _vp, _i, _i, _i, _i, _i, _i, _i, _vp],
    "mvsdet_conv3d_k3_bf16x3": [_vp, _vp, _vp, _vp, _vp, _vp, _i, _i, _i, _i, _i, _i, _i, _vp],
    "mvsdet_conv3d_k3_bf16x3_f32in": [_vp, _i64p, _vp, _vp, _vp, _vp, _vp, _i, _i, _i, _i, _i, _i, _i, _vp],
    "mvsdet_conv3d_k3_bf16x3_workspace_bytes": [_i, _i, _i, _i, _i, _i],
    "mvsdet_conv3d_k3_s2_bf16x3_workspace_bytes": [_i, _i, _i, _i, _i, _i],
    "mvsdet_conv3d_k3_s2_bf16x3_f32in_ws": [_vp, _i64p, _vp, _vp, _vp, _vp, _vp, _sz, _i, _i, _i, _i, _i, _i, _i, _vp],
    "mvsdet_conv3d_k3_bf16x3_ws": [_vp, _vp, _vp, _vp, _vp, _vp, _vp, _sz, _i, _i, _i, _i, _i, _i, _i, _vp],
    "mvsdet_conv3d_k3_bf16x3_f32in_ws": [_vp, _i64p, _vp, _vp, _vp, _vp, _vp, _vp, _sz, _i, _i, _i, _i, _i, _i, _i, _vp],
    "mvsdet_split_conv_weight_mx_bytes": [_i, _i],
    "mvsdet_split_conv_weight_mx": [_vp, _vp, _i, _i, _vp],
    "mvsdet_conv3d_k3_fp16mx_f32in": [_vp, _i64p, _vp, _vp, _vp, _vp, _vp, _vp, _i, _i, _i, _i, _i, _i, _i, _vp],
}
_RESTYPE = {"mvsdet_last_error": ctypes.c_char_p, "mvsdet_conv3d_k3_bf16x3_stats_parts": ctypes.c_size_t, "mvsdet_convT3d_k3_s2_bf16x3_stats_parts": ctypes.c_size_t, "mvsdet_gemm_split_weight_bytes": ctypes.c_size_t, "mvsdet_scl_bytes": ctypes.c_size_t, "mvsdet_pscl_bytes": ctypes.c_size_t,
            "mvsdet_split_conv_weight_bytes": ctypes.c_size_t, "mvsdet_packed_bytes": ctypes.c_size_t,
            "mvsdet_split_conv_weight_mx_bytes": ctypes.c_size_t,
            "mvsdet_plane_sweep_scratch_bytes": ctypes.c_size_t, "mvsdet_plane_sweep_workspace_bytes": ctypes.c_size_t,
            "mvsdet_plane_sweep_bwd_workspace_bytes": ctypes.c_size_t,
            "mvsdet_conv3d_k3_dw_partial_bytes": ctypes.c_size_t,
            "mvsdet_conv3d_k3_bf16x3_workspace_bytes": ctypes.c_size_t,
            "mvsdet_conv3d_k3_s2_bf16x3_workspace_bytes": ctypes.c_size_t,
            "mvsdet_conv3d_k3_mfma_workspace_bytes": ctypes.c_size_t, "mvsdet_bn3d_workspace_bytes": ctypes.c_size_t}


def build(verbose: bool = False) -> str:
    """hipcc --offload-arch=gfx950 every kernel into mvsdet_amd/libmvsdet_hip.so (cross-compiles without a GPU)."""
    cmd = ["make", "-C", os.path.join(_HERE, "csrc"), "-j4"]
    subprocess.check_call(cmd, stdout=None if verbose else subprocess.DEVNULL)
    return LIB_PATH


def load():
    """Load the HIP library; raise loudly if it is not there."""
    global _lib
    if _lib is not None:
        return _lib
    if not os.path.exists(LIB_PATH):
        raise RuntimeError(
            f"{LIB_PATH} not found: the HIP extension is not built. Run `python -c 'import __graft_entry__ as g; "
            "g.build()'` (or `make -C mvsdet_amd/csrc`). mvsdet_amd has no CPU fallback.")
    lib = ctypes.CDLL(LIB_PATH)
    for name, argtypes in SIGNATURES.items():
        fn = getattr(lib, name)  # AttributeError if the library does not export a declared symbol
        fn.argtypes = argtypes
        fn.restype = _RESTYPE.get(name, ctypes.c_int)
    _lib = lib
    return lib


def set_option(name: str, value: int):
    """Schedule-only tuning option of the library (include/mvsdet_hip.h: mvsdet_set_option)."""
    check(load().mvsdet_set_option(name.encode(), int(value)), f"set_option({name})")


def get_option(name: str) -> int:
    v = ctypes.c_int(0)
    check(load().mvsdet_get_option(name.encode(), ctypes.byref(v)), f"get_option({name})")
    return int(v.value)


def check(rc: int, what: str):
    if rc != 0:
        msg = load().mvsdet_last_error().decode("utf-8", "replace")
        kinds = {1: ValueError, 2: RuntimeError, 3: RuntimeError}
        raise kinds.get(rc, RuntimeError)(f"{what} failed (code {rc}): {msg}")


def strides4(t) -> ctypes.Array:
    """HOST int64[4] with the element strides of a 4-D tensor."""
    assert t.dim() == 4
    return (ctypes.c_int64 * 4)(*[int(s) for s in t.stride()])


def ptr(t):
    """Device pointer of a tensor (None -> NULL)."""
    return None if t is None else ctypes.c_void_p(t.data_ptr())


def current_stream(device) -> ctypes.c_void_p:
    import torch
    return ctypes.c_void_p(torch.cuda.current_stream(device).cuda_stream)


def sweep_tile_shape(K: int, D: int, H: int, W: int):
    """(tile_w, tile_h, box_texels) the sweep uses for this problem (mvsdet_plane_sweep_tile_shape)."""
    tw, th, cap = ctypes.c_int(0), ctypes.c_int(0), ctypes.c_int(0)
    check(load().mvsdet_plane_sweep_tile_shape(K, D, H, W, ctypes.byref(tw), ctypes.byref(th), ctypes.byref(cap)), "sweep_tile_shape")
    return tw.value, th.value, cap.value
