"""Minimal reader for the reference's mmengine-style python configs, for machines without mmengine
("config reader for the `model` dict + extract_feat-equivalent orchestrator").

Only what the hot path needs is interpreted: the file is executed as python, `_base_` files are merged underneath
it the way mmengine does (dicts merge recursively, everything else is replaced, `_delete_=True` drops the
inherited dict), and `hotpath_from_config` reads the five `model` keys the path depends on
(mvsdet.py:165-230: n_voxels, voxel_size, near_far_range, gs_cfg.num_monocular_samples, topk).

    cfg = load_config("projects/NeRF-Det/configs/mvsdet_res50_2x_low_res.py")
    hp  = hotpath_from_config(cfg)            # MVSDetHotPath(n_voxels=[40,40,16], ..., num_monocular_samples=12)
"""
from __future__ import annotations

import os
import types
from typing import Any, Callable, Dict, Optional

_RESERVED = ("_base_", "_delete_")


def _merge(base: Dict[str, Any], child: Dict[str, Any]) -> Dict[str, Any]:
    out = dict(base)
    for key, val in child.items():
        if key == "_delete_":
            continue
        if isinstance(val, dict) and isinstance(out.get(key), dict) and not val.get("_delete_", False):
            out[key] = _merge(out[key], val)
        elif isinstance(val, dict):
            out[key] = _merge({}, val)
        else:
            out[key] = val
    return out


def load_config(path: str, missing_base_ok: bool = True, _seen: Optional[tuple] = None) -> Dict[str, Any]:
    """Execute a python config and fold its `_base_` chain underneath it.

    `missing_base_ok`: the shipped configs inherit `configs/_base_/default_runtime.py`, which carries runtime
    hooks only (nothing the hot path reads); a base file that is absent is skipped unless this is False."""
    path = os.path.abspath(path)
    _seen = _seen or ()
    if path in _seen:
        raise ValueError(f"circular _base_ chain through {path}")
    with open(path, "r") as fh:
        source = fh.read()
    scope: Dict[str, Any] = {"__file__": path}
    exec(compile(source, path, "exec"), scope)  # configs are python by design (mmengine executes them too)
    own = {k: v for k, v in scope.items()
           if not k.startswith("__") and not isinstance(v, (types.ModuleType, types.FunctionType, type))}
    bases = own.pop("_base_", [])
    if isinstance(bases, str):
        bases = [bases]
    merged: Dict[str, Any] = {}
    for rel in bases:
        base_path = os.path.normpath(os.path.join(os.path.dirname(path), rel))
        if not os.path.exists(base_path):
            if missing_base_ok:
                continue
            raise FileNotFoundError(f"{path}: _base_ file {base_path} not found")
        merged = _merge(merged, load_config(base_path, missing_base_ok, _seen + (path,)))
    return _merge(merged, own)


def hotpath_kwargs(cfg: Dict[str, Any]) -> Dict[str, Any]:
    """The constructor arguments of `MVSDetHotPath` from a loaded config (or from a bare `model` dict)."""
    model = cfg.get("model", cfg)
    if model.get("type", "MVSDet") != "MVSDet":
        raise ValueError(f"model.type is {model.get('type')!r}; this package accelerates 'MVSDet' only")
    missing = [k for k in ("n_voxels", "voxel_size", "near_far_range", "gs_cfg") if model.get(k) is None]
    if missing:
        raise KeyError(f"model config lacks {missing} (mvsdet.py:125-155 requires them)")
    return dict(n_voxels=list(model["n_voxels"]), voxel_size=list(model["voxel_size"]),
                near_far_range=list(model["near_far_range"]),
                num_monocular_samples=int(model["gs_cfg"]["num_monocular_samples"]),
                topk=int(model.get("topk", 3)))


def hotpath_from_config(cfg, cost_regularization: Optional[Callable] = None):
    """Build the orchestrator of a1..a10 from a config path, a loaded config or a `model` dict."""
    from .hotpath import MVSDetHotPath
    if isinstance(cfg, (str, os.PathLike)):
        cfg = load_config(os.fspath(cfg))
    return MVSDetHotPath(cost_regularization=cost_regularization, **hotpath_kwargs(cfg))
