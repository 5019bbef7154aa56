"""Import the reference MVSDet hot-path functions in THIS container (no GPU).

Only used by ``make_goldens.py`` (and by the optional ``-m refcheck`` tests) to
produce the golden vectors under ``tests/golden/*.npz``.  It never travels to
the GPU box in a usable form: ``/root/reference`` does not exist there, and
``load_reference()`` raises ``FileNotFoundError`` when it is missing.

Recipe: the reference module
``projects/NeRF-Det/nerfdet/mvsdet.py`` has top-level imports of packages that
are not installed here (mmdet3d, mmengine, torch_scatter, jaxtyping, gs_src).
None of them is touched by the hot-path functions, so they are satisfied with
empty stand-in modules registered in ``sys.modules``; the real
``mvs_models/module.py`` / ``mvsnet.py`` are loaded from where they lie.
"""
import importlib
import importlib.util
import os
import sys
import types

REF_ROOT = os.environ.get("MVSDET_REFERENCE", "/root/reference")
_NERFDET = os.path.join(REF_ROOT, "projects", "NeRF-Det", "nerfdet")


class _Anything:
    """Placeholder for any class/function imported from an absent package."""

    def __init__(self, *a, **k):
        pass

    def __call__(self, *a, **k):
        return _Anything()

    def __class_getitem__(cls, item):
        return cls

    def __getattr__(self, name):
        if name.startswith("__"):
            raise AttributeError(name)
        return _Anything()

    @staticmethod
    def register_module(*a, **k):
        def deco(obj):
            return obj
        return deco


class _StubModule(types.ModuleType):
    def __getattr__(self, name):
        if name.startswith("__"):
            raise AttributeError(name)
        return _Anything


def _stub(name):
    parts = name.split(".")
    for i in range(1, len(parts) + 1):
        sub = ".".join(parts[:i])
        if sub not in sys.modules:
            m = _StubModule(sub)
            m.__path__ = []  # behave like a package
            sys.modules[sub] = m
            if i > 1:
                setattr(sys.modules[".".join(parts[:i - 1])], parts[i - 1], m)
    return sys.modules[name]


def load_reference():
    """Returns (mvsdet_module, module_module) of the reference."""
    if not os.path.isdir(_NERFDET):
        raise FileNotFoundError(f"reference tree not found at {_NERFDET}")
    if "refpkg.mvsdet" in sys.modules:
        return sys.modules["refpkg.mvsdet"], sys.modules["refpkg.mvs_models.module"]

    import torch.nn as nn

    for name in [
        "mmdet3d", "mmdet3d.models", "mmdet3d.models.detectors", "mmdet3d.registry",
        "mmdet3d.structures", "mmdet3d.structures.det3d_data_sample", "mmdet3d.utils",
        "mmengine", "mmengine.logging", "torch_scatter", "jaxtyping",
        "gs_src", "gs_src.model", "gs_src.model.encoder", "gs_src.model.encoder.epipolar",
        "gs_src.model.encoder.epipolar.depth_predictor_monocular",
        "gs_src.geometry", "gs_src.geometry.projection",
        "gs_src.model.encoder.common", "gs_src.model.encoder.common.gaussian_adapter",
        "gs_src.model.decoder", "gs_src.model.types",
    ]:
        _stub(name)
    sys.modules["mmdet3d.models.detectors"].Base3DDetector = nn.Module
    reg = _Anything()
    sys.modules["mmdet3d.registry"].MODELS = reg
    sys.modules["mmdet3d.registry"].TASK_UTILS = reg

    # synthetic parent package; sub-packages point at the real directories
    pkg = types.ModuleType("refpkg")
    pkg.__path__ = []
    sys.modules["refpkg"] = pkg
    mvs = types.ModuleType("refpkg.mvs_models")
    mvs.__path__ = [os.path.join(_NERFDET, "mvs_models")]
    sys.modules["refpkg.mvs_models"] = mvs
    for name in ["nerf_utils", "nerf_utils.nerf_mlp", "nerf_utils.projection",
                 "nerf_utils.render_ray", "nerf_utils.save_rendered_img"]:
        _stub("refpkg." + name)

    def _load(modname, path):
        spec = importlib.util.spec_from_file_location(modname, path)
        mod = importlib.util.module_from_spec(spec)
        sys.modules[modname] = mod
        spec.loader.exec_module(mod)
        return mod

    module = _load("refpkg.mvs_models.module", os.path.join(_NERFDET, "mvs_models", "module.py"))
    _load("refpkg.mvs_models.mvsnet", os.path.join(_NERFDET, "mvs_models", "mvsnet.py"))
    _load("refpkg.mvs_models.homography", os.path.join(_NERFDET, "mvs_models", "homography.py"))
    mvsdet = _load("refpkg.mvsdet", os.path.join(_NERFDET, "mvsdet.py"))
    return mvsdet, module
