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


# ---------------------------------------------------------------------------------------------------------------
# Row f-3: the 3-D neck and the detection heads' convolutions.  Their files import mmcv / mmdet /
# mmengine at the top, none of which is installed here, so the CLASS TEXT is taken from the reference file where it lies
# and executed with the three helpers it uses spelled out (what each does in the configuration the shipped configs ask
# for); nothing of the reference is stored.
# ---------------------------------------------------------------------------------------------------------------
def load_reference_neck():
    """`IndoorImVoxelNeck` (and `ResModule`) of mmdet3d/models/necks/imvoxel_neck.py:68-231 as executable classes."""
    import textwrap
    from torch import nn
    path = os.path.join(REF_ROOT, "mmdet3d", "models", "necks", "imvoxel_neck.py")
    if not os.path.isfile(path):
        raise FileNotFoundError(path)
    src = open(path).read().splitlines()
    first = next(i for i, l in enumerate(src) if l.startswith("class IndoorImVoxelNeck("))
    block = textwrap.dedent("\n".join(src[first:]))

    class ConvModule(nn.Module):   # mmcv.cnn.ConvModule for conv_cfg Conv3d, norm_cfg BN3d, act_cfg ReLU | None
        def __init__(self, in_channels, out_channels, kernel_size, stride=1, padding=0, conv_cfg=None, norm_cfg=None, act_cfg=None):
            super().__init__()
            assert conv_cfg == dict(type='Conv3d') and norm_cfg == dict(type='BN3d')
            self.conv = nn.Conv3d(in_channels, out_channels, kernel_size, stride=stride, padding=padding, bias=False)
            self.bn = nn.BatchNorm3d(out_channels)
            self.with_activation = act_cfg is not None
            if self.with_activation:
                self.activate = nn.ReLU(inplace=act_cfg.get('inplace', True))

        def forward(self, x):
            x = self.bn(self.conv(x))
            return self.activate(x) if self.with_activation else x

    ns = dict(nn=nn, ConvModule=ConvModule, BaseModule=nn.Module)
    exec(compile(block, path, "exec"), ns)
    return ns["IndoorImVoxelNeck"]


def load_reference_head(cls_name: str):
    """`_init_layers` / `_forward_single` / `forward` of NerfDetHead (nerfdet_head.py:90-118) or ImVoxelHead_ARKit (:663-700)
    as methods of a bare nn.Module: RefHead(n_channels, n_reg_outs, n_classes, n_levels)."""
    import textwrap
    import torch
    from torch import nn, Tensor
    path = os.path.join(_NERFDET, "nerfdet_head.py")
    if not os.path.isfile(path):
        raise FileNotFoundError(path)
    src = open(path).read().splitlines()
    cls_line = next(i for i, l in enumerate(src) if l.startswith(f"class {cls_name}("))
    first = next(i for i in range(cls_line, len(src)) if src[i].startswith("    def _init_layers"))
    last = next(i for i in range(first, len(src)) if src[i].startswith("    def loss("))
    block = textwrap.dedent("\n".join(src[first:last]))

    class Scale(nn.Module):                      # mmcv.cnn.Scale: a learnable scalar factor
        def __init__(self, scale=1.0):
            super().__init__()
            self.scale = nn.Parameter(torch.tensor(scale, dtype=torch.float))

        def forward(self, x):
            return x * self.scale

    def multi_apply(func, *args):                # mmdet.models.utils.multi_apply without its kwargs
        return tuple(map(list, zip(*map(func, *args))))

    ns = dict(nn=nn, torch=torch, Tensor=Tensor, Scale=Scale, multi_apply=multi_apply, normal_init=lambda *a, **k: None,
              bias_init_with_prob=lambda p: 0.0)
    exec(compile(block, path, "exec"), ns)

    class RefHead(nn.Module):
        _init_layers, _forward_single, forward = ns["_init_layers"], ns["_forward_single"], ns["forward"]

        def __init__(self, n_channels, n_reg_outs, n_classes, n_levels):
            super().__init__()
            self._init_layers(n_channels, n_reg_outs, n_classes, n_levels)

    return RefHead
