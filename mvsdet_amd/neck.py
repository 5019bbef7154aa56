"""The 3-D neck that consumes the hot path's voxel volume: `IndoorImVoxelNeck` of
mmdet3d/models/necks/imvoxel_neck.py:70-231 -- a three-level 3-D FPN of residual blocks over the (C,40,40,16) volume,
~0.49 TFLOP per scene with the shipped configuration (in_channels=256, out_channels=128, n_blocks=[1,1,1]).

The module is plain PyTorch with the reference's parameter names (mmcv's ConvModule registers its layers as `conv`
and `bn`, so `down_layer_1.0.conv0.conv.weight`, `down_layer_1.0.downsample.bn.running_mean`, `up_block_2.0.weight`,
`out_block_0.1.bias` ... load from a reference checkpoint's `neck_3d.*` entries with `load_state_dict`).  In eval mode
without autograd, on a ROCm device, every 3x3x3 convolution runs on the bf16x3 MFMA kernels of csrc/costreg_bf16.hip
(three bf16 products per fp32-equivalent product, fp32 accumulation: DESIGN 4.3; `matrix_precision = "fp32"` keeps the
fp32-MFMA kernels of csrc/costreg_conv0.hip) with eval-mode BatchNorm, ReLU and the residual addition folded into the
epilogue; the small levels split their input channels until ~768 blocks run.  The 1x1x1 stride-2 shortcut and the
kernel-2 stride-2 transposed convolutions are one GEMM each on csrc/neck_gemm.hip (a kernel-2 stride-2 transposed
convolution is 8 independent single-tap classes; bias, ReLU and the 2x2x2 interleave in the epilogue; no rocBLAS call).
Training, CPU tensors and other shapes take the framework's layers.
"""
from __future__ import annotations

import functools
import os
import weakref

from typing import List, Sequence

import torch
from torch import Tensor, nn
from torch.nn import functional as F


def fp32_under_autocast(forward):
    """`--amp` (tools/train.py:24-28): under torch.autocast a module of this package still computes in float32 (bf16x3 on the
    matrix cores is fp32-equivalent; the framework's layers it falls back to would otherwise run float16 convolutions beside
    it): low-precision inputs are cast up and autocast is off for the call.  Outside autocast the call is untouched."""
    def _up(v):
        if isinstance(v, Tensor):
            return v.float() if v.is_floating_point() and v.dtype != torch.float32 else v
        if isinstance(v, (list, tuple)):
            return type(v)(_up(t) for t in v)
        return v

    @functools.wraps(forward)
    def wrapped(self, x, *args, **kwargs):
        if torch.is_autocast_enabled("cuda"):
            with torch.autocast("cuda", enabled=False):
                return forward(self, _up(x), *args, **kwargs)
        return forward(self, x, *args, **kwargs)
    return wrapped


class _ConvModule(nn.Module):
    """mmcv.cnn.ConvModule(conv_cfg=Conv3d, norm_cfg=BN3d, act_cfg=ReLU|None) as used by imvoxel_neck.py:196-217:
    Conv3d without bias -> BatchNorm3d [-> ReLU], sub-modules named `conv` and `bn` as mmcv names them."""

    def __init__(self, cin: int, cout: int, kernel: int, stride: int = 1, padding: int = 0, act: bool = True):
        super().__init__()
        self.conv = nn.Conv3d(cin, cout, kernel, stride=stride, padding=padding, bias=False)
        self.bn = nn.BatchNorm3d(cout)
        self.with_act = act

    def forward(self, x):
        x = self.bn(self.conv(x))
        return F.relu(x, inplace=True) if self.with_act else x


_DERIVED = ("_mvs_affine", "_fused", "_mvs_wsplit", "_mvs_wmat", "_mvs_sclbuf")   # tensors computed from parameters and kept on a module


def drop_derived_tensors(root: nn.Module) -> None:
    """Forget every tensor this package derived from `root`'s parameters (BatchNorm affines, fused / permuted / split
    weights).  The caches key on (data_ptr, _version), which an in-place update through `.data` does not bump (mmengine's
    EMAHook swaps parameters that way) -- so they are also dropped on every train()/eval() switch and after
    load_state_dict (`DerivedTensorsMixin`); call this by hand after any other out-of-band `.data` write."""
    for m in root.modules():
        for name in _DERIVED:
            if m.__dict__.get(name) is not None:
                m.__dict__[name] = None


def _drop_after_load(module: nn.Module, incompatible_keys) -> None:
    # module-level (not a lambda or a closure): the hook is stored on the module and must pickle with it
    # (torch.save(model), mp.spawn)
    drop_derived_tensors(module)


class DerivedTensorsMixin:
    """nn.Module mixin of the modules that own derived-tensor caches: mode switches and state-dict loads drop them."""

    def _init_derived_hooks(self):
        self.register_load_state_dict_post_hook(_drop_after_load)

    def train(self, mode: bool = True):
        drop_derived_tensors(self)
        return super().train(mode)


# Derived tensors are computed by whichever stream first needs them and kept on the module; a call on ANOTHER stream shortly
# afterwards (the two halves of CostRegNet3DGS.view_streams; a detector moved to a side stream) must not read them before the
# kernels that fill them ran.  Every derived tensor is therefore registered with the event recorded behind its computation, and every
# use makes the using stream wait for it while it is pending.  Kept outside the modules (events do not pickle), by the tensor.
_PENDING: dict = {}   # id(tensor) -> (weak reference to it, event); by identity: tensors compare element-wise


def _mark_made(*tensors: Tensor) -> None:
    ts = [t for t in tensors if isinstance(t, Tensor) and t.is_cuda]
    if ts:
        ev = torch.cuda.Event()
        ev.record(torch.cuda.current_stream(ts[0].device))
        for t in ts:
            _PENDING[id(t)] = (weakref.ref(t), ev)
        if len(_PENDING) > 4096:   # entries of tensors that died before anyone asked again
            for k in [k for k, (r, e) in _PENDING.items() if r() is None or e.query()]:
                _PENDING.pop(k, None)


def _await_made(*tensors: Tensor) -> None:
    for t in tensors:
        ent = _PENDING.get(id(t)) if isinstance(t, Tensor) else None
        if ent is not None and ent[0]() is t:
            if ent[1].query():
                _PENDING.pop(id(t), None)
            else:
                torch.cuda.current_stream(t.device).wait_event(ent[1])


def _bn_affine(bn: nn.BatchNorm3d):
    """Eval-mode BatchNorm as a per-channel affine; kept on the module until one of its four tensors changes (five tiny
    kernels per layer otherwise: a twentieth of the neck's time at one scene) or `drop_derived_tensors` runs."""
    key = tuple((t.data_ptr(), t._version) for t in (bn.weight, bn.bias, bn.running_mean, bn.running_var))
    cached = getattr(bn, "_mvs_affine", None)
    if cached is None or cached[0] != key:
        with torch.no_grad():
            scale = bn.weight / torch.sqrt(bn.running_var + bn.eps)
            cached = (key, scale, bn.bias - bn.running_mean * scale)
        _mark_made(cached[1], cached[2])
        bn._mvs_affine = cached
    _await_made(cached[1], cached[2])
    return cached[1], cached[2]


def _hip_ok(x: Tensor, module: nn.Module) -> bool:
    return x.is_cuda and x.dtype == torch.float32 and not torch.is_grad_enabled() and not module.training


# Stride-1 3x3x3 convolutions on volumes of at least this many voxels go to the bf16 matrix cores with three-term split
# operands (csrc/costreg_bf16.hip; outputs within ~1e-5 of the fp32 sums' scale).  The 20x20x8 and 10x10x4 levels are a
# handful of tiles (padded 1.9x): there the kernel splits the 512 / 1024 input channels over blocks and a second kernel
# adds the partial sums (neck 4.05 -> 3.10 ms).  0 disables the bf16 route.
BF16X3_MIN_VOXELS = 256
# stride-1 layers with this many output channels or more (four blocks of 64 per tile) on volumes of at least PACK_INPUT_MIN_VOXELS
# voxels (grids that are not split over the input channels) read a packed SCL copy of their input by DMA; 0 = never
PACK_INPUT_FROM_COUT = int(os.environ.get("MVSDET_NECK_PACK_COUT", "256"))
PACK_INPUT_MIN_VOXELS = 16384
# the two stride-2 layers on the bf16x3 stride-2 kernel (its 3x16x8 tiles fit their outputs: neck 2.77 -> 2.58 ms; on 4x8x16 tiles
# it lost to the fp32 kernel, 0.72 against 0.64 ms)
S2_BF16X3 = True


def _split_weight(conv: nn.Module, order: int | None = None) -> Tensor:
    """The weight cut into bf16 pieces in the bf16x3 kernel's layout (tap order of the layer's stride; `order` 2 = a
    ConvTranspose3d weight), kept on the module (the neck's weights are 300 MB: not re-split per call) until the weight tensor
    changes or `drop_derived_tensors` runs."""
    from . import ops
    w = conv.weight
    if order is None:
        order = 1 if conv.stride[0] == 2 else 0
    # the layout of an order-0 split follows the library option `conv_mfma16` (16x16x32 or 32x32x16 row order), and the launcher
    # picks the kernel form from the same option at launch time: the option is part of the key, so weights cut under one setting
    # are never fed to the other form after `mvsdet_set_option` / an A/B run in one process
    key = (w.data_ptr(), w._version, w.device, order, ops.get_option("conv_mfma16") if order == 0 and w.is_cuda else 0)
    cached = conv.__dict__.get("_mvs_wsplit")
    if cached is None or cached[0] != key:
        cached = (key, ops.split_conv_weight(w, order=order))
        _mark_made(cached[1])
        conv.__dict__["_mvs_wsplit"] = cached
    _await_made(cached[1])
    return cached[1]


# the 1x1x1 stride-2 shortcut and the kernel-2 stride-2 transposed layers on our GEMM kernel (csrc/neck_gemm.hip: bf16x3, bias, ReLU
# and the 2x2x2 interleave in the epilogue) where their shapes allow; False: one rocBLAS fp32 GEMM + ATen glue each (rounds 2-4)
GEMM_BF16X3 = os.environ.get("MVSDET_NECK_GEMM", "bf16x3") == "bf16x3"


def _gemm_weight(conv: nn.Module, bn: nn.BatchNorm3d, split: bool = False):
    """The 1x1x1 convolution (Cout x Cin) or the ConvTranspose3d(k=2, s=2) (8*Cout x Cin) as ONE matrix with the eval-mode
    BatchNorm's scale folded in, and the matching bias; kept on the module like the split weights (three tiny kernels per layer
    and call otherwise).  split=False: (wmat fp32 with rows (p, q, r, o), bias column (1, 8 Cout, 1)) for torch.baddbmm;
    split=True: (the bf16 pieces of the matrix with rows 8 o + 4 p + 2 q + r in the GEMM kernel's fragment order, bias (Cout,))."""
    w = conv.weight
    key = (w.data_ptr(), w._version, w.device, bool(split)) + tuple((t.data_ptr(), t._version) for t in (bn.weight, bn.bias, bn.running_mean, bn.running_var))
    cached = conv.__dict__.get("_mvs_wmat")
    if cached is None or cached[0] != key:
        scale, shift = _bn_affine(bn)
        with torch.no_grad():
            if isinstance(conv, nn.ConvTranspose3d):
                cout = conv.out_channels
                ws = w.detach() * scale.view(1, -1, 1, 1, 1)
                if split:
                    wmat, bias = ws.permute(1, 2, 3, 4, 0).reshape(8 * cout, w.shape[0]).contiguous(), shift.contiguous()
                else:
                    wmat, bias = ws.permute(2, 3, 4, 1, 0).reshape(8 * cout, w.shape[0]).contiguous(), shift.repeat(8).view(1, -1, 1).contiguous()
            else:
                wmat = (w.detach().reshape(conv.out_channels, -1) * scale[:, None]).contiguous()
                bias = shift.contiguous() if split else shift.view(1, -1, 1).contiguous()
            if split:
                from . import ops
                wmat = ops.gemm_split_weight(wmat)
        cached = (key, wmat, bias)
        _mark_made(wmat, bias)
        conv.__dict__["_mvs_wmat"] = cached
    _await_made(cached[1], cached[2])
    return cached[1], cached[2]


def _conv_k3(x: Tensor, conv: nn.Conv3d, bn: nn.BatchNorm3d, relu: bool, residual: Tensor | None = None) -> Tensor:
    """Conv3d(k=3, p=1, stride 1|2) + BN(eval) [+ residual] [+ ReLU] in one MFMA kernel."""
    from . import ops
    scale, shift = _bn_affine(bn)
    if conv.stride[0] == 1 and BF16X3_MIN_VOXELS and x[0, 0].numel() >= BF16X3_MIN_VOXELS and conv.out_channels % 64 == 0:
        src = x
        if PACK_INPUT_FROM_COUT and conv.out_channels >= PACK_INPUT_FROM_COUT and x[0, 0].numel() >= PACK_INPUT_MIN_VOXELS:
            # four blocks of output channels per tile would each cut the same fp32 values into bf16 pieces: one packing pass
            # (26 MB at the 40x40x16 level) and the DMA-fed form instead; the copy is this call's own (from the caching allocator,
            # the packing kernel writes its zero border): nothing kept on the module, nothing tied to a stream
            src = ops.scl_pack(x)
        return ops.conv3d_k3_bf16x3(src, _split_weight(conv), scale, shift, relu, residual)
    if S2_BF16X3 and conv.stride[0] == 2 and residual is None and conv.out_channels % 64 == 0:
        return ops.conv3d_k3_s2_bf16x3(x, _split_weight(conv), scale, shift, relu)
    return ops.conv3d_k3_mfma(x, ops.permute_conv_weight(conv.weight), scale, shift, relu, conv.stride[0], residual)


class ResModule(nn.Module):
    """imvoxel_neck.py:183-231."""

    def __init__(self, cin: int, cout: int, stride: int = 1):
        super().__init__()
        self.conv0 = _ConvModule(cin, cout, 3, stride, 1, act=True)
        self.conv1 = _ConvModule(cout, cout, 3, 1, 1, act=False)
        if stride != 1:
            self.downsample = _ConvModule(cin, cout, 1, stride, 0, act=False)
        self.stride = stride

    def forward(self, x):
        if _hip_ok(x, self) and self.conv0.conv.out_channels % 64 == 0:
            identity = x
            if self.stride != 1:   # 1x1x1 stride-2 convolution + BN: one GEMM on the sub-sampled volume
                ds = self.downsample
                from . import ops
                if (GEMM_BF16X3 and self.stride == 2 and ops.gemm_layer_ok(ds.conv.out_channels, ds.conv.in_channels)
                        and not any(v % 2 for v in x.shape[2:])):
                    wq, bias = _gemm_weight(ds.conv, ds.bn, split=True)      # the sub-sampling is the kernel's gather
                    identity = ops.conv3d_k1_s2_bf16x3(x, wq, bias, ds.conv.out_channels)
                else:
                    xs = x[:, :, ::self.stride, ::self.stride, ::self.stride]
                    n, c, d, h, w = xs.shape
                    wmat, bias = _gemm_weight(ds.conv, ds.bn)
                    identity = torch.baddbmm(bias, wmat.unsqueeze(0).expand(n, -1, -1), xs.reshape(n, c, -1))
                    identity = identity.view(n, -1, d, h, w)
            h0 = _conv_k3(x, self.conv0.conv, self.conv0.bn, True)
            return _conv_k3(h0, self.conv1.conv, self.conv1.bn, True, identity)   # relu(bn(conv1) + identity)
        identity = x
        x = self.conv1(self.conv0(x))
        if self.stride != 1:
            identity = self.downsample(identity)
        return F.relu(x + identity, inplace=True)


class _UpBlock(nn.Sequential):
    """imvoxel_neck.py:166-180: ConvTranspose3d(k=2, s=2) BN ReLU Conv3d(k=3) BN ReLU, Sequential indices as the reference's."""

    def __init__(self, cin: int, cout: int):
        super().__init__(nn.ConvTranspose3d(cin, cout, 2, 2, bias=False), nn.BatchNorm3d(cout), nn.ReLU(inplace=True),
                         nn.Conv3d(cout, cout, 3, 1, 1, bias=False), nn.BatchNorm3d(cout), nn.ReLU(inplace=True))

    def forward(self, x):
        if _hip_ok(x, self) and self[3].out_channels % 64 == 0:
            deconv, bn = self[0], self[1]
            n, cin, d, h, w = x.shape
            cout = deconv.out_channels
            # out[:, o, 2i+p, 2j+q, 2k+r] = sum_c x[:, c, i, j, k] * W[c, o, p, q, r]: one (8*Cout x Cin) GEMM with the BatchNorm's
            # shift as its bias; the ReLU writes the interleaved (N, Cout, 2D, 2H, 2W) tensor directly (one pass, no copy)
            from . import ops
            if GEMM_BF16X3 and ops.gemm_layer_ok(8 * cout, cin):
                wq, bias = _gemm_weight(deconv, bn, split=True)      # bias, ReLU and the 2x2x2 interleave in the GEMM's epilogue
                return _conv_k3(ops.convT3d_k2_s2_bf16x3(x, wq, bias, cout, True), self[3], self[4], True)
            wmat, bias = _gemm_weight(deconv, bn)
            y = torch.baddbmm(bias, wmat.unsqueeze(0).expand(n, -1, -1), x.reshape(n, cin, -1)).view(n, 2, 2, 2, cout, d, h, w)
            out = torch.empty((n, cout, 2 * d, 2 * h, 2 * w), dtype=x.dtype, device=x.device)
            torch.clamp_min(y.permute(0, 4, 5, 1, 6, 2, 7, 3), 0.0, out=out.view(n, cout, d, 2, h, 2, w, 2))
            return _conv_k3(out, self[3], self[4], True)
        return super().forward(x)


class _OutBlock(nn.Sequential):
    """imvoxel_neck.py:152-163: Conv3d(k=3) BN ReLU."""

    def __init__(self, cin: int, cout: int):
        super().__init__(nn.Conv3d(cin, cout, 3, 1, 1, bias=False), nn.BatchNorm3d(cout), nn.ReLU(inplace=True))

    def forward(self, x):
        if _hip_ok(x, self) and self[0].out_channels % 64 == 0:
            return _conv_k3(x, self[0], self[1], True)
        return super().forward(x)


class IndoorImVoxelNeck(DerivedTensorsMixin, nn.Module):
    """imvoxel_neck.py:70-131: x (N, C_in, Nx, Ny, Nz) -> list of n_scales tensors (N, C_out, Nx/2^i, Ny/2^i, Nz/2^i)."""

    def __init__(self, in_channels: int, out_channels: int, n_blocks: Sequence[int]):
        super().__init__()
        self._init_derived_hooks()
        self.n_scales = len(n_blocks)
        n_channels = in_channels
        for i, nb in enumerate(n_blocks):
            stride = 1 if i == 0 else 2
            blocks = []
            for b in range(nb):   # imvoxel_neck.py:133-149
                if b == 0 and stride != 1:
                    blocks.append(ResModule(n_channels, n_channels * 2, stride))
                    n_channels = n_channels * 2
                else:
                    blocks.append(ResModule(n_channels, n_channels))
            setattr(self, f"down_layer_{i}", nn.Sequential(*blocks))
            if i > 0:
                setattr(self, f"up_block_{i}", _UpBlock(n_channels, n_channels // 2))
            setattr(self, f"out_block_{i}", _OutBlock(n_channels, out_channels))

    @fp32_under_autocast
    def forward(self, x: Tensor) -> List[Tensor]:
        down_outs = []
        for i in range(self.n_scales):
            x = getattr(self, f"down_layer_{i}")(x)
            down_outs.append(x)
        outs = []
        for i in range(self.n_scales - 1, -1, -1):
            if i < self.n_scales - 1:
                x = getattr(self, f"up_block_{i + 1}")(x)
                x = down_outs[i] + x
            outs.append(getattr(self, f"out_block_{i}")(x))
        return outs[::-1]

    @staticmethod
    def flops(n: int, grid: Sequence[int], in_channels: int = 256, out_channels: int = 128, n_scales: int = 3) -> float:
        """2 x multiply-adds of one forward pass with n_blocks = [1]*n_scales (for the MFMA roofline)."""
        total, c = 0.0, in_channels
        v = [n * (grid[0] >> i) * (grid[1] >> i) * (grid[2] >> i) for i in range(n_scales)]
        for i in range(n_scales):
            if i == 0:
                total += 2 * 54 * c * c * v[0]
            else:
                total += 54 * c * 2 * c * v[i] + 54 * (2 * c) ** 2 * v[i] + 2 * c * 2 * c * v[i]
                c *= 2
            total += 54 * c * out_channels * v[i]
            if i > 0:
                total += 2 * c * (c // 2) * v[i - 1] + 54 * (c // 2) ** 2 * v[i - 1]
        return total
