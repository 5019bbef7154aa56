"""The convolutions of the detection head that consume the neck's outputs ("neck + head"):
`NerfDetHead._init_layers / _forward_single / forward` of projects/NeRF-Det/nerfdet/nerfdet_head.py:90-118, and the 7-DoF
`ImVoxelHead_ARKit` (:663-700: the same three layers with n_reg_outs = 7; `arkit_head=True`).  Per neck level:

    centerness = conv_center(x)               Conv3d(C -> 1,         k=3, p=1, no bias)
    bbox       = exp(scale_l(conv_reg(x)))    Conv3d(C -> n_reg_outs, k=3, p=1, no bias), one learnable scalar per level;
                                              ImVoxelHead_ARKit: exp(scale_l(.)) on the 6 distances, the angle channel raw
    cls        = conv_cls(x)                  Conv3d(C -> n_classes,  k=3, p=1, bias)

Target assignment, the losses and NMS (nerfdet_head.py:120 ff.) are detection logic outside the path and stay the
reference's.  Parameter names equal the reference's (`conv_center.weight`, `conv_reg.weight`, `conv_cls.weight/bias`,
`scales.<l>.scale`), so a checkpoint's `bbox_head.*` entries load.

1 + 6 + 18 = 25 output channels are no GEMM shape for a library (nine launches per scene).  In eval mode without
autograd, on a ROCm device, the three convolutions of a level run as ONE 3x3x3 MFMA convolution whose 64 output channels
are [center | reg | cls | zeros] -- the input is read once, the neck's small grids take the split over input channels of
csrc/costreg_conv0.hip -- followed by the slices, the exponential and the bias: 5 GFLOP per scene, 13 GFLOP padded.
"""
from __future__ import annotations

from typing import List, Sequence, Tuple

import torch
from torch import Tensor, nn

from .neck import DerivedTensorsMixin, _await_made, _mark_made, fp32_under_autocast


class Scale(nn.Module):
    """mmcv.cnn.Scale: a learnable scalar factor (parameter name `scale`)."""

    def __init__(self, scale: float = 1.0):
        super().__init__()
        self.scale = nn.Parameter(torch.tensor(scale, dtype=torch.float))

    def forward(self, x: Tensor) -> Tensor:
        return x * self.scale


# The fused 128 -> 25 (padded to 64) convolution of a level on the bf16 matrix cores with three-term split operands
# (outputs within ~1e-5 of the fp32 sums' scale; the small levels split over the input channels) instead of the fp32 MFMA.
HEAD_BF16X3 = True


class NerfDetHeadConvs(DerivedTensorsMixin, nn.Module):
    """The learnable layers of NerfDetHead and their forward pass (nerfdet_head.py:94-118)."""

    def __init__(self, n_classes: int = 18, n_levels: int = 3, n_channels: int = 128, n_reg_outs: int = 6,
                 arkit_head: bool = False):
        super().__init__()
        self.n_classes, self.n_levels, self.n_reg_outs = n_classes, n_levels, n_reg_outs
        self.arkit_head = bool(arkit_head)   # ImVoxelHead_ARKit._forward_single (nerfdet_head.py:677-692)
        self.conv_center = nn.Conv3d(n_channels, 1, 3, padding=1, bias=False)
        self.conv_reg = nn.Conv3d(n_channels, n_reg_outs, 3, padding=1, bias=False)
        self.conv_cls = nn.Conv3d(n_channels, n_classes, 3, padding=1)
        self.scales = nn.ModuleList([Scale(1.0) for _ in range(n_levels)])
        self._fused = None   # (key, permuted fused weight); dropped on train()/eval() and load_state_dict
        self._init_derived_hooks()

    def init_weights(self):
        """nerfdet_head.py:104-108: normal_init(std=0.01), classification bias for a prior probability of 0.01."""
        for conv in (self.conv_center, self.conv_reg, self.conv_cls):
            nn.init.normal_(conv.weight, 0.0, 0.01)
        nn.init.constant_(self.conv_cls.bias, float(-torch.log(torch.tensor((1 - 0.01) / 0.01))))

    def _fused_weight(self) -> Tensor:
        from . import ops
        ws = (self.conv_center.weight, self.conv_reg.weight, self.conv_cls.weight)
        key = tuple((w.data_ptr(), w._version, w.device) for w in ws)
        if self._fused is None or self._fused[0] != key or self._fused[2] != HEAD_BF16X3:
            w = torch.cat([t.detach() for t in ws], 0)
            pad = (-w.shape[0]) % 64
            if pad:
                w = torch.cat([w, w.new_zeros((pad,) + tuple(w.shape[1:]))], 0)
            # cut into bf16 pieces in the bf16x3 kernel's layout (csrc/costreg_bf16.hip; HEAD_BF16X3), or permuted for the fp32 MFMA
            self._fused = (key, ops.split_conv_weight(w) if HEAD_BF16X3 else ops.permute_conv_weight(w), HEAD_BF16X3)
            _mark_made(self._fused[1])
        _await_made(self._fused[1])   # computed on another stream a moment ago: this stream waits for it (neck._PENDING)
        return self._fused[1]

    def _forward_single(self, x: Tensor, scale: Scale) -> Tuple[Tensor, Tensor, Tensor]:
        if x.is_cuda and x.dtype == torch.float32 and not torch.is_grad_enabled() and not self.training:
            from . import ops
            if HEAD_BF16X3:
                y = ops.conv3d_k3_bf16x3(x, self._fused_weight(), None, None, False)
            else:
                y = ops.conv3d_k3_mfma(x, self._fused_weight(), None, None, False)
            r, c = self.n_reg_outs, self.n_classes
            center = y[:, :1].contiguous()
            cls = y[:, 1 + r:1 + r + c] + self.conv_cls.bias.detach().view(1, -1, 1, 1, 1)
            if self.arkit_head:
                reg = torch.cat((torch.exp(y[:, 1:7] * scale.scale.detach()), y[:, 7:1 + r]), dim=1)
            else:
                reg = torch.exp(y[:, 1:1 + r] * scale.scale.detach())
            return center, reg, cls
        if self.arkit_head:
            reg_final = self.conv_reg(x)
            return self.conv_center(x), torch.cat((torch.exp(scale(reg_final[:, :6])), reg_final[:, 6:]), dim=1), self.conv_cls(x)
        return self.conv_center(x), torch.exp(scale(self.conv_reg(x))), self.conv_cls(x)

    @fp32_under_autocast
    def forward(self, x: Sequence[Tensor]) -> Tuple[List[Tensor], List[Tensor], List[Tensor]]:
        """mmdet's multi_apply(self._forward_single, x, self.scales): a tuple of three per-level lists."""
        res = [self._forward_single(xi, s) for xi, s in zip(x, self.scales)]
        return tuple(map(list, zip(*res)))

    @staticmethod
    def flops(grid: Sequence[int], n_classes: int = 18, n_levels: int = 3, n_channels: int = 128, n_reg_outs: int = 6) -> float:
        v = sum((grid[0] >> i) * (grid[1] >> i) * (grid[2] >> i) for i in range(n_levels))
        return 54.0 * n_channels * (1 + n_reg_outs + n_classes) * v
