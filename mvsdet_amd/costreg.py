"""The cost regularisation network that sits between a4 and a5 ("next"):
`CostRegNet_3DGS` of mvs_models/mvsnet.py:73-113 -- a three-level 3-D U-Net, (N,256,D,H,W) variance ->
(N,2,D,H,W) {cost logits, offset logits}.

The module is plain PyTorch; under autograd every convolution (stride 1, stride 2, transposed, head) uses our forward /
input-gradient / weight-gradient kernels and BatchNorm + ReLU the streaming kernels of csrc/costreg_bn.hip (training step
83 ms instead of 858 ms on MIOpen), in eval mode without autograd every layer is
routed to the fp32-MFMA / streaming HIP kernels of csrc/costreg_conv0.hip and
csrc/costreg_head.hip (25 ms per scene instead of 59.3 ms at the reference-true shape, same fp32 sums).  Parameter names and shapes equal the reference's
(`conv0.conv.weight`, `conv0.bn.*`, ..., `conv9.0.weight`, `conv9.1.*`, `conv11.0.weight`, `conv11.1.*`,
`prob.weight/bias`), so a reference checkpoint's `cost_regularization.*` entries load with `load_state_dict`
(tests/test_integration.py compares the outputs with the reference module itself).  D, H, W must be divisible by 4
(two stride-2 levels), as in the reference.

At the reference-true shape the network is ~2.8 TFLOP per scene against ~1 ms for the whole hot path around it:
on the GPU it, not the plane sweep, is what a scene costs.
"""
from __future__ import annotations

import os
import weakref
from typing import Optional

import torch
from torch import nn

from .neck import DerivedTensorsMixin, _bn_affine, fp32_under_autocast
from .scratch import EventPool


class _ConvK3S1(torch.autograd.Function):
    """Conv3d(kernel 3, stride 1, padding 1, no bias) with all three passes on the fp32 matrix cores: forward and input
    gradient through `ops.conv3d_k3_mfma` (the input gradient is the same convolution of grad_out with the weights
    transposed and flipped), weight gradient through `ops.conv3d_k3_dw`.  MIOpen needs 34 + 42 + 360 ms for conv0 at the
    reference-true shape, these kernels 15 + 15 + 22 ms."""

    @staticmethod
    def forward(ctx, x, weight, bf16x3=False, stats=False, pivot=None):
        """stats (bf16x3 only): also return the per-channel partial sums of the output and of its squares from the kernel's
        epilogue (`ops.conv3d_k3_bf16x3_stats`, sums of value - pivot_c), for the training-mode BatchNorm behind the layer."""
        from . import ops
        ctx.save_for_backward(x, weight)
        ctx.bf16x3 = bool(bf16x3)
        if ctx.bf16x3 and stats:
            y, parts = ops.conv3d_k3_bf16x3_stats(x, ops.split_conv_weight(weight), pivot)
            ctx.mark_non_differentiable(parts)
            return y, parts
        if ctx.bf16x3:   # forward and input gradient on the bf16 matrix cores, three-term split (csrc/costreg_bf16.hip)
            return ops.conv3d_k3_bf16x3(x, ops.split_conv_weight(weight), None, None, False)
        return ops.conv3d_k3_mfma(x, ops.permute_conv_weight(weight), None, None, False)

    @staticmethod
    def backward(ctx, gy, gparts=None):
        from . import ops
        x, weight = ctx.saved_tensors
        gy = gy.contiguous()
        gx = gw = None
        if ctx.needs_input_grad[0]:
            wflip = weight.detach().transpose(0, 1).flip(2, 3, 4).contiguous()      # (Cin, Cout, 3,3,3)
            if ctx.bf16x3 and wflip.shape[0] % 64 == 0:
                src = gy
                if wflip.shape[0] >= 256:
                    # four or more blocks of output channels per tile would each cut the same grad_out values into bf16 pieces:
                    # one packing pass and the DMA-fed form instead (conv0: 4.78 -> 4.65 ms, the same bits).  The SCL copy
                    # (larger than grad_out itself) lives for this one convolution: it comes from the caching allocator and goes
                    # back to it when `src` dies below -- the packing kernel writes the zero border itself, nothing is kept
                    src = ops.scl_pack(gy)
                gx = ops.conv3d_k3_bf16x3(src, ops.split_conv_weight(wflip), None, None, False)
                del src
            else:
                gx = ops.conv3d_k3_mfma(gy, ops.permute_conv_weight(wflip), None, None, False)
        if ctx.needs_input_grad[1]:   # bf16x3: csrc/costreg_dw_bf16.hip (rows read as float4)
            gw = ops.conv3d_k3_dw(x, gy, 0, 1, ctx.bf16x3 and x.shape[-1] % 4 == 0)
        return gx, gw, None, None, None


class _ConvK3S2(torch.autograd.Function):
    """Conv3d(kernel 3, stride 2, padding 1, no bias) of conv1 / conv3 (mvsnet.py:77,80) under autograd: forward on
    `ops.conv3d_k3_mfma(stride=2)`; the input gradient is the transposed convolution of grad_out with the same weight
    (`ops.convT3d_k3_s2_mfma`: the (Cout,Cin,3,3,3) tensor read as a ConvTranspose3d weight), the weight gradient
    `ops.conv3d_k3_dw(stride=2)`.  D, H, W even (the network asks for multiples of 4)."""

    @staticmethod
    def forward(ctx, x, weight, bf16x3=False, split_skip=False):
        """split_skip: also return x itself as a second output, for the skip connection that reads it (mvsnet.py:109-111).  The
        input then has this one consumer, both gradients arrive here together, and the skip's is added in the epilogue of the
        input-gradient kernel instead of by a pass of autograd's own over the full-resolution tensor."""
        from . import ops
        ctx.save_for_backward(x, weight)
        ctx.bf16x3 = bool(bf16x3)
        ctx.set_materialize_grads(False)
        if ctx.bf16x3:
            y = ops.conv3d_k3_s2_bf16x3(x, ops.split_conv_weight(weight, 1), None, None, False)
        else:
            y = ops.conv3d_k3_mfma(x, ops.permute_conv_weight(weight), None, None, False, 2)
        return (y, x.view_as(x)) if split_skip else y

    @staticmethod
    def backward(ctx, gy, gskip=None):
        from . import ops
        x, weight = ctx.saved_tensors
        gx = gw = None
        if gy is None:   # only the skip branch reached the loss
            return gskip, None, None, None
        gy = gy.contiguous()
        if ctx.needs_input_grad[0]:
            res = None if gskip is None else gskip.contiguous()
            if ctx.bf16x3 and weight.shape[1] % 64 == 0:   # the (Cout,Cin,3,3,3) tensor read as a ConvTranspose3d weight
                gx = ops.convT3d_k3_s2_bf16x3(gy, ops.split_conv_weight(weight.detach(), 2), None, None, res, False)
            else:
                gx = ops.convT3d_k3_s2_mfma(gy, ops.permute_convT_weight(weight.detach()), None, None, res, False)
        if ctx.needs_input_grad[1]:
            gw = ops.conv3d_k3_dw(x, gy, 0, 2, ctx.bf16x3 and x.shape[-1] % 8 == 0)
        return gx, gw, None, None


class _ConvT3S2(torch.autograd.Function):
    """ConvTranspose3d(kernel 3, stride 2, padding 1, output_padding 1, no bias) of conv9 / conv11 (mvsnet.py:92-100) under
    autograd, the mirror image of `_ConvK3S2`: forward on `ops.convT3d_k3_s2_mfma`, input gradient = the stride-2
    convolution of grad_out with the (Cin,Cout,3,3,3) weight read as a Conv3d weight, weight gradient = the stride-2
    weight-gradient kernel with the two tensors exchanged."""

    @staticmethod
    def forward(ctx, x, weight, bf16x3=False, stats=False, pivot=None):
        """stats (bf16x3 only): also return the partial sums of the output's BatchNorm statistics from the kernel's epilogue
        (`ops.convT3d_k3_s2_bf16x3_stats`); an empty tensor where the shape has no such form."""
        from . import ops
        ctx.save_for_backward(x, weight)
        ctx.bf16x3 = bool(bf16x3)
        if ctx.bf16x3 and stats:
            got = ops.convT3d_k3_s2_bf16x3_stats(x, ops.split_conv_weight(weight, 2), pivot)
            if got is None:
                y, parts = ops.convT3d_k3_s2_bf16x3(x, ops.split_conv_weight(weight, 2), None, None, None, False), x.new_empty(0, dtype=torch.float64)
            else:
                y, parts = got
            ctx.mark_non_differentiable(parts)
            return y, parts
        if ctx.bf16x3:
            return ops.convT3d_k3_s2_bf16x3(x, ops.split_conv_weight(weight, 2), None, None, None, False)
        return ops.convT3d_k3_s2_mfma(x, ops.permute_convT_weight(weight), None, None, None, False)

    @staticmethod
    def backward(ctx, gy, gparts=None):
        from . import ops
        x, weight = ctx.saved_tensors
        gy = gy.contiguous()
        gx = gw = None
        if ctx.needs_input_grad[0]:
            if ctx.bf16x3 and weight.shape[0] % 64 == 0:   # the (Cin,Cout,3,3,3) tensor read as a Conv3d weight
                gx = ops.conv3d_k3_s2_bf16x3(gy, ops.split_conv_weight(weight.detach(), 1), None, None, False)
            else:
                gx = ops.conv3d_k3_mfma(gy, ops.permute_conv_weight(weight.detach()), None, None, False, 2)
        if ctx.needs_input_grad[1]:
            gw = ops.conv3d_k3_dw(gy, x, 0, 2, ctx.bf16x3 and gy.shape[-1] % 8 == 0)
        return gx, gw, None, None, None


# Test hook (tests/test_gpu_parity.py, tests/test_f3_goldens.py G12c): the ReLU DECISIONS of a training pass recorded, or imposed.
# None (always, outside those tests) | ("record", {}) -- filled with {BatchNorm module: mask of relu's positive side} -- |
# ("apply", {BatchNorm module: bool mask}): the layer computes bn(x) * mask (+ residual) instead of relu(bn(x)) (+ residual).
# With the decisions of ONE pass imposed on two routes (or the reference's on ours) no activation can fall on the other side
# of zero, and gradients can be compared element-wise instead of by direction.
RELU_MASKS = None


def _bn_relu_train(bn: nn.BatchNorm3d, x: torch.Tensor, residual: Optional[torch.Tensor] = None,
                   parts: Optional[torch.Tensor] = None, pivot: Optional[torch.Tensor] = None) -> torch.Tensor:
    """relu(bn(x)) with batch statistics (module.py:26-37; mvsnet.py:92-100) on the streaming kernels of
    csrc/costreg_bn.hip -- two passes over x forward, ReLU in the second, the mask recomputed going backward -- and the
    running statistics updated the way torch.nn.BatchNorm3d does (momentum, unbiased variance, num_batches_tracked)."""
    from . import ops
    hook = RELU_MASKS
    # residual: added after the ReLU; parts: the statistics' partial sums from the convolution's epilogue (no pass over x for them)
    if hook is None:
        out, mean, invstd = ops.bn3d_relu_train(x, bn.weight, bn.bias, bn.eps, True, residual, parts, pivot)
    elif hook[0] == "record":
        out, mean, invstd = ops.bn3d_relu_train(x, bn.weight, bn.bias, bn.eps, True, None, parts, pivot)
        hook[1][bn] = out.detach() > 0
        out = out if residual is None else out + residual
    else:
        out, mean, invstd = ops.bn3d_relu_train(x, bn.weight, bn.bias, bn.eps, False, None, parts, pivot)
        out = out * hook[1][bn].to(out.dtype)
        out = out if residual is None else out + residual
    if bn.track_running_stats and bn.running_mean is not None:
        with torch.no_grad():
            m = x.numel() // x.shape[1]
            bn.num_batches_tracked += 1
            mom = bn.momentum if bn.momentum is not None else 1.0 / float(bn.num_batches_tracked)
            var = (1.0 / (invstd * invstd) - bn.eps).clamp_min_(0.0) * (m / max(m - 1, 1))
            bn.running_mean.mul_(1.0 - mom).add_(mean, alpha=mom)
            bn.running_var.mul_(1.0 - mom).add_(var, alpha=mom)
    return out


# training: the statistics of a BatchNorm behind a stride-1 bf16x3 convolution come from that convolution's epilogue (partial sums
# per block, finished in a fixed order) instead of a pass of their own over the tensor; MVSDET_FUSED_BN_STATS=0: the separate pass
FUSED_BN_STATS = os.environ.get("MVSDET_FUSED_BN_STATS", "1") != "0"


def ops_option(name: str) -> int:
    from . import ops
    return int(ops.get_option(name))


def _bn_hip_ok(bn: nn.BatchNorm3d, x: torch.Tensor) -> bool:
    return bn.training and x.is_cuda and x.dtype == torch.float32 and bn.affine


# module -> {device: the stream the second half of its views runs on} (CostRegNet3DGS.view_streams); outside the modules because a
# stream does not pickle, per module because the layers' SCL buffers are keyed by the stream
_SIDE_STREAMS = weakref.WeakKeyDictionary()


class _HeadConv(torch.autograd.Function):
    """The head Conv3d(64 -> 2, k=3, p=1, bias) with forward and both gradients on the streaming kernels of
    csrc/costreg_head.hip (MIOpen: 363 ms for one forward + backward at the reference-true shape; here about 2 ms)."""

    @staticmethod
    def forward(ctx, x, weight, bias, bf16x3=False):
        from . import ops
        ctx.save_for_backward(x, weight)
        ctx.bf16x3 = bool(bf16x3)   # the weight gradient on the bf16 matrix cores (three-term split operands)
        return ops.conv3d_k3_cout2(x, weight, bias)

    @staticmethod
    def backward(ctx, gy):
        from . import ops
        x, weight = ctx.saved_tensors
        gx, gw = ops.conv3d_k3_cout2_backward(x, weight.detach(), gy.contiguous(), 32, ctx.bf16x3)
        gb = gy.sum(dim=(0, 2, 3, 4)) if ctx.needs_input_grad[2] else None
        return (gx if ctx.needs_input_grad[0] else None), (gw if ctx.needs_input_grad[1] else None), gb, None


class _ConvBnReLU3d(nn.Module):
    """Conv3d(no bias) -> BatchNorm3d -> ReLU, with the reference's sub-module names `conv` and `bn` (module.py:26)."""

    def __init__(self, cin: int, cout: int, stride: int = 1):
        super().__init__()
        self.conv = nn.Conv3d(cin, cout, 3, stride=stride, padding=1, bias=False)
        self.bn = nn.BatchNorm3d(cout)

    def forward(self, x):
        return torch.relu_(self.bn(self.conv(x)))


def _up(cin: int, cout: int) -> nn.Sequential:
    # indices 0 / 1 / 2 of the Sequential are the reference's parameter names (mvsnet.py:92-100)
    return nn.Sequential(nn.ConvTranspose3d(cin, cout, 3, stride=2, padding=1, output_padding=1, bias=False),
                         nn.BatchNorm3d(cout), nn.ReLU(inplace=True))


class CostRegNet3DGS(DerivedTensorsMixin, nn.Module):
    def __init__(self, in_channels: int = 256, base: int = 64):
        super().__init__()
        self._init_derived_hooks()
        self.conv0 = _ConvBnReLU3d(in_channels, base)
        self.conv1 = _ConvBnReLU3d(base, 2 * base, stride=2)
        self.conv2 = _ConvBnReLU3d(2 * base, 2 * base)
        self.conv3 = _ConvBnReLU3d(2 * base, 4 * base, stride=2)
        self.conv4 = _ConvBnReLU3d(4 * base, 4 * base)
        self.conv9 = _up(4 * base, 2 * base)
        self.conv11 = _up(2 * base, base)
        self.prob = nn.Conv3d(base, 2, 3, stride=1, padding=1)
        # under autograd every convolution (stride 1, stride 2, transposed, head) uses our forward / dX / dW kernels
        self.hip_backward = True
        # eval route of the stride-1 layers: "bf16x3" = bf16 matrix cores on split operands (logits within 2e-6 .. 4e-6 of
        # fp32: tools/study/split_bf16_emulation.py), "fp32" = the fp32 MFMA kernels (bit-level fp32 FMA sums)
        # MVSDET_COSTREG_PRECISION=fp32 makes the fp32 route the default of new modules
        self.matrix_precision = os.environ.get("MVSDET_COSTREG_PRECISION", "bf16x3")
        if self.matrix_precision not in ("bf16x3", "fp32"):
            raise ValueError(f"MVSDET_COSTREG_PRECISION must be 'bf16x3' or 'fp32', got {self.matrix_precision!r}")
        # eval route on bf16x3: "scl" = every layer hands the next one its output already cut into bf16 pieces (SCL / PSCL
        # forms: no fp32 round trip through a packing pass or a strided gather, inputs by LDS-DMA); "f32" = fp32 tensors
        # between the layers (round 3).  Same values bit for bit.
        # conv0 (256 -> 64 on the fp32 variance volume, 41 % of a scene) in the eval chain: "fp16mx" (default) = one fp16 product + ONE
        # block-scaled FP6 product that carries both correction terms (csrc/costreg_mx.h: 11.6 instead of 21 matrix-pipe units per 8
        # channels; 3.9 against 4.3 ms; G8 logits 5e-6 from the reference's instead of 2e-6, G13 depth_coding 7e-5 instead of 5e-5:
        # the bar is 1e-4); "bf16x3" = three bf16 products per fp32-equivalent product.  Training keeps bf16x3.
        self.conv0_precision = os.environ.get("MVSDET_CONV0_PRECISION", "fp16mx")
        if self.conv0_precision not in ("bf16x3", "fp16mx"):
            raise ValueError(f"MVSDET_CONV0_PRECISION must be 'bf16x3' or 'fp16mx', got {self.conv0_precision!r}")
        self.skip_in_head = False   # True: conv0 + conv11(x) formed by the head while it stages its input (round 4's first form)
        self.layer_forms = "scl"
        # The eval chain on TWO halves of the views, the second on a stream of its own (views are independent: BatchNorm is an
        # affine): several of its kernels have grids of 3.1 rounds of the chip (800 blocks on 256 CUs at 40 views), and the tail of
        # one half's kernel fills with the other half's blocks -- 7.6 -> 7.35 ms, the same bits (three pieces 7.6, four 7.8:
        # tools/study/two_stream_network.py).  1 = one batch on the caller's stream.
        self.view_streams = int(os.environ.get("MVSDET_COSTREG_STREAMS", "2"))
        # (layer, kind, shape, device) -> the SCL / PSCL buffers a layer of the chain writes; their zero borders are written once
        # and the buffers refilled on every call.  Every buffer carries the event behind its last use: whichever stream takes it
        # next waits for that event (mvsdet_amd/scratch.py) -- no dependence on which stream ran the module before
        self._scl = EventPool(64)

    def _chain_ok(self, x) -> bool:
        """The eval route on which every layer hands the next one its output already cut into bf16 pieces."""
        return (x.is_cuda and x.dtype == torch.float32 and not torch.is_grad_enabled() and not self.training
                and self.matrix_precision == "bf16x3" and self.layer_forms == "scl"
                and all(c.out_channels % 64 == 0 for c in (self.conv0.conv, self.conv1.conv, self.conv2.conv, self.conv3.conv,
                                                           self.conv4.conv, self.conv9[0], self.conv11[0])))

    def _buf(self, leases, name, kind, shape, dev):
        """The SCL / PSCL buffer a layer of the chain writes: allocated and zeroed once per (layer, shape, device) -- the kernels
        write interior voxels only, the zero border stays -- and refilled on every call.  Taken from the module's pool for the
        duration of one `_forward_chain` (`leases` collects what that call must hand back)."""
        from . import ops
        make = ops.scl_empty if kind == "scl" else ops.pscl_empty
        lease = self._scl.acquire((name, kind, tuple(shape), str(dev)), lambda: make(shape, dev), lambda b: (b.data,), dev)
        leases.append(lease)
        return lease.buf

    def _forward_chain(self, x):
        """mvsnet.py:104-112 with the layer-to-layer forms of include/mvsdet_hip.h: conv0 -(fp32 skip, PSCL)-> conv1 -(SCL)->
        conv2 -(fp32 skip, PSCL)-> conv3 -(SCL)-> conv4 -(SCL)-> conv9 -(SCL)-> conv11 -(fp32)-> [+ conv0] prob.  Same values as the
        fp32-handover route bit for bit (a producer cuts exactly the pieces the consumer would have cut)."""
        from . import ops
        n, _, d, h, w = x.shape
        dev = x.device
        b = self.conv0.conv.out_channels
        leases = []

        # all seven weight tensors cut into their bf16 pieces by ONE launch, on every call: an in-place update is always seen
        layers = [(self.conv0.conv, 0), (self.conv1.conv, 1), (self.conv2.conv, 0), (self.conv3.conv, 1), (self.conv4.conv, 0),
                  (self.conv9[0], 2), (self.conv11[0], 2)]
        wsplit = dict(zip((id(m) for m, _ in layers), ops.split_conv_weights([(m.weight, o) for m, o in layers])))

        def cbr(layer, inp, order, outputs, name, oshape):
            sc, sh = _bn_affine(layer.bn)
            wq = wsplit[id(layer.conv)]
            kw = {}
            if "scl" in outputs:
                kw["scl_out"] = self._buf(leases, name, "scl", oshape, dev)
            if "pscl" in outputs:
                kw["pscl_out"] = self._buf(leases, name, "pscl", oshape, dev)
            fn = ops.conv3d_k3_bf16x3 if order == 0 else ops.conv3d_k3_s2_bf16x3
            return fn(inp, wq, sc, sh, True, outputs=outputs, **kw)

        def up(seq, inp, skip, outputs, name, oshape):
            sc, sh = _bn_affine(seq[1])
            kw = {"scl_out": self._buf(leases, name, "scl", oshape, dev)} if "scl" in outputs else {}
            return ops.convT3d_k3_s2_bf16x3(inp, wsplit[id(seq[0])], sc, sh, skip, True, outputs=outputs, **kw)

        if self.conv0_precision == "fp16mx":
            sc0, sh0 = _bn_affine(self.conv0.bn)
            full, full_p = ops.conv3d_k3_fp16mx(x, ops.split_conv_weight_mx(self.conv0.conv.weight), sc0, sh0, True, outputs=("f32", "pscl"),
                                                pscl_out=self._buf(leases, "conv0", "pscl", (n, b, d, h, w), dev))
        else:
            full, full_p = cbr(self.conv0, x, 0, ("f32", "pscl"), "conv0", (n, b, d, h, w))
        h1 = cbr(self.conv1, full_p, 1, ("scl",), "conv1", (n, 2 * b, d // 2, h // 2, w // 2))
        half, half_p = cbr(self.conv2, h1, 0, ("f32", "pscl"), "conv2", (n, 2 * b, d // 2, h // 2, w // 2))
        q1 = cbr(self.conv3, half_p, 1, ("scl",), "conv3", (n, 4 * b, d // 4, h // 4, w // 4))
        q2 = cbr(self.conv4, q1, 0, ("scl",), "conv4", (n, 4 * b, d // 4, h // 4, w // 4))
        half2 = up(self.conv9, q2, half, ("scl",), "conv9", (n, 2 * b, d // 2, h // 2, w // 2))
        # mvsnet.py:111-112: x = conv0 + conv11(x); prob(x).  With one block per (PD, PH) in the transposed kernel the addition was
        # cheaper in the head's staging (conv11's epilogue then only stores); with all eight classes in one block the skip tensor read
        # in conv11's epilogue costs less than a second input costs the head (0.65 + 0.31 against 0.48 + 0.53 ms): same bits either way
        # (one fp32 addition of the same two values)
        if self.skip_in_head:
            up11 = up(self.conv11, half2, None, ("f32",), "conv11", (n, b, d, h, w))
            logits = ops.conv3d_k3_cout2_sum(full, up11, self.prob.weight.detach(), self.prob.bias.detach())
        else:
            up11 = up(self.conv11, half2, full, ("f32",), "conv11", (n, b, d, h, w))
            logits = ops.conv3d_k3_cout2_sum(up11, None, self.prob.weight.detach(), self.prob.bias.detach())
        self._scl.release(leases, dev)   # every reader of the buffers is enqueued: one event behind them on this stream
        return logits

    @fp32_under_autocast
    def forward(self, x):
        if any(s % 4 for s in x.shape[2:]):
            raise ValueError(f"CostRegNet3DGS: D, H, W must be divisible by 4, got {tuple(x.shape[2:])}")
        if self._chain_ok(x):
            n = x.shape[0]
            k = min(int(self.view_streams), n // 4)
            if k < 2:
                return self._forward_chain(x)
            # the views in k contiguous pieces, piece 0 on the caller's stream, the others on streams of the module's own (views are
            # independent in eval mode: the same bits whatever the split)
            dev = x.device
            cur = torch.cuda.current_stream(dev)
            mine = _SIDE_STREAMS.setdefault(self, {})
            sides = mine.get(str(dev))
            if sides is None:
                sides = mine[str(dev)] = []
            while len(sides) < k - 1:
                sides.append(torch.cuda.Stream(device=dev))
            bounds = [(n * i + k - 1) // k for i in range(k + 1)]      # piece i = views bounds[i] .. bounds[i + 1]: the first ones larger
            parts = [None] * k
            for i in range(1, k):
                sides[i - 1].wait_stream(cur)          # x is ready; the caller keeps it alive until this call returns
                with torch.cuda.stream(sides[i - 1]):
                    parts[i] = self._forward_chain(x[bounds[i]:bounds[i + 1]])
            parts[0] = self._forward_chain(x[:bounds[1]])
            for i in range(1, k):
                cur.wait_stream(sides[i - 1])
                parts[i].record_stream(cur)            # allocated under a side stream, read by the concatenation on this one
            return torch.cat(parts, 0)
        full = self._cbr(self.conv0, x)                           # (N, 64, D, H, W)
        # the stride-2 layers hand their input back as the skip tensor (`_ConvK3S2`: its gradient joins the input gradient in
        # that layer's own kernel)
        h1, full = self._cbr(self.conv1, full, split_skip=True)
        half = self._cbr(self.conv2, h1)                          # (N, 128, D/2, H/2, W/2)
        q1, half = self._cbr(self.conv3, half, split_skip=True)
        quarter = self._cbr(self.conv4, q1)                       # (N, 256, D/4, H/4, W/4)
        half = self._up(self.conv9, quarter, half)        # half + relu(bn(deconv(quarter)))
        full = self._up(self.conv11, half, full)
        return self._head(full)                       # (N, 2, D, H, W)

    def _cbr(self, layer, x, split_skip: bool = False):
        """A ConvBnReLU3D layer (mvsnet.py:76-82: conv0..conv4, stride 1 or 2).  Without autograd and in eval mode
        (BatchNorm = per-channel affine) the fp32-MFMA kernel of csrc/costreg_conv0.hip runs conv + BN + ReLU in one
        pass -- conv0 at the reference-true shape: 15.7 ms (130 TFLOP/s) instead of 33.6 + 0.5 ms for MIOpen, the same
        fp32 FMA sums."""
        conv, bn = layer.conv, layer.bn
        if (x.is_cuda and x.dtype == torch.float32 and not torch.is_grad_enabled() and not self.training
                and conv.out_channels % 64 == 0 and conv.stride in ((1, 1, 1), (2, 2, 2))):
            from . import ops
            scale, shift = _bn_affine(bn)
            if self.matrix_precision == "bf16x3":
                # the bf16 matrix cores with three-term split operands (csrc/costreg_bf16.hip): conv0 4.8 ms instead of 15.2
                # on the fp32 MFMA; the stride-2 layers as sums over the 8 parity classes of their input
                if layer is self.conv0 and self.conv0_precision == "fp16mx":
                    y = ops.conv3d_k3_fp16mx(x, ops.split_conv_weight_mx(conv.weight), scale, shift, True)   # as the layer-form chain
                elif conv.stride == (1, 1, 1):
                    y = ops.conv3d_k3_bf16x3(x, ops.split_conv_weight(conv.weight), scale, shift, True)
                else:
                    y = ops.conv3d_k3_s2_bf16x3(x, ops.split_conv_weight(conv.weight, 1), scale, shift, True)
                return (y, x) if split_skip else y
            wperm = ops.permute_conv_weight(conv.weight)   # a few MB at most, negligible next to the convolution
            y = ops.conv3d_k3_mfma(x, wperm, scale, shift, True, conv.stride[0])
            return (y, x) if split_skip else y
        if (self.hip_backward and x.is_cuda and x.dtype == torch.float32 and torch.is_grad_enabled()
                and conv.stride in ((1, 1, 1), (2, 2, 2)) and conv.out_channels % 64 == 0 and conv.in_channels % 64 == 0):
            # autograd: convolution forward / backward on our kernels; BatchNorm + ReLU too when it uses batch statistics
            skip = x
            parts = pivot = None
            if conv.stride == (1, 1, 1):
                # the BatchNorm's statistics from the convolution's epilogue (16x16x32 form of the kernel: the default)
                if (FUSED_BN_STATS and self.matrix_precision == "bf16x3" and _bn_hip_ok(bn, x) and x.shape[0] * x[0, 0].numel() > 1
                        and ops_option("conv_mfma16") and ops_option("conv_subpairs") != 2):
                    # sums around the running mean (read by the convolution and by the BatchNorm's finishing kernel before the
                    # in-place update of the buffer that follows them on the same stream)
                    pivot = bn.running_mean.detach() if bn.running_mean is not None else None
                    y, parts = _ConvK3S1.apply(x, conv.weight, True, True, pivot)
                else:
                    y = _ConvK3S1.apply(x, conv.weight, self.matrix_precision == "bf16x3")
            elif split_skip:
                y, skip = _ConvK3S2.apply(x, conv.weight, self.matrix_precision == "bf16x3", True)
            else:
                y = _ConvK3S2.apply(x, conv.weight, self.matrix_precision == "bf16x3")
            y = _bn_relu_train(bn, y, None, parts, pivot) if _bn_hip_ok(bn, y) else torch.relu_(bn(y))
            return (y, skip) if split_skip else y
        return (layer(x), x) if split_skip else layer(x)

    def _up(self, seq, x, skip):
        """mvsnet.py:110-111: skip + Sequential(ConvTranspose3d, BatchNorm3d, ReLU)(x); one fp32-MFMA kernel per output
        parity in (d, h) (csrc/costreg_conv0.hip) without autograd in eval mode."""
        deconv, bn = seq[0], seq[1]
        if (x.is_cuda and x.dtype == torch.float32 and not torch.is_grad_enabled() and not self.training
                and deconv.out_channels % 64 == 0):
            from . import ops
            scale, shift = _bn_affine(bn)
            if self.matrix_precision == "bf16x3":
                # 8 output parity classes = 8 small stride-1 convolutions over the coarse input (csrc/costreg_bf16.hip)
                # the coarse input in split channel-last form: a buffer of this call's own (the packing kernel writes its border)
                xs = ops.scl_pack(x)
                return ops.convT3d_k3_s2_bf16x3(xs, ops.split_conv_weight(deconv.weight, 2), scale, shift, skip, True)
            wperm = ops.permute_convT_weight(deconv.weight)
            return ops.convT3d_k3_s2_mfma(x, wperm, scale, shift, skip, True)
        if (self.hip_backward and x.is_cuda and x.dtype == torch.float32 and torch.is_grad_enabled()
                and deconv.out_channels % 64 == 0 and deconv.in_channels % 64 == 0):
            if FUSED_BN_STATS and self.matrix_precision == "bf16x3" and _bn_hip_ok(bn, x):
                # the BatchNorm's statistics from the transposed kernel's epilogue, around the running mean
                pivot = bn.running_mean.detach() if bn.running_mean is not None else None
                y, parts = _ConvT3S2.apply(x, deconv.weight, True, True, pivot)
                if parts.numel():
                    return _bn_relu_train(bn, y, skip, parts, pivot)
                return _bn_relu_train(bn, y, skip)
            y = _ConvT3S2.apply(x, deconv.weight, self.matrix_precision == "bf16x3")
            if _bn_hip_ok(bn, y):
                return _bn_relu_train(bn, y, skip)   # the skip addition in the BatchNorm's second pass
            return skip + torch.relu_(bn(y))
        return skip + seq(x)

    def _head(self, full):
        """mvsnet.py:112.  Two output channels make a poor GEMM (MIOpen: 8.2 ms at the reference-true shape); without
        autograd the streaming HIP kernel of csrc/costreg_head.hip does it in a fraction of that."""
        if full.is_cuda and full.dtype == torch.float32 and not torch.is_grad_enabled():
            from . import ops
            return ops.conv3d_k3_cout2(full, self.prob.weight.detach(), self.prob.bias.detach())
        if self.hip_backward and full.is_cuda and full.dtype == torch.float32 and torch.is_grad_enabled():
            return _HeadConv.apply(full, self.prob.weight, self.prob.bias, self.matrix_precision == "bf16x3")
        return self.prob(full)

    @staticmethod
    def flops(n: int, d: int, h: int, w: int, in_channels: int = 256, base: int = 64) -> float:
        """Multiply-add count x2 of one forward pass (for the MFMA roofline of this network)."""
        v = n * d * h * w
        k = 27 * 2
        full, half, quarter = v, v / 8, v / 64
        return k * (in_channels * base * full + base * 2 * base * half + (2 * base) ** 2 * half +
                    2 * base * 4 * base * quarter + (4 * base) ** 2 * quarter +
                    4 * base * 2 * base * quarter + 2 * base * base * half + base * 2 * full)
