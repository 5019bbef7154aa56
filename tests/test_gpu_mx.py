"""conv0 on one fp16 + two block-scaled FP6 products per fp32-equivalent product (csrc/costreg_mx.h).

  * exact data: inputs and weights built so that EVERY piece of the scheme is exact -- xh = fp16(x), Q(xh), xr = x - xh, Q(xr) and
    the same for the weights, under the scales the kernel and the weight split derive -- so the kernel must return the float64 sum
    of the three product terms to fp32 rounding: operand maps, the 6-bit packing, fragment assembly, tap order and scale bytes;
  * random data: within 2^-14 of the output's scale of the float64 convolution (the scheme's own error; bf16x3: 2^-16);
  * the whole network on G8 (the reference module's logits): 1e-4, conv0 on this route;
  * values beyond fp16's range are cut on a block-uniform power of two: no cliff at 65504.
"""
import sys

import numpy as np
import pytest
import torch
import torch.nn.functional as F

from conftest import GOLDEN, load_golden

sys.path.insert(0, GOLDEN)
from lcg import lcg_fill_state, lcg_uniform  # noqa: E402

pytestmark = pytest.mark.gpu


def _exact_case(N, Cin, Cout, D, H, W, seed):
    g = torch.Generator().manual_seed(seed)
    ri = lambda lo, hi, shape: torch.randint(lo, hi + 1, shape, generator=g).double()   # noqa: E731
    # x = 128 k + j / 16, k in 8..15, j in 0..7:  fp16(x) = 128 k (ulp 1 in [1024, 2048)), stage exponent 8 -> Q(xh) = k / 2 exact,
    # xr = j / 16 -> Q(xr * 2^3) = j / 2 exact
    xh = 128.0 * ri(8, 15, (N, Cin, D, H, W))
    xr = ri(0, 7, (N, Cin, D, H, W)) / 16.0
    # w = +-p 2^-5 +- q 2^-16, p in 8..15, q in 0..7: fp16(w) = +-p 2^-5 (ulp 2^-12), Q(wh) = p / 2 and Q(wr) exact under their block scales
    sgn = lambda shape: torch.randint(0, 2, shape, generator=g).double() * 2 - 1   # noqa: E731
    wh = sgn((Cout, Cin, 3, 3, 3)) * ri(8, 15, (Cout, Cin, 3, 3, 3)) * 2.0 ** -5
    wr = sgn((Cout, Cin, 3, 3, 3)) * ri(0, 7, (Cout, Cin, 3, 3, 3)) * 2.0 ** -16
    ref = F.conv3d(xh, wh, padding=1) + F.conv3d(xh, wr, padding=1) + F.conv3d(xr, wh, padding=1)
    return (xh + xr).float(), (wh + wr).float(), ref


@pytest.mark.parametrize("N,Cin,Cout,D,H,W", [(1, 16, 64, 4, 12, 16), (2, 24, 128, 5, 13, 35), (1, 8, 64, 3, 8, 16), (1, 20, 64, 2, 9, 7)])
def test_every_piece_exact(gpu, N, Cin, Cout, D, H, W):
    from mvsdet_amd import ops
    x, w, ref = _exact_case(N, Cin, Cout, D, H, W, seed=Cin + W)
    assert torch.equal((x.double() - x.half().double()), (x.double() - (x.double() / 128).floor() * 128))   # the construction holds
    got = ops.conv3d_k3_fp16mx(x.to(gpu), ops.split_conv_weight_mx(w.to(gpu)), None, None, False)
    err = float((got.double().cpu() - ref).abs().max())
    assert err <= 2e-7 * float(ref.abs().max()) * 8, (err, float(ref.abs().max()))


@pytest.mark.parametrize("N,Cin,Cout,D,H,W", [(2, 64, 64, 4, 12, 32), (1, 256, 64, 4, 24, 16), (1, 30, 128, 6, 10, 20)])
def test_random_data_against_float64(gpu, N, Cin, Cout, D, H, W, record_property):
    from mvsdet_amd import ops
    g = torch.Generator().manual_seed(5)
    f = torch.randn((3, N, Cin, D, H, W), generator=g)
    x = (f * f).mean(0) - f.mean(0) ** 2                                   # variance-like: what the sweep hands conv0
    w = torch.randn((Cout, Cin, 3, 3, 3), generator=g) / (27 * Cin) ** 0.5
    sc, sh = torch.rand(Cout, generator=g) + 0.5, torch.randn(Cout, generator=g) * 0.1
    ref = torch.relu(F.conv3d(x.double(), w.double(), padding=1) * sc.double().view(1, -1, 1, 1, 1) + sh.double().view(1, -1, 1, 1, 1))
    xd, wd = x.to(gpu), w.to(gpu)
    got, pscl = ops.conv3d_k3_fp16mx(xd, ops.split_conv_weight_mx(wd), sc.to(gpu), sh.to(gpu), True, outputs=("f32", "pscl"))
    b3 = ops.conv3d_k3_bf16x3(xd, ops.split_conv_weight(wd), sc.to(gpu), sh.to(gpu), True)
    scale = float(ref.abs().max())
    err, err3 = float((got.double().cpu() - ref).abs().max()) / scale, float((b3.double().cpu() - ref).abs().max()) / scale
    record_property("fp16mx_err_of_scale", err)
    print(f"fp16mx {err:.2e} of scale, bf16x3 {err3:.2e}")
    assert err <= 2.0 ** -14, err
    # the parity-split SCL output is the fp32 output cut into bf16 pieces: the same as the bf16x3 kernel's form of the same values
    _, pscl3 = ops.conv3d_k3_bf16x3(got.new_zeros(0) if False else xd, ops.split_conv_weight(wd), sc.to(gpu), sh.to(gpu), True, outputs=("f32", "pscl"))
    assert pscl.data.shape == pscl3.data.shape


def test_strided_view_and_values_beyond_fp16(gpu):
    """A row-pitched view gives the contiguous tensor's bits; values beyond fp16's range (65504) are handled by the block-uniform power
    of two a stage is cut on (costreg_mx.h "fp16"): the result stays within the scheme's error of the float64 convolution, in the
    blocks that meet such values and in all others."""
    from mvsdet_amd import ops
    g = torch.Generator().manual_seed(9)
    big = torch.rand((1, 16, 4, 12, 48), generator=g).to(gpu)
    w = (torch.randn((64, 16, 3, 3, 3), generator=g) * 0.05).to(gpu)
    wq = ops.split_conv_weight_mx(w)
    view = big[..., :32]                                    # a row-pitched volume: w stride 1, h stride 48
    assert torch.equal(ops.conv3d_k3_fp16mx(view, wq, None, None, False), ops.conv3d_k3_fp16mx(view.contiguous(), wq, None, None, False))
    x = view.contiguous().clone()
    x[0, 3, 1, 5, 20] = 7.0e4                               # no fp16 without the stage's power of two
    x[0, 9, 2, 7, 25] = -3.0e9                              # another channel group
    x[0, :, 3, 2:4, 22:26] *= 1.0e5
    got = ops.conv3d_k3_fp16mx(x, wq, None, None, False)
    ref = F.conv3d(x.double().cpu(), w.double().cpu(), padding=1)
    err = (got.double().cpu() - ref).abs()
    assert torch.isfinite(got).all()
    # the blocks that meet the huge values (w >= 16): within the scheme's error of THEIR scale (a stage is cut on one power of two per
    # block: a value 2^-14 below the block's largest loses bits, as any fp16 does) ...
    assert float(err.max()) <= 2.0 ** -13 * float(ref.abs().max())
    # ... the blocks that do not (their halo ends at w = 16): untouched, at their own scale
    assert float(err[..., :12].max()) <= 2.0 ** -14 * float(ref[..., :12].abs().max())
    assert float(ref[..., :12].abs().max()) < 1e3 < float(ref.abs().max())
    with pytest.raises(ValueError):
        ops.conv3d_k3_fp16mx(view, wq[:, :1], None, None, False)


def test_the_three_forms_of_the_kernel_give_the_same_bits(gpu):
    """Library option conv_mx_th: 0 = the wave-specialised kernel (default), 8 / 12 = every wave does everything on 4 x 8 x 16 /
    4 x 12 x 16 tiles: the same sums in the same order."""
    from mvsdet_amd import _lib, ops
    g = torch.Generator().manual_seed(21)
    x = torch.rand((2, 40, 5, 13, 35), generator=g).to(gpu)
    w = (torch.randn((64, 40, 3, 3, 3), generator=g) * 0.05).to(gpu)
    wq = ops.split_conv_weight_mx(w)
    was = _lib.get_option("conv_mx_th")
    try:
        outs = []
        for form in (0, 8, 12):
            _lib.set_option("conv_mx_th", form)
            outs.append(ops.conv3d_k3_fp16mx(x, wq, None, None, True))
        assert torch.equal(outs[0], outs[1]) and torch.equal(outs[0], outs[2]) and float(outs[0].abs().max()) > 0
    finally:
        _lib.set_option("conv_mx_th", was)


def test_g8_with_conv0_on_the_mixed_format_route(gpu):
    """G8 (the REFERENCE CostRegNet_3DGS's logits, mvs_models/mvsnet.py:73-113) with conv0 on fp16 + MX FP6: 1e-4."""
    from mvsdet_amd.costreg import CostRegNet3DGS
    g = load_golden("g8_cost_regularisation")
    net = CostRegNet3DGS(256, 64).eval()
    with torch.no_grad():
        lcg_fill_state(net, int(g["weight_seed"]))
        shape = tuple(int(v) for v in g["in_shape"])
        x = torch.from_numpy(lcg_uniform(int(np.prod(shape)), int(g["input_seed"]))).reshape(shape).abs().to(gpu)
        net = net.to(gpu)
        assert net.conv0_precision == "fp16mx"            # the default of the eval chain
        y = net(x)
        net.conv0_precision = "bf16x3"
        y3 = net(x)
    e, e3 = float(np.abs(y.cpu().numpy() - g["logits"]).max()), float(np.abs(y3.cpu().numpy() - g["logits"]).max())
    print(f"G8 logits: fp16mx conv0 {e:.2e}, bf16x3 {e3:.2e}")
    assert e <= 1e-4 and not torch.equal(y, y3)
