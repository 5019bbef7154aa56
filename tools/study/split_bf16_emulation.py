"""Numerical study for VERDICT r2 item 3: which split-bf16 scheme keeps CostRegNet_3DGS (mvs_models/mvsnet.py:73-113)
within 1e-4 of the fp32 logits?  Every fp32 operand is cut into bf16 pieces (hi = bf16(x), mid = bf16(x - hi),
lo = bf16(x - hi - mid)); a product term of two pieces is exact in fp32 and MFMA accumulates in fp32, so a convolution of
pieces in fp32 on the CPU is a faithful model of the bf16 matrix-core route (up to summation order).
Run: python tools/study/split_bf16_emulation.py"""
import os
import sys

import numpy as np
import torch
import torch.nn.functional as F

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests", "golden"))
from lcg import lcg_fill_state, lcg_uniform  # noqa: E402
from mvsdet_amd.costreg import CostRegNet3DGS  # noqa: E402


def pieces(t, n):
    out, r = [], t
    for _ in range(n):
        p = r.bfloat16().float()
        out.append(p)
        r = r - p
    return out


SCHEMES = {  # (activation pieces, weight pieces, list of (i, j) product terms)
    "bf16x1": (1, 1, [(0, 0)]),
    "bf16x3": (2, 2, [(0, 0), (0, 1), (1, 0)]),
    "bf16x4": (2, 2, [(0, 0), (0, 1), (1, 0), (1, 1)]),
    "bf16x6": (3, 3, [(0, 0), (0, 1), (1, 0), (1, 1), (0, 2), (2, 0)]),
    "bf16x9": (3, 3, [(i, j) for i in range(3) for j in range(3)]),
}


def split_conv(fn, x, w, scheme, **kw):
    na, nw, terms = SCHEMES[scheme]
    xs, ws = pieces(x, na), pieces(w, nw)
    acc = None
    for i, j in sorted(terms, key=lambda t: -(t[0] + t[1])):   # small terms first
        y = fn(xs[i], ws[j], **kw)
        acc = y if acc is None else acc + y
    return acc


def forward(net, x, scheme, layers):
    """layers: set of layer names routed through the split scheme; the rest stays fp32."""
    def cbr(name, layer, t):
        conv, bn = layer.conv, layer.bn
        if name in layers:
            y = split_conv(F.conv3d, t, conv.weight, scheme, stride=conv.stride, padding=1)
        else:
            y = F.conv3d(t, conv.weight, stride=conv.stride, padding=1)
        return torch.relu(bn(y))

    def up(name, seq, t, skip):
        dc, bn = seq[0], seq[1]
        if name in layers:
            y = split_conv(F.conv_transpose3d, t, dc.weight, scheme, stride=2, padding=1, output_padding=1)
        else:
            y = F.conv_transpose3d(t, dc.weight, stride=2, padding=1, output_padding=1)
        return skip + torch.relu(bn(y))
    full = cbr("conv0", net.conv0, x)
    half = cbr("conv2", net.conv2, cbr("conv1", net.conv1, full))
    quarter = cbr("conv4", net.conv4, cbr("conv3", net.conv3, half))
    half = up("conv9", net.conv9, quarter, half)
    full = up("conv11", net.conv11, half, full)
    return net.prob(full)


def main():
    torch.set_num_threads(8)
    g = np.load(os.path.join(ROOT, "tests", "golden", "g8_cost_regularisation.npz"))
    net = CostRegNet3DGS(256, 64).eval()
    all_layers = {"conv0", "conv1", "conv2", "conv3", "conv4", "conv9", "conv11"}
    with torch.no_grad():
        lcg_fill_state(net, int(g["weight_seed"]))
        shape = tuple(int(v) for v in g["in_shape"])
        x_lcg = torch.from_numpy(lcg_uniform(int(np.prod(shape)), int(g["input_seed"]))).reshape(shape).abs()
        # a variance-like input: chi-square-ish values of N(0,1) features over 3 views, as the sweep produces
        gen = torch.Generator().manual_seed(3)
        f = torch.randn((3,) + shape, generator=gen)
        x_var = (f * f).mean(0) - f.mean(0) ** 2
        for tag, x, ref_gold in (("G8 input (LCG, |u|)", x_lcg, torch.from_numpy(g["logits"])), ("variance-like input", x_var, None)):
            ref = forward(net, x.double(), "bf16x1", set()) if False else None
            net64 = CostRegNet3DGS(256, 64).eval().double()
            net64.load_state_dict({k: v.double() for k, v in net.state_dict().items()})
            y64 = net64(x.double())
            y32 = forward(net, x, "bf16x1", set())
            print(f"== {tag}: logits |max| {float(y64.abs().max()):.3f}, std {float(y64.std()):.3f}")
            print(f"   fp32 torch vs fp64          : max {float((y32 - y64).abs().max()):.2e}")
            if ref_gold is not None:
                print(f"   fp32 torch vs reference G8  : max {float((y32 - ref_gold).abs().max()):.2e}")
            for scheme in ("bf16x1", "bf16x3", "bf16x4", "bf16x6", "bf16x9"):
                for lay_tag, lays in (("conv0 only", {"conv0"}), ("all 7 layers", all_layers)):
                    y = forward(net, x, scheme, lays)
                    e64 = float((y - y64).abs().max())
                    prob = torch.softmax(y[:, 0], 1)
                    p64 = torch.softmax(y64[:, 0], 1)
                    ep = float((prob - p64).abs().max())
                    extra = f", vs G8 {float((y - ref_gold).abs().max()):.2e}" if ref_gold is not None else ""
                    print(f"   {scheme:7s} {lay_tag:13s}: logits max err {e64:.2e}, prob max err {ep:.2e}{extra}")


if __name__ == "__main__":
    main()
