"""Gate for VERDICT r4 task 1: can conv0 of CostRegNet_3DGS (mvs_models/mvsnet.py:76,104: ConvBnReLU3D(256, 64), k3 s1 p1)
run as a transform-domain (Winograd) convolution on the three-term split-bf16 matrix path and stay within 2e-5 of the
fp32 logits (budget 1e-4 on the depth probabilities, mvsdet.py:470-475)?

Model of the kernel that would be built: input tiles transformed in fp32 by the staging waves (B^T d B, adds only),
weights transformed in fp32 once per call (G g G^T), BOTH cut into bf16 pieces hi / mid AFTER the transform, the
products hi*hi + hi*mid + mid*hi exact in fp32 (MFMA), fp32 accumulation over Cin and the non-transformed taps, output
transform A^T M A in fp32 in the epilogue.  Two forms:
  hw  : F(2x2, 3x3) over (H, W), the three depth taps direct            27 -> 12 products per output
  dhw : F(2x2x2, 3x3x3) over (D, H, W)                                   27 ->  8 products per output
and for comparison the same two forms with fp32 products (what Winograd alone costs) and the shipped direct bf16x3.

Run: python tools/study/winograd_bf16x3_emulation.py            (about two minutes on 8 cores)
"""
import os
import sys

import numpy as np
import torch
import torch.nn.functional as F

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests", "golden"))
sys.path.insert(0, os.path.join(ROOT, "tools", "study"))
from lcg import lcg_fill_state, lcg_uniform  # noqa: E402
from mvsdet_amd.costreg import CostRegNet3DGS  # noqa: E402
from split_bf16_emulation import pieces  # noqa: E402

BT = torch.tensor([[1, 0, -1, 0], [0, 1, 1, 0], [0, -1, 1, 0], [0, 1, 0, -1]], dtype=torch.float32)
G = torch.tensor([[1, 0, 0], [.5, .5, .5], [.5, -.5, .5], [0, 0, 1]], dtype=torch.float32)
AT = torch.tensor([[1, 1, 1, 0], [0, 1, -1, -1]], dtype=torch.float32)
TERMS3 = [(1, 0), (0, 1), (0, 0)]          # small terms first, as the shipped kernels accumulate


def _apply(mat, t, dim):
    """t contracted with mat along dim (mat: out x in), in t's dtype, as a chain of adds in a fixed order."""
    mat = mat.to(t.dtype)
    t = t.movedim(dim, -1)
    out = []
    for r in range(mat.shape[0]):
        acc = None
        for c in range(mat.shape[1]):
            m = float(mat[r, c])
            if m == 0.0:
                continue
            term = t[..., c] * m
            acc = term if acc is None else acc + term
        out.append(acc)
    return torch.stack(out, -1).movedim(-1, dim)


def _tiles(x, dim):
    """Unfold x along dim into overlapping windows of 4, stride 2: (..., n_tiles, ..., 4) with the window last."""
    return x.unfold(dim, 4, 2)


def winograd_conv3d(x, w, dims, split):
    """x (N,Ci,D,H,W) fp32 / fp64, w (Co,Ci,3,3,3); dims = the spatial axes (2,3,4) transformed (others direct).
    split: True = bf16x3 products, False = products in x's dtype.  Even sizes along the transformed axes."""
    tdims = sorted(dims)
    full = x.shape
    odd = [int(x.shape[ax] % 2) if ax in tdims else 0 for ax in (2, 3, 4)]
    if any(odd):                                    # an odd extent: one more zero row on the high side, cropped at the end
        x = F.pad(x, (0, odd[2], 0, odd[1], 0, odd[0]))
    n, ci, d, h, wd = x.shape
    xp = F.pad(x, (1, 1, 1, 1, 1, 1))
    # input transform: windows along every transformed axis, B^T along each window
    v = xp
    for ax in tdims:
        v = _tiles(v, ax)                           # window axes appended at the end in order of tdims
    for i in range(len(tdims)):
        v = _apply(BT, v, v.dim() - len(tdims) + i)
    u = w
    for ax in tdims:
        u = _apply(G, u, ax)                        # 3 -> 4 along that tap axis
    direct = [ax for ax in (2, 3, 4) if ax not in tdims]
    # v: (N, Ci, sD, sH, sW, p...) where transformed axes hold tile counts and direct axes padded extents
    vs = pieces(v, 2) if split else [v]
    us = pieces(u, 2) if split else [u]
    terms = TERMS3 if split else [(0, 0)]
    acc = None
    for i, j in terms:
        vv, uu = vs[i], us[j]
        # contraction over Ci and the direct taps: loop the direct taps (at most 3)
        if direct == [2]:
            y = None
            for kd in range(3):
                t = torch.einsum("nidhwpq,oipq->nodhwpq", vv[:, :, kd:kd + d], uu[:, :, kd])
                y = t if y is None else y + t
        elif direct == []:
            y = torch.einsum("nidhwrpq,oirpq->nodhwrpq", vv, uu)
        else:
            raise NotImplementedError
        acc = y if acc is None else acc + y
    m = acc
    for i in range(len(tdims)):
        m = _apply(AT, m, m.dim() - len(tdims) + i)           # 4 -> 2
    # m: (N, Co, tD|D, tH, tW, 2...) -> interleave
    if direct == [2]:
        m = m.permute(0, 1, 2, 3, 5, 4, 6).reshape(n, w.shape[0], d, h, wd)
    else:
        m = m.permute(0, 1, 2, 5, 3, 6, 4, 7).reshape(n, w.shape[0], d, h, wd)
    return m[:, :, :full[2], :full[3], :full[4]]


def direct_split(x, w):
    xs, ws = pieces(x, 2), pieces(w, 2)
    acc = None
    for i, j in TERMS3:
        y = F.conv3d(xs[i], ws[j], padding=1)
        acc = y if acc is None else acc + y
    return acc


def forward(net, x, conv_s1, which):
    """conv_s1(x, w) replaces the stride-1 k3 convolution of the layers named in `which`."""
    def cbr(name, layer, t):
        conv, bn = layer.conv, layer.bn
        if name in which:
            y = conv_s1(t, conv.weight)
        else:
            y = F.conv3d(t, conv.weight, stride=conv.stride, padding=1)
        return torch.relu(bn(y))

    def up(seq, t, skip):
        return skip + torch.relu(seq[1](F.conv_transpose3d(t, seq[0].weight, stride=2, padding=1, output_padding=1)))
    full = cbr("conv0", net.conv0, x)
    half = cbr("conv2", net.conv2, cbr("conv1", net.conv1, full))
    quarter = cbr("conv4", net.conv4, cbr("conv3", net.conv3, half))
    half = up(net.conv9, quarter, half)
    full = up(net.conv11, half, full)
    return net.prob(full), None


def report(tag, net, x, gold=None):
    net64 = CostRegNet3DGS(256, 64).double()
    net64.load_state_dict({k: v.double() for k, v in net.state_dict().items()})
    net64.train(net.training)
    y64 = forward(net64, x.double(), None, set())[0]
    p64 = torch.softmax(y64[:, 0], 1)
    print(f"== {tag}: input {tuple(x.shape)}, logits |max| {float(y64.abs().max()):.3f}, std {float(y64.std()):.3f}")
    rows = [
        ("fp32 direct (torch)", lambda a, b: F.conv3d(a, b, padding=1)),
        ("bf16x3 direct (shipped)", direct_split),
        ("fp32 winograd hw", lambda a, b: winograd_conv3d(a, b, (3, 4), False)),
        ("bf16x3 winograd hw", lambda a, b: winograd_conv3d(a, b, (3, 4), True)),
        ("fp32 winograd dhw", lambda a, b: winograd_conv3d(a, b, (2, 3, 4), False)),
        ("bf16x3 winograd dhw", lambda a, b: winograd_conv3d(a, b, (2, 3, 4), True)),
    ]
    out = {}
    for name, fn in rows:
        for lay_tag, lays in (("conv0", {"conv0"}), ("conv0+2+4", {"conv0", "conv2", "conv4"})):
            y = forward(net, x, fn, lays)[0]
            e = float((y - y64).abs().max())
            ep = float((torch.softmax(y[:, 0], 1) - p64).abs().max())
            extra = f"  vs fixture {float((y - gold).abs().max()):.2e}" if gold is not None else ""
            print(f"   {name:24s} {lay_tag:10s}: max |dlogit| {e:.2e}   max |dprob| {ep:.2e}{extra}", flush=True)
            out[(name, lay_tag)] = e
    return out


def conv0_only_error(x, w):
    """The error of conv0's own output, relative to its scale -- independent of the network behind it."""
    y64 = F.conv3d(x.double(), w.double(), padding=1)
    s = float(y64.abs().max())
    print(f"   conv0 alone (output |max| {s:.3f}):")
    for name, fn in (("fp32 direct", lambda a, b: F.conv3d(a, b, padding=1)), ("bf16x3 direct", direct_split),
                     ("fp32 winograd hw", lambda a, b: winograd_conv3d(a, b, (3, 4), False)),
                     ("bf16x3 winograd hw", lambda a, b: winograd_conv3d(a, b, (3, 4), True)),
                     ("fp32 winograd dhw", lambda a, b: winograd_conv3d(a, b, (2, 3, 4), False)),
                     ("bf16x3 winograd dhw", lambda a, b: winograd_conv3d(a, b, (2, 3, 4), True))):
        y = fn(x, w)
        d = (y - y64).abs()
        print(f"      {name:22s}: max {float(d.max()):.2e}  rms {float(d.pow(2).mean().sqrt()):.2e}  (max / scale {float(d.max()) / s:.2e})", flush=True)


def main():
    torch.set_num_threads(8)
    gdir = os.path.join(ROOT, "tests", "golden")
    g8 = np.load(os.path.join(gdir, "g8_cost_regularisation.npz"))
    g12 = np.load(os.path.join(gdir, "g12_cost_regularisation_grads.npz"))
    with torch.no_grad():
        # self-check of the transform algebra in float64
        xs = torch.randn(1, 3, 4, 6, 8, dtype=torch.float64)
        ws = torch.randn(5, 3, 3, 3, 3, dtype=torch.float64)
        ref = F.conv3d(xs, ws, padding=1)
        for dims in ((3, 4), (2, 3, 4)):
            err = float((winograd_conv3d(xs, ws, dims, False) - ref).abs().max())
            assert err < 1e-12, (dims, err)
        print("transform algebra exact in float64: ok")

        net = CostRegNet3DGS(256, 64).eval()
        lcg_fill_state(net, int(g8["weight_seed"]))
        shape = tuple(int(v) for v in g8["in_shape"])
        x = torch.from_numpy(lcg_uniform(int(np.prod(shape)), int(g8["input_seed"]))).reshape(shape).abs()
        report("G8 (eval mode, LCG |u| input)", net, x, torch.from_numpy(g8["logits"]))
        conv0_only_error(x, net.conv0.conv.weight)

        gen = torch.Generator().manual_seed(3)
        f = torch.randn((3,) + shape, generator=gen)
        xv = (f * f).mean(0) - f.mean(0) ** 2
        report("variance-like input (eval mode)", net, xv)
        conv0_only_error(xv, net.conv0.conv.weight)

        net12 = CostRegNet3DGS(256, 64).train()
        lcg_fill_state(net12, int(g12["weight_seed"]))
        shape = tuple(int(v) for v in g12["in_shape"])
        x12 = torch.from_numpy(lcg_uniform(int(np.prod(shape)), int(g12["input_seed"]))).reshape(shape).abs()
        # train-mode BatchNorm: batch statistics; running buffers are side effects we do not look at
        report("G12 (train mode, LCG input)", net12, x12, torch.from_numpy(g12["logits"]))


if __name__ == "__main__":
    main()
