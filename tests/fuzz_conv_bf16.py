#!/usr/bin/env python3
"""Randomised soak of the bf16x3 convolution family (csrc/costreg_bf16.hip; not collected by pytest; run on the GPU box):
    python tests/fuzz_conv_bf16.py [cases] [first_seed]
Random shapes (ragged tiles in d, h and w, channel counts that are no multiple of 8, single-voxel volumes, 64 / 128 / 192
output channels), the three layer kinds (stride 1 on the SCL form, on the fp32 tensor and on a row-pitched view of it; stride 2;
transposed), with and without affine / ReLU / residual -- each against a float64 evaluation of the SAME three products
(x_hi*w_hi + x_hi*w_mid + x_mid*w_hi): what remains is fp32 accumulation order, bounded by 4e-7 of the summed products.
Every fourth case is a WEIGHT GRADIENT (stride 1: csrc/costreg_dw_bf16.hip, W a multiple of 4; half of them stride 2 / transposed:
csrc/costreg_dw_s2_bf16.hip, even D and H, W a multiple of 8; one in four of the stride-1 ones the 64 -> 2 head's), any channel counts, any number of splits; and every eighth forward case has 256-1024 input channels on a small volume, so that the kernels split
the input channels over blocks (partial sums + epilogue kernel)."""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np  # noqa: E402
import torch  # noqa: E402
import torch.nn.functional as F  # noqa: E402

from mvsdet_amd import ops  # noqa: E402


def three_terms(fn, x, w, **kw):
    xh, xm = (t.double() for t in ops.split_bf16(x))
    wh, wm = (t.double() for t in ops.split_bf16(w))
    return fn(xh, wh, **kw) + fn(xh, wm, **kw) + fn(xm, wh, **kw)


def main():
    cases = int(sys.argv[1]) if len(sys.argv) > 1 else 60
    first = int(sys.argv[2]) if len(sys.argv) > 2 else 0
    dev = torch.device("cuda:0")
    bad = 0
    for seed in range(first, first + cases):
        rng = np.random.default_rng(90000 + seed)
        g = torch.Generator().manual_seed(90000 + seed)
        kind = ("s1", "s2", "t", "dw")[seed % 4]
        N = int(rng.integers(1, 4))
        Cin = int(rng.choice([1, 3, 8, 13, 16, 24, 40, 64, 100]))
        Cout = int(rng.choice([64, 64, 128, 192]))
        D, H, W = int(rng.integers(1, 11)), int(rng.integers(1, 30)), int(rng.integers(1, 40))
        if kind != "dw" and seed % 8 >= 4:   # few tiles, many input channels: the split form
            N, Cin = 1, int(rng.choice([256, 384, 512, 1024]))
            D, H, W = int(rng.integers(1, 5)), int(rng.integers(1, 13)), int(rng.integers(1, 17))
        dw_stride = 1
        head = kind == "dw" and seed % 16 == 3
        if kind == "dw":
            W = 4 * int(rng.integers(1, 12))
            Cout = int(rng.choice([1, 2, 31, 32, 33, 64, 70]))
            if head:
                Cin, Cout, W = int(rng.choice([16, 32, 64])), 2, 4 * int(rng.integers(1, 30))
            if seed % 8 >= 4:   # the stride-2 / transposed weight gradient (csrc/costreg_dw_s2_bf16.hip): even D, H, W a multiple of 8
                dw_stride, D, H, W = 2, 2 * int(rng.integers(1, 6)), 2 * int(rng.integers(1, 15)), 8 * int(rng.integers(1, 7))
                Cout = int(rng.choice([1, 3, 63, 64, 65, 128, 130]))
        x = torch.randn(N, Cin, D, H, W, generator=g) * float(rng.uniform(0.1, 4.0))
        affine, relu, resid = bool(rng.integers(0, 2)), bool(rng.integers(0, 2)), bool(rng.integers(0, 2))
        scale = (torch.rand(Cout, generator=g) + 0.5) if affine else None
        shift = (torch.randn(Cout, generator=g) * 0.1) if affine else None
        dv = lambda t: None if t is None else t.to(dev)   # noqa: E731
        if kind == "dw":
            st = dw_stride
            gy = torch.randn(N, Cout, D // st, H // st, W // st, generator=g) * float(rng.uniform(0.1, 4.0))
            shape = (Cout, Cin, 3, 3, 3)
            cw = lambda a, b: torch.nn.grad.conv3d_weight(a, shape, b, stride=st, padding=1)   # noqa: E731
            xh, xm = (t.double() for t in ops.split_bf16(x))
            yh, ym = (t.double() for t in ops.split_bf16(gy))
            ref = cw(xh, yh) + cw(xh, ym) + cw(xm, yh)
            mag = float(cw(x.abs().double(), gy.abs().double()).max())
            nsplit = int(rng.choice([0, 1, 2, 5, 8, 16, 40]))
            if head:   # the 64 -> 2 head's weight gradient (csrc/costreg_head.hip: conv3d_k3_cout2_dw_bf16x3_kernel)
                wz = torch.zeros(2, Cin, 3, 3, 3, device=dev)
                got = ops.conv3d_k3_cout2_backward(x.to(dev), wz, gy.to(dev), 5, True)[1]
            else:
                got = ops.conv3d_k3_dw(x.to(dev), gy.to(dev), nsplit, st, True)
            err = float((got.cpu().double() - ref).abs().max())
            tol = 4e-7 * mag + 1e-30
            if not (got.shape == ref.shape and err <= tol):
                bad += 1
                print(f"seed {seed} {'head ' if head else ''}dw stride {st} N={N} Cin={Cin} Cout={Cout} {D}x{H}x{W} nsplit={nsplit}: err {err:.3e} tol {tol:.3e}", flush=True)
            continue
        if kind == "s1":
            w = torch.randn(Cout, Cin, 3, 3, 3, generator=g) / (27 * Cin) ** 0.5
            want = three_terms(F.conv3d, x, w, padding=1)
            mag = float(F.conv3d(x.abs().double(), w.abs().double(), padding=1).max())
            res = torch.randn(want.shape, generator=g) if resid else None
            wq = ops.split_conv_weight(w.to(dev))
            pitched = torch.zeros(N, Cin, D, H, W + int(rng.integers(1, 9)), device=dev)
            pitched[..., :W] = x.to(dev)
            outs = [ops.conv3d_k3_bf16x3(ops.scl_pack(x.to(dev)), wq, dv(scale), dv(shift), relu, dv(res)),
                    ops.conv3d_k3_bf16x3(x.to(dev), wq, dv(scale), dv(shift), relu, dv(res)),
                    ops.conv3d_k3_bf16x3(pitched[..., :W], wq, dv(scale), dv(shift), relu, dv(res))]
            same = all(torch.equal(outs[0], o) for o in outs[1:])
        elif kind == "s2":
            w = torch.randn(Cout, Cin, 3, 3, 3, generator=g) / (27 * Cin) ** 0.5
            want = three_terms(F.conv3d, x, w, padding=1, stride=2)
            mag = float(F.conv3d(x.abs().double(), w.abs().double(), padding=1, stride=2).max())
            res, resid = None, False
            outs = [ops.conv3d_k3_s2_bf16x3(x.to(dev), ops.split_conv_weight(w.to(dev), 1), dv(scale), dv(shift), relu)]
            same = True
        else:
            w = torch.randn(Cin, Cout, 3, 3, 3, generator=g) / (27 * Cin / 8) ** 0.5
            kw = dict(stride=2, padding=1, output_padding=1)
            want = three_terms(F.conv_transpose3d, x, w, **kw)
            mag = float(F.conv_transpose3d(x.abs().double(), w.abs().double(), **kw).max())
            res = torch.randn(want.shape, generator=g) if resid else None
            outs = [ops.convT3d_k3_s2_bf16x3(x.to(dev), ops.split_conv_weight(w.to(dev), 2), dv(scale), dv(shift), dv(res), relu)]
            same = True
        ref = want
        if affine:
            ref = ref * scale.double().view(1, -1, 1, 1, 1) + shift.double().view(1, -1, 1, 1, 1)
        if kind == "t":          # transposed: affine, ReLU, then the skip tensor
            ref = torch.relu(ref) if relu else ref
            ref = ref + res.double() if resid else ref
        else:                    # stride 1: affine, residual, then ReLU
            ref = ref + res.double() if resid else ref
            ref = torch.relu(ref) if relu else ref
        tol = 4e-7 * mag * (float(scale.max()) if affine else 1.0) + 2e-6 * max(1.0, float(ref.abs().max())) * (affine or resid)
        err = float((outs[0].cpu().double() - ref).abs().max())
        ok = same and outs[0].shape == ref.shape and err <= tol
        if not ok:
            bad += 1
            print(f"seed {seed} {kind} N={N} Cin={Cin} Cout={Cout} {D}x{H}x{W} affine={affine} relu={relu} resid={resid}: "
                  f"err {err:.3e} tol {tol:.3e} same={same}", flush=True)
    print(f"fuzz_conv_bf16: {cases} cases from seed {first}, {bad} mismatches")
    return 1 if bad else 0


if __name__ == "__main__":
    sys.exit(main())
