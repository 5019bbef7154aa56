#!/usr/bin/env python3
"""Randomised soak of the bf16x3 convolution family (csrc/costreg_bf16.hip; not collected by pytest; run on the GPU box):
    python tests/fuzz_conv_bf16.py [cases] [first_seed]
Random shapes (ragged tiles in d, h and w, channel counts that are no multiple of 8, single-voxel volumes, 64 / 128 / 192
output channels), the three layer kinds (stride 1 on the SCL form, on the fp32 tensor and on a row-pitched view of it; stride 2;
transposed), with and without affine / ReLU / residual -- each against a float64 evaluation of the SAME three products
(x_hi*w_hi + x_hi*w_mid + x_mid*w_hi): what remains is fp32 accumulation order, bounded by 4e-7 of the summed products."""
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
        kind = ("s1", "s2", "t")[seed % 3]
        N = int(rng.integers(1, 4))
        Cin = int(rng.choice([1, 3, 8, 13, 16, 24, 40, 64, 100]))
        Cout = int(rng.choice([64, 64, 128, 192]))
        D, H, W = int(rng.integers(1, 11)), int(rng.integers(1, 30)), int(rng.integers(1, 40))
        x = torch.randn(N, Cin, D, H, W, generator=g) * float(rng.uniform(0.1, 4.0))
        affine, relu, resid = bool(rng.integers(0, 2)), bool(rng.integers(0, 2)), bool(rng.integers(0, 2))
        scale = (torch.rand(Cout, generator=g) + 0.5) if affine else None
        shift = (torch.randn(Cout, generator=g) * 0.1) if affine else None
        dv = lambda t: None if t is None else t.to(dev)   # noqa: E731
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
