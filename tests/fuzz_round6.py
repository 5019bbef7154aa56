#!/usr/bin/env python3
"""Randomised soak of round 6's kernels (not collected by pytest; run on the GPU box):  python tests/fuzz_round6.py [cases] [first_seed]
  * conv3d_k3_fp16mx (csrc/costreg_mx.h: fp16 + block-scaled FP6, wave-specialised and hand-pipelined): random shapes (ragged tiles, channel
    counts that are no multiple of 8, 1 .. 300 input channels, 64 / 128 output channels, strided views, value ranges from 1e-3 to 1e4), all
    three forms of the kernel bit for bit against each other, and against a float64 convolution within 2^-13 of the summed |products|
    (the scheme's own error: two 6-bit correction terms on 11-bit main pieces);
  * the persistent transposed kernel (csrc/convt_persist.h) bit for bit against the per-tile kernel: random shapes and block counts."""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np  # noqa: E402
import torch  # noqa: E402
import torch.nn.functional as F  # noqa: E402

from mvsdet_amd import _lib, ops  # noqa: E402


def main():
    cases = int(sys.argv[1]) if len(sys.argv) > 1 else 40
    first = int(sys.argv[2]) if len(sys.argv) > 2 else 0
    dev = torch.device("cuda:0")
    bad = 0
    for seed in range(first, first + cases):
        rng = np.random.default_rng(60000 + seed)
        g = torch.Generator().manual_seed(60000 + seed)
        if seed % 2 == 0:
            N, Cin, Cout = int(rng.integers(1, 4)), int(rng.choice([1, 5, 8, 16, 20, 24, 33, 64, 100, 256, 300])), int(rng.choice([64, 128]))
            D, H, W = int(rng.integers(1, 10)), int(rng.integers(1, 28)), int(rng.integers(1, 40))
            mag = float(10.0 ** rng.uniform(-3, 4))
            x = (torch.rand(N, Cin, D, H, W + 3, generator=g) * mag).to(dev)[..., 1:W + 1]   # a strided view (rows pitched)
            w = (torch.randn(Cout, Cin, 3, 3, 3, generator=g) / (27 * Cin) ** 0.5).to(dev)
            affine, relu = bool(rng.integers(0, 2)), bool(rng.integers(0, 2))
            sc = (torch.rand(Cout, generator=g) + 0.5).to(dev) if affine else None
            sh = (torch.randn(Cout, generator=g) * 0.1 * mag).to(dev) if affine else None
            wmx = ops.split_conv_weight_mx(w)
            outs = []
            try:
                for form in (0, 8, 12):
                    _lib.set_option("conv_mx_th", form)
                    outs.append(ops.conv3d_k3_fp16mx(x, wmx, sc, sh, relu))
            finally:
                _lib.set_option("conv_mx_th", 0)
            same = all(torch.equal(outs[0], o) for o in outs[1:])
            ref = F.conv3d(x.double(), w.double(), padding=1)
            mags = F.conv3d(x.double().abs(), w.double().abs(), padding=1)
            if affine:
                ref = ref * sc.double().view(1, -1, 1, 1, 1) + sh.double().view(1, -1, 1, 1, 1)
                mags = mags * sc.double().view(1, -1, 1, 1, 1)
            if relu:
                ref = ref.clamp_min(0)
            err = float(((outs[0].double() - ref).abs() / (mags + 1e-300)).max())
            ok = same and err <= 2.0 ** -13
            print(f"seed {seed}: fp16mx N={N} Cin={Cin} Cout={Cout} {D}x{H}x{W} |x|<={mag:.3g} affine={affine} relu={relu}: forms "
                  f"{'equal' if same else 'DIFFER'}, error {err:.2e} of the summed |products| {'ok' if ok else 'BAD'}", flush=True)
        else:
            N, Cin, Cout = int(rng.integers(1, 4)), int(rng.choice([96, 128, 160, 256])), int(rng.choice([64, 128, 192]))
            D, H, W = int(rng.integers(1, 8)), int(rng.integers(1, 34)), int(rng.integers(1, 44))
            out = ("f32", "scl")[int(rng.integers(0, 2))]
            x = torch.randn(N, Cin, D, H, W, generator=g).to(dev)
            wq = ops.split_conv_weight((torch.randn(Cin, Cout, 3, 3, 3, generator=g) / (27 * Cin / 8) ** 0.5).to(dev), 2)
            affine, relu = bool(rng.integers(0, 2)), bool(rng.integers(0, 2))
            sc = (torch.rand(Cout, generator=g) + 0.5).to(dev) if affine else None
            sh = (torch.randn(Cout, generator=g) * 0.1).to(dev) if affine else None
            res = torch.randn(N, Cout, 2 * D, 2 * H, 2 * W, generator=g).to(dev)
            xs = ops.scl_pack(x)
            got = []
            blocks = (0, 1, int(rng.choice([8, 16, 64, 200, 512, 2048])))
            try:
                for persist in blocks:
                    _lib.set_option("convT_persist", persist)
                    y = ops.convT3d_k3_s2_bf16x3(xs, wq, sc, sh, res, relu, outputs=(out,))
                    got.append(y.clone() if out == "f32" else y.data.view(torch.int16).clone())
            finally:
                _lib.set_option("convT_persist", 0)
            ok = all(torch.equal(got[0], o) for o in got[1:]) and float(got[0].float().abs().max()) > 0
            print(f"seed {seed}: persistent convT N={N} Cin={Cin} Cout={Cout} {D}x{H}x{W} -> {out} affine={affine} relu={relu} blocks {blocks}: "
                  f"{'equal bits' if ok else 'BAD'}", flush=True)
        bad += not ok
    torch.cuda.synchronize()
    print(f"{cases} cases, {bad} bad")
    sys.exit(1 if bad else 0)


if __name__ == "__main__":
    main()
