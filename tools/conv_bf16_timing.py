"""Timing of the bf16x3 convolution against the fp32-MFMA one at the cost network's shapes (GPU box):
python tools/conv_bf16_timing.py"""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from mvsdet_amd import ops  # noqa: E402


def timeit(fn, reps=5):
    fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps):
        fn()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / reps


def main():
    dev = torch.device("cuda:0")
    shapes = [("conv0", 40, 256, 64, 12, 60, 80), ("conv2", 40, 128, 128, 6, 30, 40), ("conv4", 40, 256, 256, 3, 15, 20)]
    for name, N, Cin, Cout, D, H, W in shapes:
        x = torch.randn(N, Cin, D, H, W, device=dev).abs()
        w = torch.randn(Cout, Cin, 3, 3, 3, device=dev) / (27 * Cin) ** 0.5
        sc, sh = torch.ones(Cout, device=dev), torch.zeros(Cout, device=dev)
        wp = ops.permute_conv_weight(w)
        wq = ops.split_conv_weight(w)
        xs = ops.scl_pack(x)
        t32 = timeit(lambda: ops.conv3d_k3_mfma(x, wp, sc, sh, True, 1))
        tpk = timeit(lambda: ops.scl_pack(x, out=xs))
        tbf = timeit(lambda: ops.conv3d_k3_bf16x3(xs, wq, sc, sh, True))
        tdir = timeit(lambda: ops.conv3d_k3_bf16x3(x, wq, sc, sh, True))
        fl = 2.0 * 27 * Cin * Cout * N * D * H * W
        y32 = ops.conv3d_k3_mfma(x, wp, sc, sh, True, 1)
        ybf = ops.conv3d_k3_bf16x3(xs, wq, sc, sh, True)
        err = float((y32 - ybf).abs().max()) / float(y32.abs().max())
        print(f"{name}: fp32 MFMA {t32:.3f} ms ({fl / t32 / 1e9:.1f} TF)  bf16x3 {tbf:.3f} ms ({fl / tbf / 1e9:.1f} TF useful, "
              f"{3 * fl / tbf / 1e9:.0f} TF of bf16 MFMA)  pack {tpk:.3f} ms  fp32-input form {tdir:.3f} ms  max rel diff {err:.2e}", flush=True)


def stride2():
    dev = torch.device("cuda:0")
    for name, N, Cin, Cout, D, H, W in (("conv1", 40, 64, 128, 12, 60, 80), ("conv3", 40, 128, 256, 6, 30, 40)):
        x = torch.randn(N, Cin, D, H, W, device=dev).abs()
        w = torch.randn(Cout, Cin, 3, 3, 3, device=dev) / (27 * Cin) ** 0.5
        sc, sh = torch.ones(Cout, device=dev), torch.zeros(Cout, device=dev)
        wp = ops.permute_conv_weight(w)
        wq = ops.split_conv_weight(w, 1)
        t32 = timeit(lambda: ops.conv3d_k3_mfma(x, wp, sc, sh, True, 2))
        tbf = timeit(lambda: ops.conv3d_k3_s2_bf16x3(x, wq, sc, sh, True))
        y32 = ops.conv3d_k3_mfma(x, wp, sc, sh, True, 2)
        ybf = ops.conv3d_k3_s2_bf16x3(x, wq, sc, sh, True)
        fl = 2.0 * 27 * Cin * Cout * y32[:, 0].numel()
        err = float((y32 - ybf).abs().max()) / float(y32.abs().max())
        print(f"{name} (stride 2): fp32 MFMA {t32:.3f} ms ({fl / t32 / 1e9:.1f} TF)  bf16x3 {tbf:.3f} ms ({fl / tbf / 1e9:.1f} TF useful)  "
              f"max rel diff {err:.2e}", flush=True)


def transposed():
    dev = torch.device("cuda:0")
    for name, N, Cin, Cout, D, H, W in (("conv9", 40, 256, 128, 3, 15, 20), ("conv11", 40, 128, 64, 6, 30, 40)):
        x = torch.randn(N, Cin, D, H, W, device=dev).abs()
        w = torch.randn(Cin, Cout, 3, 3, 3, device=dev) / (27 * Cin / 8) ** 0.5
        skip = torch.randn(N, Cout, 2 * D, 2 * H, 2 * W, device=dev)
        sc, sh = torch.ones(Cout, device=dev), torch.zeros(Cout, device=dev)
        wp = ops.permute_convT_weight(w)
        wq = ops.split_conv_weight(w, 2)
        t32 = timeit(lambda: ops.convT3d_k3_s2_mfma(x, wp, sc, sh, skip, True))
        tbf = timeit(lambda: ops.convT3d_k3_s2_bf16x3(x, wq, sc, sh, skip, True))
        y32 = ops.convT3d_k3_s2_mfma(x, wp, sc, sh, skip, True)
        ybf = ops.convT3d_k3_s2_bf16x3(x, wq, sc, sh, skip, True)
        fl = 2.0 * 27 * Cin * Cout * x[:, 0].numel()
        err = float((y32 - ybf).abs().max()) / float(y32.abs().max())
        print(f"{name} (transposed): fp32 MFMA {t32:.3f} ms ({fl / t32 / 1e9:.1f} TF)  bf16x3 incl. packing {tbf:.3f} ms "
              f"({fl / tbf / 1e9:.1f} TF useful)  max rel diff {err:.2e}", flush=True)


if __name__ == "__main__":
    transposed()
    stride2()
    main()
