"""The transposed layers of the cost network on the persistent kernel (csrc/convt_persist.h) against the one-block-per-(tile, 32
channels) kernel: the same bits, and the time of both, at conv9 / conv11 of the reference-true shape (40 views) and at a ragged shape.
Run on the GPU box: python tools/study/r06_convt_persist.py"""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from mvsdet_amd import _lib, ops  # noqa: E402


def run(tag, N, Cin, Cout, D, H, W, outputs, reps=20):
    dev = torch.device("cuda")
    g = torch.Generator(device="cpu").manual_seed(5)
    x = torch.randn((N, Cin, D, H, W), generator=g).to(dev)
    w = (torch.randn((Cin, Cout, 3, 3, 3), generator=g) * 0.05).to(dev)
    sc = (torch.rand(Cout, generator=g) + 0.5).to(dev)
    sh = torch.randn(Cout, generator=g).to(dev)
    res = torch.randn((N, Cout, 2 * D, 2 * H, 2 * W), generator=g).to(dev)
    xs = ops.scl_pack(x)
    wq = ops.split_conv_weight(w, 2)
    got = {}
    for mode in (0, 1):
        _lib.set_option("convT_persist", mode)
        y = ops.convT3d_k3_s2_bf16x3(xs, wq, sc, sh, res, True, outputs=outputs)
        torch.cuda.synchronize()
        ev = [torch.cuda.Event(enable_timing=True) for _ in range(reps + 1)]
        for i in range(reps):
            ev[i].record()
            ops.convT3d_k3_s2_bf16x3(xs, wq, sc, sh, res, True, outputs=outputs, scl_out=y if "scl" in outputs else None)
        ev[reps].record()
        torch.cuda.synchronize()
        ts = sorted(ev[i].elapsed_time(ev[i + 1]) for i in range(reps))
        got[mode] = (y, ts[0], ts[len(ts) // 2])
    _lib.set_option("convT_persist", 1)
    a, b = got[0][0], got[1][0]
    ta = a if isinstance(a, torch.Tensor) else a.data.view(torch.int16)     # the whole SCL buffer, zero border included
    tb = b if isinstance(b, torch.Tensor) else b.data.view(torch.int16)
    same = torch.equal(ta, tb)
    nz = float((ta != 0).float().mean())
    print(f"{tag}: N={N} Cin={Cin} Cout={Cout} {D}x{H}x{W} -> {outputs}: bits {'EQUAL' if same else 'DIFFER'}"
          f" | per-tile kernel min {got[0][1]:.3f} median {got[0][2]:.3f} ms | persistent min {got[1][1]:.3f} median {got[1][2]:.3f} ms (nonzero share {nz:.2f})",
          flush=True)
    if not same:
        d = (ta.float() - tb.float()).abs()
        print("   max |d|", float(d.max()), "differing share", float((d > 0).float().mean()))
    return same


def main():
    ok = True
    ok &= run("ragged", 3, 128, 64, 5, 13, 21, ("f32",), reps=5)
    ok &= run("ragged scl", 2, 96, 128, 4, 20, 9, ("scl",), reps=5)
    ok &= run("conv11", 40, 128, 64, 6, 30, 40, ("f32",))
    ok &= run("conv9", 40, 256, 128, 3, 15, 20, ("scl",))
    print("ALL EQUAL" if ok else "MISMATCH")
    sys.exit(0 if ok else 1)


if __name__ == "__main__":
    main()
