#!/usr/bin/env python3
"""Per-layer HIP-event times of the cost regularisation network's eval route at the reference-true shape (GPU box):

    python tools/costreg_layers_timing.py [reps] [option=value ...]

Every operator call of `CostRegNet3DGS.forward` is bracketed by two events on the current stream; `option=value` pairs are
passed to mvsdet_set_option first (tuning knobs of the library), so variants can be compared in one process."""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np  # noqa: E402
import torch  # noqa: E402

from mvsdet_amd import _lib, ops  # noqa: E402
from mvsdet_amd.costreg import CostRegNet3DGS  # noqa: E402

NAMES = ["conv3d_k3_bf16x3", "conv3d_k3_s2_bf16x3", "convT3d_k3_s2_bf16x3", "conv3d_k3_cout2", "conv3d_k3_cout2_sum", "scl_pack", "split_conv_weight"]


def main():
    reps = int(sys.argv[1]) if len(sys.argv) > 1 and sys.argv[1].isdigit() else 5
    for kv in sys.argv[1:]:
        if "=" in kv:
            k, v = kv.split("=")
            _lib.check(_lib.load().mvsdet_set_option(k.encode(), int(v)), "set_option")
    dev = torch.device("cuda:0")
    torch.manual_seed(0)
    net = CostRegNet3DGS(256).to(dev).eval()
    net.view_streams = 1   # per-kernel times: one batch on one stream (two halves on two streams overlap their kernels)
    x = torch.rand(40, 256, 12, 60, 80, device=dev)
    log = []
    real = {n: getattr(ops, n) for n in NAMES}

    def wrap(name):
        def f(*a, **k):
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            out = real[name](*a, **k)
            e1.record()
            log.append((name, e0, e1))
            return out
        return f

    for n in NAMES:
        setattr(ops, n, wrap(n))
    with torch.no_grad():
        net(x)
        torch.cuda.synchronize()
        log.clear()
        t0, t1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        t0.record()
        for _ in range(reps):
            y = net(x)
        t1.record()
        torch.cuda.synchronize()
    per = len(log) // reps
    rows = []
    for i in range(per):
        ts = [log[r * per + i][1].elapsed_time(log[r * per + i][2]) for r in range(reps)]
        rows.append((log[i][0], float(np.median(ts)), float(np.min(ts))))
    labels = iter(["conv0", "conv1", "conv2", "conv3", "conv4", "conv9", "conv11"])
    tot = 0.0
    for name, med, mn in rows:
        lab = next(labels) if name in NAMES[:3] else ("head" if name.startswith("conv3d_k3_cout2") else "")
        if name != "split_conv_weight":
            print(f"{lab:7s} {name:24s} median {med:7.3f} ms   min {mn:7.3f} ms")
        tot += med
    sw = sum(m for n, m, _ in rows if n == "split_conv_weight")
    print(f"weight splitting ({sum(1 for n, _, _ in rows if n == 'split_conv_weight')} launches) {sw:.3f} ms;  sum of the operators {tot:.3f} ms;  "
          f"network {t0.elapsed_time(t1) / reps:.3f} ms;  checksum {float(y.double().abs().sum()):.6f}")


if __name__ == "__main__":
    main()
