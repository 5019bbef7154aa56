#!/usr/bin/env python3
"""Per-layer forward time of the cost regularisation network on the HIP route (eval mode, no autograd)."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
import bench
from mvsdet_amd.costreg import CostRegNet3DGS
w = bench.WORKLOADS[sys.argv[1] if len(sys.argv) > 1 else "scannet_ref_40v_12d_60x80"]
dev = torch.device("cuda:0")
net = CostRegNet3DGS(w["C"]).to(dev).eval()
x = torch.randn(w["N"], w["C"], w["D"], w["H"], w["W"], device=dev)
def timed(name, fn, flop=None, reps=5):
    y = fn(); torch.cuda.synchronize()
    ts = []
    for _ in range(reps):
        a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        a.record(); y = fn(); b.record(); torch.cuda.synchronize(); ts.append(a.elapsed_time(b))
    t = float(np.median(ts))
    print(f"{name:34s} {t:7.2f} ms" + (f"  {flop / t / 1e9:6.1f} TFLOP/s" if flop else ""))
    return y
V = w["N"] * w["D"] * w["H"] * w["W"]
with torch.no_grad():
    full = timed("conv0 256->64 (+bn+relu)", lambda: net._cbr(net.conv0, x), 54 * 256 * 64 * V)
    h1 = timed("conv1 64->128 s2", lambda: net._cbr(net.conv1, full), 54 * 64 * 128 * V / 8)
    half = timed("conv2 128->128", lambda: net._cbr(net.conv2, h1), 54 * 128 * 128 * V / 8)
    q1 = timed("conv3 128->256 s2", lambda: net._cbr(net.conv3, half), 54 * 128 * 256 * V / 64)
    q = timed("conv4 256->256", lambda: net._cbr(net.conv4, q1), 54 * 256 * 256 * V / 64)
    half2 = timed("conv9 deconv 256->128 (+skip)", lambda: net._up(net.conv9, q, half), 54 * 256 * 128 * V / 64)
    full2 = timed("conv11 deconv 128->64 (+skip)", lambda: net._up(net.conv11, half2, full), 54 * 128 * 64 * V / 8)
    timed("prob 64->2", lambda: net._head(full2), 54 * 64 * 2 * V)
    timed("whole network", lambda: net(x), CostRegNet3DGS.flops(w["N"], w["D"], w["H"], w["W"], w["C"]))
