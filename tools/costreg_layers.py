#!/usr/bin/env python3
"""Per-layer forward time of the cost regularisation network at a workload's shape (MIOpen, fp32)."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
import bench
from mvsdet_amd.costreg import CostRegNet3DGS
name = sys.argv[1] if len(sys.argv) > 1 else "scannet_ref_40v_12d_60x80"
w = bench.WORKLOADS[name]
dev = torch.device("cuda:0")
net = CostRegNet3DGS(w["C"]).to(dev).eval()
x = torch.randn(w["N"], w["C"], w["D"], w["H"], w["W"], device=dev)
def timed(fn, reps=3):
    fn(); torch.cuda.synchronize()
    ts = []
    for _ in range(reps):
        a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        a.record(); y = fn(); b.record(); torch.cuda.synchronize(); ts.append(a.elapsed_time(b))
    return float(np.median(ts)), y
with torch.no_grad():
    t, full = timed(lambda: net.conv0.conv(x)); print(f"conv0.conv 256->64 s1: {t:.2f} ms  ({27*2*256*64*x.shape[0]*x[0,0].numel()/t/1e9:.1f} TFLOP/s)")
    t, _ = timed(lambda: torch.relu_(net.conv0.bn(full))); print(f"conv0 bn+relu: {t:.2f} ms")
    full = net.conv0(x)
    t, h1 = timed(lambda: net.conv1(full)); print(f"conv1 64->128 s2: {t:.2f} ms")
    t, half = timed(lambda: net.conv2(h1)); print(f"conv2 128->128: {t:.2f} ms")
    t, q1 = timed(lambda: net.conv3(half)); print(f"conv3 128->256 s2: {t:.2f} ms")
    t, q = timed(lambda: net.conv4(q1)); print(f"conv4 256->256: {t:.2f} ms")
    t, u1 = timed(lambda: net.conv9(q)); print(f"conv9 deconv 256->128: {t:.2f} ms")
    half = half + u1
    t, u2 = timed(lambda: net.conv11(half)); print(f"conv11 deconv 128->64: {t:.2f} ms")
    full = full + u2
    t, _ = timed(lambda: net.prob(full)); print(f"prob 64->2: {t:.2f} ms")
