#!/usr/bin/env python3
"""Forward time of the cost regularisation network (mvsdet_amd.costreg, MIOpen convolutions) on one scene's variance
volume, next to the hot path that surrounds it.  Usage: python tools/costreg_timing.py [workload] [views_per_chunk]"""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np  # noqa: E402
import torch  # noqa: E402

import bench  # noqa: E402
from mvsdet_amd.costreg import CostRegNet3DGS  # noqa: E402

name = sys.argv[1] if len(sys.argv) > 1 else "scannet_ref_40v_12d_60x80"
chunk = int(sys.argv[2]) if len(sys.argv) > 2 else 0
w = bench.WORKLOADS[name]
dev = torch.device("cuda:0")
n = chunk or w["N"]
x = torch.randn(n, w["C"], w["D"], w["H"], w["W"], device=dev)
net = CostRegNet3DGS(w["C"]).to(dev).eval()
flop = CostRegNet3DGS.flops(n, w["D"], w["H"], w["W"], w["C"])


def timed(fn, reps=3):
    fn(); torch.cuda.synchronize()
    ts = []
    for _ in range(reps):
        a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        a.record(); fn(); b.record(); torch.cuda.synchronize()
        ts.append(a.elapsed_time(b))
    return float(np.median(ts))


with torch.no_grad():
    t = timed(lambda: net(x))
    print(f"{name} ({n} views): fp32 NCDHW {t:.1f} ms = {flop / t / 1e9:.1f} TFLOP/s ({flop / 1e12:.2f} TFLOP)")
    xc = x.contiguous(memory_format=torch.channels_last_3d)
    netc = net.to(memory_format=torch.channels_last_3d)
    t = timed(lambda: netc(xc))
    print(f"{name}: fp32 channels_last_3d {t:.1f} ms = {flop / t / 1e9:.1f} TFLOP/s")
    with torch.autocast("cuda", dtype=torch.bfloat16):
        t = timed(lambda: netc(xc))
    print(f"{name}: bf16 autocast channels_last_3d {t:.1f} ms = {flop / t / 1e9:.1f} TFLOP/s  (outside the 1e-4 tolerance)")
