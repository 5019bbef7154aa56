#!/usr/bin/env python3
"""The cost regularisation network's eval route as the product runs it (the chain of SCL / PSCL forms, mvsdet_amd.costreg
`_forward_chain`), 12 forward passes at a BASELINE workload: target of rocprofv3 --kernel-trace --stats / --pmc
(tools/collect_costreg_profiles.sh).  Per-layer HIP-event times: tools/costreg_layers_timing.py."""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa: E402

import bench  # noqa: E402
from mvsdet_amd.costreg import CostRegNet3DGS  # noqa: E402

w = bench.WORKLOADS[sys.argv[1] if len(sys.argv) > 1 else "scannet_ref_40v_12d_60x80"]
reps = int(sys.argv[2]) if len(sys.argv) > 2 else 12
dev = torch.device("cuda:0")
torch.manual_seed(0)
net = CostRegNet3DGS(w["C"]).to(dev).eval()
net.view_streams = 1   # per-kernel times: one batch on one stream (two halves on two streams overlap their kernels)
x = torch.rand(w["N"], w["C"], w["D"], w["H"], w["W"], device=dev)
with torch.no_grad():
    net(x)
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps):
        y = net(x)
    e1.record()
    torch.cuda.synchronize()
flop = CostRegNet3DGS.flops(w["N"], w["D"], w["H"], w["W"], w["C"])
t = e0.elapsed_time(e1) / reps
print(f"whole network (layer forms: {net.layer_forms}, {net.matrix_precision}): {t:.3f} ms = {flop / t / 1e9:.1f} TFLOP/s useful; "
      f"checksum {float(y.double().abs().sum()):.6f}")
