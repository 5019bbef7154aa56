#!/usr/bin/env python3
"""Training step (forward + backward) of the cost regularisation network at the reference-true shape: our forward /
dX / dW kernels for the stride-1 convolutions against the all-MIOpen route."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from mvsdet_amd.costreg import CostRegNet3DGS
dev = torch.device("cuda:0")
net = CostRegNet3DGS(256).to(dev).train()
x = torch.randn(40, 256, 12, 60, 80, device=dev, requires_grad=True)
def step():
    net.zero_grad(set_to_none=True); x.grad = None
    net(x).sum().backward()
for flag, prec in ((True, "bf16x3"), (True, "fp32"), (False, "fp32")):
    if not flag and len(sys.argv) < 2:
        continue   # the all-MIOpen route takes ~1 s per step and minutes of kernel search: `... miopen` to include it
    net.hip_backward, net.matrix_precision = flag, prec
    step(); torch.cuda.synchronize()
    ts = []
    for _ in range(3):
        a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        a.record(); step(); b.record(); torch.cuda.synchronize(); ts.append(a.elapsed_time(b))
    print(f"cost network training step, hip_backward={flag}, forward / dX on {prec}: {min(ts):.1f} ms", flush=True)
