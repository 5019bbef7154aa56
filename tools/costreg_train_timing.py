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
for flag in (True, False):
    net.hip_backward = flag
    step(); torch.cuda.synchronize()
    ts = []
    for _ in range(3):
        a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        a.record(); step(); b.record(); torch.cuda.synchronize(); ts.append(a.elapsed_time(b))
    print(f"cost network training step, hip_backward={flag}: {min(ts):.1f} ms")
