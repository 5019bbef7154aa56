#!/usr/bin/env python3
"""Top GPU kernels of one training step (forward + backward) of the cost regularisation network, torch profiler."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from torch.profiler import profile, ProfilerActivity
from mvsdet_amd.costreg import CostRegNet3DGS
dev = torch.device("cuda:0")
net = CostRegNet3DGS(256).to(dev).train()
x = torch.randn(40, 256, 12, 60, 80, device=dev, requires_grad=True)
def step():
    net.zero_grad(set_to_none=True); x.grad = None
    net(x).sum().backward()
for _ in range(2):
    step()
torch.cuda.synchronize()
with profile(activities=[ProfilerActivity.CUDA]) as prof:
    step(); torch.cuda.synchronize()
rows = sorted(prof.key_averages(), key=lambda r: -r.device_time_total)[:28]
tot = sum(r.device_time_total for r in prof.key_averages())
print(f"total device time {tot / 1e3:.1f} ms")
for r in rows:
    print(f"{r.device_time_total / 1e3:8.2f} ms  x{r.count:<3d} {r.key[:100]}")
