#!/usr/bin/env python3
"""Top GPU kernels of one eval forward of IndoorImVoxelNeck at the shipped configuration (torch profiler)."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from torch.profiler import profile, ProfilerActivity
from mvsdet_amd.neck import IndoorImVoxelNeck
dev = torch.device("cuda:0")
torch.manual_seed(0)
m = IndoorImVoxelNeck(256, 128, [1, 1, 1]).to(dev).eval()
x = torch.randn(int(sys.argv[1]) if len(sys.argv) > 1 else 1, 256, 40, 40, 16, device=dev)
with torch.no_grad():
    for _ in range(3):
        m(x)
    torch.cuda.synchronize()
    with profile(activities=[ProfilerActivity.CUDA]) as prof:
        m(x); torch.cuda.synchronize()
rows = sorted(prof.key_averages(), key=lambda r: -r.device_time_total)[:24]
tot = sum(r.device_time_total for r in prof.key_averages())
print(f"total device time {tot / 1e3:.2f} ms")
for r in rows:
    print(f"{r.device_time_total / 1e3:8.3f} ms  x{r.count:<3d} {r.key[:150]}")
