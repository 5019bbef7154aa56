#!/usr/bin/env python3
"""IndoorImVoxelNeck at the shipped configuration (256 -> 128, n_blocks [1,1,1], 40x40x16 volume): the MFMA route against
the framework's layers on the same device (MIOpen; its first call includes the kernel search)."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from mvsdet_amd.neck import IndoorImVoxelNeck
dev = torch.device("cuda:0")
torch.manual_seed(0)
m = IndoorImVoxelNeck(256, 128, [1, 1, 1]).to(dev).eval()
x = torch.randn(1, 256, 40, 40, 16, device=dev)
def timed(fn, reps=5):
    fn(); torch.cuda.synchronize()
    ts = []
    for _ in range(reps):
        a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        a.record(); fn(); b.record(); torch.cuda.synchronize(); ts.append(a.elapsed_time(b))
    return min(ts)
tfl = IndoorImVoxelNeck.flops(1, [40, 40, 16]) / 1e12
with torch.no_grad():
    t = timed(lambda: m(x))
print(f"neck MFMA route: {t:.3f} ms = {tfl / t * 1e3:.1f} TFLOP/s ({tfl / t * 1e3 / 157.3:.3f} of the fp32 matrix peak)", flush=True)
if len(sys.argv) > 1 and sys.argv[1] == "miopen":
    t0 = time.time()
    with torch.enable_grad():   # grad mode: the module takes the framework's layers
        xr = x.clone().requires_grad_(False)
        m.train(False)
        def run():
            with torch.enable_grad():
                return m(xr.requires_grad_(True))
        t2 = timed(run, 3)
    print(f"neck framework layers (MIOpen fp32): {t2:.3f} ms = {tfl / t2 * 1e3:.1f} TFLOP/s (first call + search {time.time() - t0:.1f} s)")
