#!/usr/bin/env python3
"""Head 64 -> 2 of the cost network at the reference-true shape: one input, and the sum of two (GPU box)."""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa: E402

from mvsdet_amd import ops  # noqa: E402

dev = torch.device("cuda:0")
x, y = torch.randn(40, 64, 12, 60, 80, device=dev), torch.randn(40, 64, 12, 60, 80, device=dev)
w, b = torch.randn(2, 64, 3, 3, 3, device=dev) / 40, torch.randn(2, device=dev)


def timed(fn, reps=10):
    fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps):
        fn()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / reps


t1 = timed(lambda: ops.conv3d_k3_cout2_sum(x, None, w, b))
t2 = timed(lambda: ops.conv3d_k3_cout2_sum(x, y, w, b))
gb = x.numel() * 4 / 1e9
print(f"head, one input {t1:.3f} ms ({gb / t1:.2f} TB/s of input)   two inputs {t2:.3f} ms ({2 * gb / t2:.2f} TB/s)")
