#!/usr/bin/env python3
"""Backward of the head 64 -> 2 at the reference-true shape: input gradient and weight gradient together (GPU box)."""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa: E402

from mvsdet_amd import ops  # noqa: E402

dev = torch.device("cuda:0")
x, gy = torch.randn(40, 64, 12, 60, 80, device=dev), torch.randn(40, 2, 12, 60, 80, device=dev)
w = torch.randn(2, 64, 3, 3, 3, device=dev) / 40
out = {}
for bf in (False, True, False, True):
    for _ in range(2):
        ops.conv3d_k3_cout2_backward(x, w, gy, 32, bf)
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(10):
        out[bf] = ops.conv3d_k3_cout2_backward(x, w, gy, 32, bf)
    e1.record()
    torch.cuda.synchronize()
    print(f"head backward (dX + dW{' on bf16x3' if bf else ' fp32'}): {e0.elapsed_time(e1) / 10:.3f} ms", flush=True)
d = float((out[True][1] - out[False][1]).abs().max() / out[False][1].abs().max())
print(f"max |dW bf16x3 - dW fp32| / max |dW| = {d:.2e}", flush=True)
