#!/usr/bin/env python3
"""Input gradient of conv0 (a 64 -> 256 stride-1 convolution of grad_out with the flipped weights): reading grad_out as fp32 in
place (every one of the four output-channel blocks of a tile cuts the same values again) against one scl_pack pass + the DMA-fed
form.  GPU box."""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch  # noqa: E402

from mvsdet_amd import ops  # noqa: E402

dev = torch.device("cuda:0")
gy = torch.randn(40, 64, 12, 60, 80, device=dev)
w = torch.randn(256, 64, 3, 3, 3, device=dev) / 40
wq = ops.split_conv_weight(w)


def timed(fn, reps=6):
    for _ in range(2):
        fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps):
        out = fn()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / reps, out


buf = [None]


def packed():
    buf[0] = ops.scl_pack(gy, out=buf[0])
    return ops.conv3d_k3_bf16x3(buf[0], wq, None, None, False)


for _ in range(2):
    t1, a = timed(lambda: ops.conv3d_k3_bf16x3(gy, wq, None, None, False))
    t2, b = timed(packed)
    print(f"fp32 in place {t1:.3f} ms   scl_pack + DMA-fed {t2:.3f} ms   equal bits: {bool(torch.equal(a, b))}", flush=True)
