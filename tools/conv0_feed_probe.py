#!/usr/bin/env python3
"""What would conv0 gain if the sweep handed it the variance already cut into bf16 pieces (SCL form)?  (GPU box)
conv0 of the cost network at the reference-true shape, fp32-input form (reads the sweep's fp32 volume, cuts in the kernel)
against the SCL-input form (LDS-DMA) on a pre-packed copy, both with the chain's outputs (fp32 skip + PSCL) -- the difference is
the most a co-designed sweep -> conv0 format can save on the convolution's side."""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa: E402

from mvsdet_amd import ops  # noqa: E402

dev = torch.device("cuda:0")
x = torch.rand(40, 256, 12, 60, 80, device=dev)
w = torch.randn(64, 256, 3, 3, 3, device=dev) / 80
sc, sh = torch.ones(64, device=dev), torch.zeros(64, device=dev)
wq = ops.split_conv_weight(w)
xs = ops.scl_pack(x)


def timed(fn, reps=8):
    fn()
    torch.cuda.synchronize()
    ts = []
    for _ in range(reps):
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        fn()
        e1.record()
        torch.cuda.synchronize()
        ts.append(e0.elapsed_time(e1))
    ts.sort()
    return ts[len(ts) // 2], ts[0]


scl_o, pscl_o = ops.scl_empty((40, 64, 12, 60, 80), dev), ops.pscl_empty((40, 64, 12, 60, 80), dev)
for name, inp in (("fp32 input (cut in the kernel)", x), ("SCL input (LDS-DMA)", xs)):
    a = timed(lambda: ops.conv3d_k3_bf16x3(inp, wq, sc, sh, True, outputs=("f32", "pscl"), pscl_out=pscl_o))
    b = timed(lambda: ops.conv3d_k3_bf16x3(inp, wq, sc, sh, True, outputs=("f32",)))
    print(f"conv0, {name:32s}: outputs fp32 + PSCL median {a[0]:.3f} ms (min {a[1]:.3f});  fp32 only {b[0]:.3f} ms (min {b[1]:.3f})")
p = timed(lambda: ops.scl_pack(x, out=xs))
print(f"scl_pack of the variance volume (what a separate packing pass costs): {p[0]:.3f} ms")
