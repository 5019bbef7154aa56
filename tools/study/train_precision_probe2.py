#!/usr/bin/env python3
"""Second probe: every convolution operator call of a training step (forward, dX, dW), fp32 route against bf16x3 route, in call
order: how far apart are the inputs, how far apart the outputs, and how far is the bf16x3 operator from the fp32 operator ON THE
bf16x3 ROUTE'S OWN INPUTS (the operator's error by itself)."""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests", "golden"))
import numpy as np  # noqa: E402
import torch  # noqa: E402
from lcg import lcg_fill_state, lcg_uniform  # noqa: E402

from mvsdet_amd import ops  # noqa: E402
from mvsdet_amd.costreg import CostRegNet3DGS  # noqa: E402

dev = torch.device("cuda:0")
shape = (2, 256, 8, 24, 32)
NAMES = ["conv3d_k3_bf16x3", "conv3d_k3_s2_bf16x3", "convT3d_k3_s2_bf16x3", "conv3d_k3_mfma", "convT3d_k3_s2_mfma", "conv3d_k3_dw",
         "bn3d_relu_train", "bn3d_relu_backward", "conv3d_k3_cout2", "conv3d_k3_cout2_backward"]
real = {n: getattr(ops, n) for n in NAMES}
log = []


def wrap(name):
    def f(*a, **k):
        out = real[name](*a, **k)
        first = a[0].unpack() if hasattr(a[0], "unpack") else a[0]
        second = a[1] if len(a) > 1 and torch.is_tensor(a[1]) and a[1].dtype == torch.float32 else None
        o = out[0] if isinstance(out, tuple) else out
        log.append((name, first.detach().clone() if torch.is_tensor(first) else None,
                    second.detach().clone() if second is not None else None, o.detach().clone()))
        return out
    return f


for n in NAMES:
    setattr(ops, n, wrap(n))


def run(prec):
    log.clear()
    net = CostRegNet3DGS(256, 64).train()
    net.matrix_precision = prec
    with torch.no_grad():
        lcg_fill_state(net, 12)
    net = net.to(dev)
    x = torch.from_numpy(lcg_uniform(int(np.prod(shape)), 120)).reshape(shape).abs().to(dev).requires_grad_(True)
    y = net(x)
    R = torch.from_numpy(lcg_uniform(y.numel(), 121)).reshape(y.shape).to(dev)
    (y * R).sum().backward()
    return list(log)


rel = lambda a, b: float((a.double() - b.double()).norm() / b.double().norm().clamp_min(1e-300))  # noqa: E731
l0 = run("fp32")
l1 = run("bf16x3")
print(len(l0), len(l1))
for i, (c0, c1) in enumerate(zip(l0, l1)):
    n0, a0, b0, o0 = c0
    n1, a1, b1, o1 = c1
    ra = rel(a1, a0) if a0 is not None and a1 is not None and a0.shape == a1.shape else float("nan")
    rb = rel(b1, b0) if b0 is not None and b1 is not None and b0.shape == b1.shape else float("nan")
    ro = rel(o1, o0) if o0.shape == o1.shape else float("nan")
    print(f"{i:3d} {n0:24s} | {n1:24s} in0 {tuple(a0.shape) if a0 is not None else None} rel-in0 {ra:.2e} rel-in1 {rb:.2e} rel-out {ro:.2e}")
