#!/usr/bin/env python3
"""Where do the gradients of the cost network's bf16x3 training route leave its fp32 route?  (GPU box)
Per layer: forward activations and the gradient arriving at each layer's output, fp32 route against bf16x3 route, on the G12
fixture's weights and input; then the parameter gradients of both against the reference's (tests/golden/g12)."""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests", "golden"))
import numpy as np  # noqa: E402
import torch  # noqa: E402
from lcg import lcg_fill_state, lcg_uniform  # noqa: E402

from mvsdet_amd.costreg import CostRegNet3DGS  # noqa: E402

g = np.load(os.path.join(ROOT, "tests", "golden", "g12_cost_regularisation_grads.npz"))
dev = torch.device("cuda:0")
shape = tuple(int(v) for v in g["in_shape"]) if len(sys.argv) < 2 else tuple(int(v) for v in sys.argv[1].split(","))


def run(prec):
    net = CostRegNet3DGS(256, 64).train()
    net.matrix_precision = prec
    with torch.no_grad():
        lcg_fill_state(net, int(g["weight_seed"]))
    net = net.to(dev)
    acts, grads = {}, {}
    for name in ("conv0", "conv1", "conv2", "conv3", "conv4", "conv9", "conv11"):
        def hook(mod, inp, out, name=name):
            acts[name] = out.detach().clone()
            out.register_hook(lambda gr, name=name: grads.__setitem__(name, gr.detach().clone()))
        getattr(net, name).register_forward_hook(hook)
    x = torch.from_numpy(lcg_uniform(int(np.prod(shape)), int(g["input_seed"]))).reshape(shape).abs().to(dev).requires_grad_(True)
    # forward hooks only fire on the framework route; call the package's own route and record by wrapping _cbr / _up
    cbr, up = net._cbr, net._up
    names = iter(["conv0", "conv1", "conv2", "conv3", "conv4"])
    unames = iter(["conv9", "conv11"])

    def cbr_w(layer, t):
        out = cbr(layer, t)
        n = next(names)
        acts[n] = out.detach().clone()
        out.register_hook(lambda gr, n=n: grads.__setitem__(n, gr.detach().clone()))
        return out

    def up_w(seq, t, skip):
        out = up(seq, t, skip)
        n = next(unames)
        acts[n] = out.detach().clone()
        out.register_hook(lambda gr, n=n: grads.__setitem__(n, gr.detach().clone()))
        return out

    net._cbr, net._up = cbr_w, up_w
    y = net(x)
    R = torch.from_numpy(lcg_uniform(y.numel(), int(g["r_seed"]))).reshape(y.shape).to(dev)
    (y * R).sum().backward()
    return net, x, y.detach(), acts, grads


rel = lambda a, b: float((a.double() - b.double()).norm() / b.double().norm().clamp_min(1e-300))  # noqa: E731
n0, x0, y0, a0, g0 = run("fp32")
n1, x1, y1, a1, g1 = run("bf16x3")
print("logits rel", rel(y1, y0), "max abs", float((y1 - y0).abs().max()))
for k in a0:
    flips = int(((a0[k] > 0) != (a1[k] > 0)).sum())
    print(f"{k:7s} forward rel {rel(a1[k], a0[k]):.2e}  sign flips {flips:5d} of {a0[k].numel():8d}   grad-at-output rel {rel(g1[k], g0[k]):.2e}")
print("input gradient rel", rel(x1.grad, x0.grad))
p0, p1 = dict(n0.named_parameters()), dict(n1.named_parameters())
for k in sorted(p0):
    ref = torch.from_numpy(g["g:" + k]).to(dev)
    s = int(g["s:" + k])
    print(f"{k:22s} bf16x3 vs fp32 {rel(p1[k].grad, p0[k].grad):.2e}   fp32 vs reference {rel(p0[k].grad.reshape(-1)[::s], ref):.2e}   "
          f"bf16x3 vs reference {rel(p1[k].grad.reshape(-1)[::s], ref):.2e}")
