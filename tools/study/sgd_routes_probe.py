"""How far do the loss trajectories of the cost network's training routes drift apart under plain SGD, and how far do two fp32
routes (our fp32 MFMA kernels against the framework's ATen / MIOpen layers) drift from each other -- the yardstick for a drift that
is dynamics, not arithmetic?  Run on the GPU box: python tools/study/sgd_routes_probe.py"""
import os
import sys

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests", "golden"))
from lcg import lcg_fill_state, lcg_uniform  # noqa: E402
from mvsdet_amd.costreg import CostRegNet3DGS  # noqa: E402

dev = torch.device("cuda:0")
g = np.load(os.path.join(ROOT, "tests", "golden", "g12_cost_regularisation_grads.npz"))
shape = tuple(int(v) for v in g["in_shape"])
x = torch.from_numpy(lcg_uniform(int(np.prod(shape)), int(g["input_seed"]))).reshape(shape).abs().to(dev)
R = torch.from_numpy(lcg_uniform(2 * int(np.prod(shape[2:])) * shape[0], int(g["r_seed"]))).reshape(shape[0], 2, *shape[2:]).to(dev)


def run(route, lr, steps=30):
    net = CostRegNet3DGS(256, 64).train()
    if route == "aten":
        net.hip_backward = False
        net.matrix_precision = "fp32"
    else:
        net.matrix_precision = route
    with torch.no_grad():
        lcg_fill_state(net, int(g["weight_seed"]))
    net = net.to(dev)
    opt = torch.optim.SGD(net.parameters(), lr=lr)
    tr = []
    for _ in range(steps):
        opt.zero_grad(set_to_none=True)
        loss = ((net(x) - R) ** 2).mean()
        loss.backward()
        opt.step()
        tr.append(float(loss.detach()))
    return np.array(tr)


for lr in (3e-3, 1e-3, 3e-4):
    t = {r: run(r, lr) for r in ("fp32", "bf16x3", "aten")}
    a = t["fp32"]
    print(f"lr {lr:g}: fp32 loss {a[0]:.5f} -> {a[-1]:.5f}")
    for r in ("bf16x3", "aten"):
        rel = np.abs(t[r] - a) / np.abs(a)
        print(f"   {r:7s} vs fp32: max rel {rel.max():.2e} (step {int(rel.argmax())}), at steps 0/9/19/29: " + " ".join(f"{rel[i]:.1e}" for i in (0, 9, 19, 29)))
