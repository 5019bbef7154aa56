#!/usr/bin/env python3
"""The cost network on 40 views as ONE batch against two halves of 20 views on two streams (views are independent in eval mode):
do the tails of one half's kernels (800-block grids = 3.1 rounds of the chip) fill with the other half's blocks?  GPU box."""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch  # noqa: E402

from mvsdet_amd.costreg import CostRegNet3DGS  # noqa: E402

dev = torch.device("cuda:0")
torch.manual_seed(0)
net = CostRegNet3DGS(256).to(dev).eval()
x = torch.rand(40, 256, 12, 60, 80, device=dev)
side = torch.cuda.Stream(device=dev)


def one():
    return net(x)


def two(cut=20):
    cur = torch.cuda.current_stream(dev)
    side.wait_stream(cur)
    with torch.cuda.stream(side):
        b = net(x[cut:])
    a = net(x[:cut])
    cur.wait_stream(side)
    b.record_stream(cur)
    return torch.cat((a, b), 0)


def timed(fn, reps=10):
    for _ in range(3):
        fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps):
        out = fn()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / reps, out


sides = [torch.cuda.Stream(device=dev) for _ in range(3)]


def many(k):
    cur = torch.cuda.current_stream(dev)
    cuts = [round(40 * i / k) for i in range(k + 1)]
    outs = [None] * k
    for i in range(1, k):
        sides[i - 1].wait_stream(cur)
        with torch.cuda.stream(sides[i - 1]):
            outs[i] = net(x[cuts[i]:cuts[i + 1]])
    outs[0] = net(x[:cuts[1]])
    for i in range(1, k):
        cur.wait_stream(sides[i - 1])
        outs[i].record_stream(cur)
    return torch.cat(outs, 0)


with torch.no_grad():
    for k in (3, 4):
        t, o = timed(lambda: many(k))
        print(f"{k} pieces on {k} streams: {t:.3f} ms (equal bits: {bool(torch.equal(o, net(x)))})", flush=True)
    for _ in range(2):
        t1, o1 = timed(one)
        t2, o2 = timed(two)
        t3, o3 = timed(lambda: two(24))
        print(f"one batch of 40: {t1:.3f} ms   20 + 20 on two streams: {t2:.3f} ms (equal bits: {bool(torch.equal(o1, o2))})   24 + 16: {t3:.3f} ms", flush=True)
