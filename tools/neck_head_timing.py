#!/usr/bin/env python3
"""The 3-D neck and the detection head's convolutions of one scene (the (1,256,40,40,16) volume of the shipped config) under
several settings of the input-channel split of the small levels: `option=value[,value..]` pairs are swept (GPU box)."""
import itertools
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa: E402

from mvsdet_amd import _lib  # noqa: E402
from mvsdet_amd.head import NerfDetHeadConvs  # noqa: E402
from mvsdet_amd.neck import IndoorImVoxelNeck  # noqa: E402

dev = torch.device("cuda:0")
torch.manual_seed(0)
neck = IndoorImVoxelNeck(256, 128, [1, 1, 1]).to(dev).eval()
head = NerfDetHeadConvs(18, 3, 128, 6).to(dev).eval()
x = torch.randn(1, 256, 40, 40, 16, device=dev)
sweeps = [(kv.split("=")[0], [int(v) for v in kv.split("=")[1].split(",")]) for kv in sys.argv[1:]] or [("conv_split_blocks", [768])]


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


with torch.no_grad():
    for combo in itertools.product(*[vals for _, vals in sweeps]):
        for (name, _), v in zip(sweeps, combo):
            _lib.set_option(name, v)
        tn, feats = timed(lambda: neck(x))
        th, _ = timed(lambda: head(feats))
        print(" ".join(f"{n}={v}" for (n, _), v in zip(sweeps, combo)), f": neck {tn:.3f} ms  head {th:.3f} ms  together {tn + th:.3f} ms", flush=True)
