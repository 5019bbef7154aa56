#!/usr/bin/env python3
"""The shipped 3-D neck (and neck + head) with its GEMM-shaped layers on csrc/neck_gemm.hip against rocBLAS + ATen glue (rounds
2-4), same process, alternating; batch 1 and 4.  GPU box: python tools/neck_gemm_ab.py"""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa: E402

from mvsdet_amd import neck as NK  # noqa: E402
from mvsdet_amd.head import NerfDetHeadConvs  # noqa: E402

dev = torch.device("cuda:0")
torch.manual_seed(0)
neck = NK.IndoorImVoxelNeck(256, 128, [1, 1, 1]).to(dev).eval()
head = NerfDetHeadConvs(18, 3, 128, 6).to(dev).eval()


def timed(fn, reps=10):
    for _ in range(3):
        fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps):
        fn()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / reps


with torch.no_grad():
    for bsz in (1, 4):
        x = torch.randn(bsz, 256, 40, 40, 16, device=dev)
        for rnd in range(2):
            for flag in (False, True):
                NK.GEMM_BF16X3 = flag
                NK.drop_derived_tensors(neck)
                tn = timed(lambda: neck(x))
                tnh = timed(lambda: head(neck(x)))
                print(f"batch {bsz} {'neck_gemm.hip' if flag else 'rocBLAS+ATen '}: neck {tn:.3f} ms ({tn / bsz:.3f} per scene), neck + head {tnh:.3f} ms ({tnh / bsz:.3f} per scene)", flush=True)
