"""IndoorImVoxelNeck (shipped configuration) with its stride-1 convolutions on fp32 MFMA vs bf16x3, whole neck and per
threshold; and the outputs' distance (GPU box): python tools/neck_bf16_timing.py"""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from mvsdet_amd import neck as NK  # noqa: E402


def timeit(fn, reps=10):
    fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps):
        fn()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / reps


def main():
    dev = torch.device("cuda:0")
    torch.manual_seed(0)
    net = NK.IndoorImVoxelNeck(256, 128, [1, 1, 1]).to(dev).eval()
    x = torch.randn(1, 256, 40, 40, 16, device=dev).relu()
    outs = {}
    with torch.no_grad():
        for thr in (0, 256, 2048, 16384, 2048, 256):
            NK.BF16X3_MIN_VOXELS = thr
            t = timeit(lambda: net(x))
            outs[thr] = [o.clone() for o in net(x)]
            print(f"BF16X3_MIN_VOXELS={thr}: neck {t:.3f} ms", flush=True)
    for thr in (16384, 2048, 256):
        errs = [float((a - b).abs().max()) / float(b.abs().max()) for a, b in zip(outs[thr], outs[0])]
        print(f"threshold {thr}: max rel diff per level vs fp32 route {['%.2e' % e for e in errs]}")


if __name__ == "__main__":
    main()
