"""Probe (GPU box): the head's fused convolutions on fp32 MFMA vs bf16x3."""
import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from mvsdet_amd import head as HD
def timeit(fn, reps=20):
    fn(); torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / reps
dev = torch.device("cuda:0"); torch.manual_seed(0)
h = HD.NerfDetHeadConvs().to(dev).eval(); h.init_weights()
xs = [torch.randn(1, 128, 40 >> i, 40 >> i, 16 >> i, device=dev) for i in range(3)]
outs = {}
with torch.no_grad():
    for flag in (False, True, False, True):
        HD.HEAD_BF16X3 = flag
        h(xs)
        print("HEAD_BF16X3", flag, f"{timeit(lambda: h(xs)):.3f} ms", flush=True)
        outs[flag] = h(xs)
for a, b in zip(outs[False], outs[True]):
    print([f"{float((u - v).abs().max() / v.abs().max()):.1e}" for u, v in zip(a, b)])
