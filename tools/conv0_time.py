"""conv0 of the cost network at the reference-true shape (40 x 256 -> 64 x 12 x 60 x 80, fp32 input in place, fp32 output), bf16x3 and
fp16 + MX FP6 (csrc/costreg_mx.h) alternating.  GPU box: python tools/conv0_time.py   (tools/ab_libs.sh runs it per library build)"""
import sys
sys.path.insert(0, ".")
import torch
from mvsdet_amd import ops
dev = torch.device("cuda:0")
torch.manual_seed(0)
x = torch.rand(40, 256, 12, 60, 80, device=dev)
w = torch.randn(64, 256, 3, 3, 3, device=dev) * 0.02
wq = ops.split_conv_weight(w)
sc = torch.ones(64, device=dev)
routes = {"bf16x3": lambda: ops.conv3d_k3_bf16x3(x, wq, sc, sc, True)}
if hasattr(ops, "conv3d_k3_fp16mx") and "mx" in sys.argv[1:]:
    wmx = ops.split_conv_weight_mx(w)
    routes["fp16mx"] = lambda: ops.conv3d_k3_fp16mx(x, wmx, sc, sc, True)
ts = {k: [] for k in routes}
for rnd in range(3):
    for name, fn in routes.items():
        for _ in range(2):
            fn()
        torch.cuda.synchronize()
        for _ in range(4):
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            fn()
            e1.record()
            torch.cuda.synchronize()
            ts[name].append(e0.elapsed_time(e1))
print("conv0 (fp32 in, fp32 out): " + "; ".join(f"{k} min {min(v):.3f} ms median {sorted(v)[len(v) // 2]:.3f} ms" for k, v in ts.items()), flush=True)
