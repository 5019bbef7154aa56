import sys, os
sys.path.insert(0, ".")
import torch
from mvsdet_amd import ops
dev = torch.device("cuda:0")
torch.manual_seed(0)
x = torch.rand(40, 256, 12, 60, 80, device=dev)
w = torch.randn(64, 256, 3, 3, 3, device=dev) * 0.02
wq = ops.split_conv_weight(w)
sc = torch.ones(64, device=dev)
ts = []
for mode in ("f32",):
    for _ in range(3):
        ops.conv3d_k3_bf16x3(x, wq, sc, sc, True)
    torch.cuda.synchronize()
    for _ in range(6):
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        ops.conv3d_k3_bf16x3(x, wq, sc, sc, True)
        e1.record()
        torch.cuda.synchronize()
        ts.append(e0.elapsed_time(e1))
print(f"conv0 (fp32 in, fp32 out): min {min(ts):.3f} ms median {sorted(ts)[3]:.3f} ms", flush=True)
