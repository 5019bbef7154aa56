"""One line of medians for tools/ab_libs.sh: the stride-2 weight gradient at the conv1 / conv3 shapes (round 5: coalesced staging loads).  Run on the GPU box from the repository root."""
import sys
sys.path.insert(0, ".")
import torch
from mvsdet_amd import ops
dev = torch.device("cuda:0")
out = []
for name, N, Cin, Cout, D, H, W in [("conv1", 40, 64, 128, 12, 60, 80), ("conv3", 40, 128, 256, 6, 30, 40)]:
    x = torch.randn(N, Cin, D, H, W, device=dev)
    gy = torch.randn(N, Cout, D // 2, H // 2, W // 2, device=dev)
    for _ in range(2):
        ops.conv3d_k3_dw(x, gy, 0, 2, True)
    torch.cuda.synchronize()
    ts = []
    for _ in range(9):
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record(); r = ops.conv3d_k3_dw(x, gy, 0, 2, True); e1.record(); torch.cuda.synchronize()
        ts.append(e0.elapsed_time(e1))
    ts.sort()
    out.append(f"{name} median {ts[4]:.3f} min {ts[0]:.3f} ms checksum {float(r.double().abs().sum()):.6f}")
print(" | ".join(out))
