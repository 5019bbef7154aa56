"""pack_features at the headline and the reference-true shape (GPU box)."""
import torch
from mvsdet_amd import ops
dev = torch.device("cuda:0")
for N, C, H, W in ((40, 256, 120, 160), (40, 256, 60, 80)):
    f = torch.randn(N, C, H, W, device=dev)
    for _ in range(3): ops.pack_features(f)
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(20): ops.pack_features(f)
    e1.record(); torch.cuda.synchronize()
    ms = e0.elapsed_time(e1) / 20
    print(f"pack {N}x{C}x{H}x{W}: {ms:.3f} ms  {2 * f.numel() * 4 / ms / 1e6:.0f} GB/s")
