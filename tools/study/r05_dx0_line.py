"""One line of medians for tools/ab_libs.sh: the stride-1 bf16x3 kernel at conv0's input-gradient shape, conv2 and the neck's 40x40x16 level (what-if build without the epilogue).  Run on the GPU box from the repository root."""
import sys
sys.path.insert(0, ".")
import torch
from mvsdet_amd import ops
dev = torch.device("cuda:0")
out = []
# conv0's input gradient: 64 -> 256 on the packed SCL copy of grad_out; conv2-like 128 -> 128; neck level 0 256 -> 256
for name, N, Cin, Cout, D, H, W, pack in [("conv0_dX", 40, 64, 256, 12, 60, 80, True), ("conv2", 40, 128, 128, 6, 30, 40, False), ("neck0", 1, 256, 256, 40, 40, 16, True)]:
    x = torch.randn(N, Cin, D, H, W, device=dev)
    wq = ops.split_conv_weight(torch.randn(Cout, Cin, 3, 3, 3, device=dev) / (27 * Cin) ** 0.5)
    src = ops.scl_pack(x) if pack else x
    for _ in range(2):
        ops.conv3d_k3_bf16x3(src, wq, None, None, False)
    torch.cuda.synchronize()
    ts = []
    for _ in range(7):
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record(); r = ops.conv3d_k3_bf16x3(src, wq, None, None, False); e1.record(); torch.cuda.synchronize()
        ts.append(e0.elapsed_time(e1))
    ts.sort()
    out.append(f"{name} median {ts[3]:.3f} min {ts[0]:.3f} ms")
print(" | ".join(out))
