"""Stride-2 bf16x3 convolution reading fp32 NCDHW (the training route and the neck): 6 waves of two column groups
(`conv_s2_cg` = 2) against 12 waves of one, 64 against 128 output channels per block (`conv_s2_ob`)."""
import torch
from mvsdet_amd import _lib, ops

dev = torch.device("cuda:0")
shapes = [("conv1 64->128", 40, 64, 128, 12, 60, 80), ("conv3 128->256", 40, 128, 256, 6, 30, 40),
          ("neck 256->512", 1, 256, 512, 40, 40, 16), ("neck 512->1024", 1, 512, 1024, 20, 20, 8)]
for name, N, Cin, Cout, D, H, W in shapes:
    x = torch.randn(N, Cin, D, H, W, device=dev)
    w = torch.randn(Cout, Cin, 3, 3, 3, device=dev) / (27 * Cin) ** 0.5
    wq = ops.split_conv_weight(w, 1)
    ref = None
    for cg, ob in ((2, 1), (0, 1), (0, 0)):
        _lib.set_option("conv_s2_cg", cg)
        _lib.set_option("conv_s2_ob", ob)
        for _ in range(2):
            y = ops.conv3d_k3_s2_bf16x3(x, wq, None, None, False)
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(10):
            y = ops.conv3d_k3_s2_bf16x3(x, wq, None, None, False)
        e1.record()
        torch.cuda.synchronize()
        if ref is None:
            ref = y
        print(f"{name}: conv_s2_cg={cg} conv_s2_ob={ob}: {e0.elapsed_time(e1) / 10:7.3f} ms   same bits as the 6-wave kernel: {bool(torch.equal(y, ref))}", flush=True)
_lib.set_option("conv_s2_cg", 0)
_lib.set_option("conv_s2_ob", 0)
