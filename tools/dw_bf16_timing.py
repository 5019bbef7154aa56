"""Weight-gradient kernels of the cost network's layers: fp32 MFMA (csrc/costreg_dw.hip) against bf16x3
(csrc/costreg_dw_bf16.hip; stride 2 / transposed: csrc/costreg_dw_s2_bf16.hip), at the layer shapes of the headline scene
(N=40 cost volumes).  Stride-2 rows: (fine channels, coarse channels, fine D, H, W) -- conv9 / conv11 are the same call with
the layer's grad_out as the fine tensor."""
import sys
import torch
from mvsdet_amd import ops

dev = torch.device("cuda:0")
shapes = [("conv0", 40, 256, 64, 12, 60, 80, 1), ("conv2", 40, 128, 128, 6, 30, 40, 1), ("conv4", 40, 256, 256, 3, 15, 20, 1),
          ("conv1", 40, 64, 128, 12, 60, 80, 2), ("conv3", 40, 128, 256, 6, 30, 40, 2),
          ("conv9", 40, 128, 256, 6, 30, 40, 2), ("conv11", 40, 64, 128, 12, 60, 80, 2)]
splits = [int(a) for a in sys.argv[1:]] or [0]
for name, N, Cin, Cout, D, H, W, st in shapes:
    x = torch.randn(N, Cin, D, H, W, device=dev)
    gy = torch.randn(N, Cout, D // st, H // st, W // st, device=dev)
    for bf in (False, True):
        for ns in splits:
            for _ in range(2):
                ops.conv3d_k3_dw(x, gy, ns, st, bf)
            torch.cuda.synchronize()
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            for _ in range(5):
                out = ops.conv3d_k3_dw(x, gy, ns, st, bf)
            e1.record()
            torch.cuda.synchronize()
            ms = e0.elapsed_time(e1) / 5
            flop = 2.0 * N * (D // st) * (H // st) * (W // st) * Cin * Cout * 27
            print(f"{name} {'bf16x3' if bf else 'fp32  '} nsplit={ns:4d}: {ms:7.3f} ms  {flop / ms / 1e9:8.1f} TFLOP/s useful", flush=True)
    a = ops.conv3d_k3_dw(x, gy, 0, st, True)
    b = ops.conv3d_k3_dw(x, gy, 0, st, False)
    print(f"{name}: max |bf16x3 - fp32| / max |fp32| = {float((a - b).abs().max() / b.abs().max()):.2e}", flush=True)
