"""Convolution and BatchNorm with and without the statistics taken in the convolution's epilogue (HIP events, medians of 8).  Run on the GPU box from the repository root."""
import sys
sys.path.insert(0, ".")
import torch
from mvsdet_amd import ops
dev = torch.device("cuda:0")
torch.manual_seed(0)
def ev(fn, reps=8):
    fn(); torch.cuda.synchronize()
    ts = []
    for _ in range(reps):
        a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        a.record(); fn(); b.record(); torch.cuda.synchronize()
        ts.append(a.elapsed_time(b))
    ts.sort()
    return ts[len(ts) // 2]
for (N, Cin, Cout, D, H, W) in [(40, 256, 64, 12, 60, 80), (40, 128, 128, 6, 30, 40), (40, 256, 256, 3, 15, 20)]:
    x = torch.rand(N, Cin, D, H, W, device=dev)
    wq = ops.split_conv_weight(torch.randn(Cout, Cin, 3, 3, 3, device=dev) / (27 * Cin) ** 0.5)
    g, b = torch.ones(Cout, device=dev), torch.zeros(Cout, device=dev)
    pv = torch.zeros(Cout, device=dev)
    t_conv = ev(lambda: ops.conv3d_k3_bf16x3(x, wq, None, None, False))
    t_convs = ev(lambda: ops.conv3d_k3_bf16x3_stats(x, wq, pv))
    y, parts = ops.conv3d_k3_bf16x3_stats(x, wq, pv)
    t_bn2 = ev(lambda: ops.bn3d_relu_train(y, g, b, 1e-5, True))
    t_bn1 = ev(lambda: ops.bn3d_relu_train(y, g, b, 1e-5, True, None, parts, pv))
    print(f"{(N, Cin, Cout, D, H, W)}: conv {t_conv:.3f} ms, conv + statistics {t_convs:.3f} ms; BatchNorm two passes {t_bn2:.3f} ms, from the partial sums {t_bn1:.3f} ms; parts per channel {parts.shape[1]}", flush=True)
