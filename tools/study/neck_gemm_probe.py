"""Probe (GPU box): the neck's 10x10x4 level (1024 -> 1024 channels, 400 voxels) as im2col + three bf16 GEMMs with fp32
output against our fp32-MFMA convolution.  Does torch.mm take out_dtype here, and what do the GEMMs cost?"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
import torch.nn.functional as F
from mvsdet_amd import ops

dev = torch.device("cuda:0")


def timeit(fn, reps=10):
    fn(); torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps):
        fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / reps


def im2col(x):   # (1,C,D,H,W) -> (D*H*W, C*27)
    _, C, D, H, W = x.shape
    xp = F.pad(x, (1, 1, 1, 1, 1, 1))
    cols = xp.unfold(2, 3, 1).unfold(3, 3, 1).unfold(4, 3, 1)        # (1,C,D,H,W,3,3,3)
    return cols.permute(0, 2, 3, 4, 1, 5, 6, 7).reshape(D * H * W, C * 27)


for C, D, H, W in ((1024, 4, 10, 10), (512, 8, 20, 20)):
    x = torch.randn(1, C, D, H, W, device=dev)
    w = torch.randn(C, C, 3, 3, 3, device=dev) / (27 * C) ** 0.5
    wp = ops.permute_conv_weight(w)
    ref = ops.conv3d_k3_mfma(x, wp, None, None, False, 1)
    t_ours = timeit(lambda: ops.conv3d_k3_mfma(x, wp, None, None, False, 1))
    wm = w.reshape(C, C * 27).t().contiguous()                        # (K, N)
    w_hi = wm.bfloat16(); w_mid = (wm - w_hi.float()).bfloat16()
    t_col = timeit(lambda: im2col(x))
    a = im2col(x)
    t32 = timeit(lambda: a @ wm)
    err32 = float(((a @ wm).t().reshape(1, C, D, H, W) - ref).abs().max() / ref.abs().max())
    print(f"C={C} {D}x{H}x{W}: ours fp32-MFMA {t_ours*1e3:.0f} us; im2col {t_col*1e3:.0f} us; fp32 GEMM {t32*1e3:.0f} us (rel {err32:.1e})", flush=True)
    try:
        def three():
            a_hi = a.bfloat16(); a_mid = (a - a_hi.float()).bfloat16()
            return (torch.mm(a_hi, w_hi, out_dtype=torch.float32) + torch.mm(a_hi, w_mid, out_dtype=torch.float32)
                    + torch.mm(a_mid, w_hi, out_dtype=torch.float32))
        y = three()
        err = float((y.t().reshape(1, C, D, H, W) - ref).abs().max() / ref.abs().max())
        print(f"   three bf16 GEMMs, fp32 out: {timeit(three)*1e3:.0f} us (rel {err:.1e})", flush=True)
    except Exception as e:
        print("   torch.mm(out_dtype=float32) not available:", repr(e)[:200], flush=True)
