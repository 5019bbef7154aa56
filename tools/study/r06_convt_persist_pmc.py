"""One launch each of the per-tile and the persistent transposed kernel at conv11 of the reference-true shape (40 views), for the counters:
rocprofv3 --pmc ... -- python3 tools/study/r06_convt_persist_pmc.py   (what-if forms: a library built with -DMVS_CONVT_WHATIF=n, MVSDET_HIP_LIB)"""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from mvsdet_amd import _lib, ops  # noqa: E402

dev = torch.device("cuda")
N, Cin, Cout, D, H, W = 40, 128, 64, 6, 30, 40
g = torch.Generator(device="cpu").manual_seed(5)
x = torch.randn((N, Cin, D, H, W), generator=g).to(dev)
w = (torch.randn((Cin, Cout, 3, 3, 3), generator=g) * 0.05).to(dev)
sc = (torch.rand(Cout, generator=g) + 0.5).to(dev)
sh = torch.randn(Cout, generator=g).to(dev)
res = torch.randn((N, Cout, 2 * D, 2 * H, 2 * W), generator=g).to(dev)
xs, wq = ops.scl_pack(x), ops.split_conv_weight(w, 2)
for persist in (0, 1):
    _lib.set_option("convT_persist", persist)
    for _ in range(3):
        ops.convT3d_k3_s2_bf16x3(xs, wq, sc, sh, res, True, outputs=("f32",))
    torch.cuda.synchronize()
print("done")
