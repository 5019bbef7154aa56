"""A bf16x3 weight-gradient kernel, three launches (for the PMC passes of tools/pmc_dw_bf16.sh): the stride-1 one at the conv0
shape (default), or `s2`: the stride-2 / transposed one at the conv1 shape."""
import sys
import torch
from mvsdet_amd import ops

dev = torch.device("cuda:0")
if len(sys.argv) > 1 and sys.argv[1] == "s2":
    x = torch.randn(40, 64, 12, 60, 80, device=dev)
    gy = torch.randn(40, 128, 6, 30, 40, device=dev)
    for _ in range(3):
        ops.conv3d_k3_dw(x, gy, 0, 2, True)
else:
    x = torch.randn(40, 256, 12, 60, 80, device=dev)
    gy = torch.randn(40, 64, 12, 60, 80, device=dev)
    for _ in range(3):
        ops.conv3d_k3_dw(x, gy, 16, 1, True)
torch.cuda.synchronize()
