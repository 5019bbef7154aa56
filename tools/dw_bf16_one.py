"""The bf16x3 weight-gradient kernel at the conv0 shape, three launches (for the PMC passes of tools/pmc_dw_bf16.sh)."""
import torch
from mvsdet_amd import ops

dev = torch.device("cuda:0")
x = torch.randn(40, 256, 12, 60, 80, device=dev)
gy = torch.randn(40, 64, 12, 60, 80, device=dev)
for _ in range(3):
    ops.conv3d_k3_dw(x, gy, 16, 1, True)
torch.cuda.synchronize()
