import sys; sys.path.insert(0, ".")
import torch
from mvsdet_amd import ops
dev = torch.device("cuda:0")
torch.manual_seed(0)
x = torch.rand(1, 8, 4, 8, 16, device=dev)
w = torch.randn(64, 8, 3, 3, 3, device=dev) * 0.05
b = ops.conv3d_k3_fp16mx(x, ops.split_conv_weight_mx(w), None, None, False)
print("e_cur seen per plane d (channel 0, h 0, w 0):", [float(b[0, 0, d, 0, 0]) for d in range(4)], " per h at d=0:", [float(b[0, 0, 0, h, 0]) for h in range(8)])
print("expected floor(log2 max) - 2 =", int(torch.floor(torch.log2(x.abs().max())).item()) - 2)
