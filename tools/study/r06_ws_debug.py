import sys; sys.path.insert(0, ".")
import torch
from mvsdet_amd import ops, _lib
dev = torch.device("cuda:0")
torch.manual_seed(0)
for shape in ((1, 16, 4, 8, 16), (1, 8, 4, 8, 16), (1, 24, 8, 16, 32)):
    x = torch.rand(*shape, device=dev)
    w = torch.randn(64, shape[1], 3, 3, 3, device=dev) * 0.05
    wq = ops.split_conv_weight_mx(w)
    _lib.set_option("conv_mx_th", 8); a = ops.conv3d_k3_fp16mx(x, wq, None, None, False)
    _lib.set_option("conv_mx_th", 0); b = ops.conv3d_k3_fp16mx(x, wq, None, None, False); b2 = ops.conv3d_k3_fp16mx(x, wq, None, None, False)
    torch.cuda.synchronize()
    d = (a - b).abs()
    print(shape, "max diff", float(d.max()), "ws deterministic", bool(torch.equal(b, b2)), "share differing", float((d > 0).float().mean()))
    if float(d.max()) > 0:
        idx = (d > 0).nonzero()
        print("  first differing (n,c,d,h,w):", idx[:5].tolist(), " channels:", sorted(set(idx[:, 1].tolist()))[:10], " d:", sorted(set(idx[:, 2].tolist())), " h:", sorted(set(idx[:, 3].tolist())))
