import sys, numpy as np, torch
"""Stage-1 gradient errors against the fixtures' scale (VERDICT r5 weak #3: what the absolute tolerances of test_backward_stage1* mean)."""
sys.path.insert(0, "."); sys.path.insert(0, "tests")
import importlib
tp = importlib.import_module("test_gpu_parity")
tp.load_golden = importlib.import_module("conftest").load_golden
from mvsdet_amd import ops
from oracle import oracle as orc
orc.build()
gpu = torch.device("cuda:0")
g = tp.load_golden("g6_backward")
feat = tp.dev(g["s1_feature"], gpu).requires_grad_(True)
var = ops.plane_sweep_variance(feat, tp.dev(g["s1_neighbor_ids"], gpu), tp.dev(g["s1_proj_rel"], gpu), tp.dev(g["s1_depth_values"], gpu))
(var * tp.dev(g["s1_R"], gpu)).sum().backward()
got, ref = feat.grad.cpu().numpy(), g["s1_grad_feature"]
print("g6 s1: scale (max |ref|)", np.abs(ref).max(), "rms", np.sqrt((ref**2).mean()), "max |d|", np.abs(got-ref).max(), "max |d| / scale", np.abs(got-ref).max()/np.abs(ref).max())
for tag in ("n3_d8", "n6_d12_arkit"):
    g = tp.load_golden("g2_variance_" + tag)
    feat = tp.dev(g["feature"], gpu).requires_grad_(True)
    args = (tp.dev(g["neighbor_ids"], gpu), tp.dev(g["proj_rel"], gpu), tp.dev(g["depth_values"], gpu))
    var = ops.plane_sweep_variance(feat, *args)
    R = torch.randn(var.shape, generator=torch.Generator().manual_seed(1)).to(gpu)
    (var * R).sum().backward()
    ref = orc.plane_sweep_variance_bwd(g["feature"], g["neighbor_ids"], g["proj_rel"], g["depth_values"], R.cpu())
    got = feat.grad.cpu().numpy()
    print(tag, "scale", np.abs(ref).max(), "rms", np.sqrt((ref**2).mean()), "max |d|", np.abs(got-ref).max(), "rel", np.abs(got-ref).max()/np.abs(ref).max())
