"""Repeat the 'scene alone' run of tests/test_gpu_scenes.py at 100 views and localise any run that differs from the first."""
import sys
sys.path.insert(0, ".")
sys.path.insert(0, "tests")
import torch
from mvsdet_amd import synthetic
from mvsdet_amd.costreg import CostRegNet3DGS
from mvsdet_amd.head import NerfDetHeadConvs
from mvsdet_amd.hotpath import MVSDetHotPath
from mvsdet_amd.neck import IndoorImVoxelNeck
gpu = torch.device("cuda:0")
pollute = [torch.cuda.Stream(device=gpu) for _ in range(int(sys.argv[1]) if len(sys.argv) > 1 else 0)]
C, D, hw, grid, vox = 256, 12, (60, 80), [40, 40, 16], [0.16, 0.16, 0.2]
big = synthetic.make_features(100, C, hw, seed=6).to(gpu)
meta = synthetic.make_img_meta(100, hw, seed=60, per_view_intrinsics=True)
ref = None
for it in range(24):
    if it % 6 == 0:   # fresh modules, as a new test would have
        torch.manual_seed(0)
        net = CostRegNet3DGS(C).to(gpu).eval()
        neck = IndoorImVoxelNeck(C, 128, [1, 1, 1]).to(gpu).eval()
        torch.manual_seed(1)
        head = NerfDetHeadConvs(17, 3, 128, 7, arkit_head=True).to(gpu).eval()
    torch.cuda.synchronize()
    net._scl.clear()
    hp = MVSDetHotPath(grid, vox, [0.2, 5.0], D, topk=3, cost_regularization=net, neck_3d=neck, bbox_head=head)
    hp.overlap_detector = bool(it % 2)
    with torch.no_grad():
        out = hp.forward_scene(big, meta)
        got = {k: out[k].clone() for k in ("variance", "prob_volume", "volume", "depth_coding")}
    torch.cuda.synchronize()
    del out
    if ref is None:
        ref = got
        continue
    diffs = {k: float((got[k].double() - ref[k].double()).abs().max()) for k in got}
    if any(v > 0 for v in diffs.values()):
        pv = (got["prob_volume"] - ref["prob_volume"]).abs().amax(dim=(1, 2, 3))
        print(f"run {it} (overlap {hp.overlap_detector}, fresh modules at {it - it % 6}): differs {diffs}; views with another prob_volume: {torch.nonzero(pv > 0).flatten().tolist()}", flush=True)
print("done", flush=True)
