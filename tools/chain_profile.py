#!/usr/bin/env python3
"""The whole chain of a scene (a1..a10 + CostRegNet_3DGS + neck + head convolutions, eval, one stream) at the reference-true shape,
SCENES times: target of `rocprofv3 --kernel-trace --stats`.  `python tools/chain_profile.py summary <stats.csv> [SCENES]` prints the
kernels of one scene by time."""
import csv
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
SCENES = 10

if len(sys.argv) > 2 and sys.argv[1] == "summary":
    n = int(sys.argv[3]) if len(sys.argv) > 3 else SCENES
    rows = list(csv.DictReader(open(sys.argv[2])))
    tot = sum(float(r["TotalDurationNs"]) for r in rows)
    print(f"all kernels: {tot / n / 1e6:.3f} ms per scene over {n} scenes (warm-up scenes included in the division: see the tool)")
    for r in sorted(rows, key=lambda r: -float(r["TotalDurationNs"]))[:40]:
        print(f"{float(r['TotalDurationNs']) / n / 1e6:8.3f} ms  {int(r['Calls']) / n:6.1f} calls  {r['Name'][:130]}")
    sys.exit(0)

import torch  # noqa: E402

import bench  # noqa: E402
from mvsdet_amd.costreg import CostRegNet3DGS  # noqa: E402
from mvsdet_amd.head import NerfDetHeadConvs  # noqa: E402
from mvsdet_amd.hotpath import MVSDetHotPath  # noqa: E402
from mvsdet_amd.neck import IndoorImVoxelNeck  # noqa: E402

dev = torch.device("cuda:0")
wr = bench.WORKLOADS["scannet_ref_40v_12d_60x80"]
torch.manual_seed(0)
net = CostRegNet3DGS(wr["C"]).to(dev).eval()
net.view_streams = 1   # per-kernel times: one batch on one stream (two halves on two streams overlap their kernels)
neck = IndoorImVoxelNeck(wr["C"], 128, [1, 1, 1]).to(dev).eval()
head = NerfDetHeadConvs(18, 3, 128, 6).to(dev).eval()
hp = MVSDetHotPath(bench.N_VOXELS, bench.VOXEL_SIZE, list(wr["near_far"]), wr["D"], topk=3, cost_regularization=net, neck_3d=neck,
                   bbox_head=head)
scene = bench.SceneInputs(wr, seed=0, device=dev)
metas = bench.unseen_metas(wr, 7, SCENES + 1)
with torch.no_grad():
    for i in range(SCENES):
        hp.prefetch_scene(metas[i + 1], dev)
        out = hp.forward_scene(scene.features, metas[i])
    torch.cuda.synchronize(dev)
print("done", flush=True)
