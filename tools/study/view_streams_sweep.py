"""CostRegNet3DGS.view_streams (the eval chain on k pieces of the views on k streams) at the view counts the shipped pipelines run:
the network alone and the whole chain on one caller stream.  GPU box: python tools/study/view_streams_sweep.py"""
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import torch  # noqa: E402

import bench  # noqa: E402
from mvsdet_amd.costreg import CostRegNet3DGS  # noqa: E402
from mvsdet_amd.head import NerfDetHeadConvs  # noqa: E402
from mvsdet_amd.hotpath import MVSDetHotPath  # noqa: E402
from mvsdet_amd.neck import IndoorImVoxelNeck  # noqa: E402

dev = torch.device("cuda:0")
torch.manual_seed(0)
net = CostRegNet3DGS(256).to(dev).eval()
neck = IndoorImVoxelNeck(256, 128, [1, 1, 1]).to(dev).eval()
head = NerfDetHeadConvs(18, 3, 128, 6).to(dev).eval()
for name in ("scannet_ref_40v_12d_60x80", "scannet_test_80v_12d_60x80", "arkit_test_100v_12d_60x80"):
    w = bench.WORKLOADS[name]
    hp = MVSDetHotPath(bench.N_VOXELS, bench.VOXEL_SIZE, list(w["near_far"]), w["D"], topk=3, cost_regularization=net, neck_3d=neck, bbox_head=head)
    s = bench.SceneInputs(w, 0, dev)
    metas = bench.unseen_metas(w, 3, 16)
    with torch.no_grad():
        out = hp.forward_scene(s.features, metas[0])
        var = out.raw("variance")
        for rnd in range(2):
            for k in (1, 2, 3, 4):
                net.view_streams = k
                for overlap in (False, True):
                    hp.overlap_detector = overlap
                    for i in range(2):
                        hp.forward_scene(s.features, metas[i])
                    torch.cuda.synchronize()
                    t0 = time.perf_counter()
                    for i in range(2, 10):
                        hp.prefetch_scene(metas[i + 1], dev)
                        hp.forward_scene(s.features, metas[i])
                    torch.cuda.synchronize()
                    el = (time.perf_counter() - t0) / 8 * 1e3
                    if overlap:
                        chain_o = el
                    else:
                        chain = el
                hp.overlap_detector = False
                net(var)
                torch.cuda.synchronize()
                e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                e0.record()
                for _ in range(4):
                    net(var)
                e1.record()
                torch.cuda.synchronize()
                print(f"{name} view_streams={k}: network {e0.elapsed_time(e1) / 4:.3f} ms, chain {chain:.3f} ms, chain with the detector on the side stream "
                      f"{chain_o:.3f} ms (there the network runs on one stream)", flush=True)
    del out, var, s, hp
    torch.cuda.empty_cache()
