"""Where the detector tail runs and how many view streams the cost network uses, at the view counts the shipped pipelines run
(the pipelined route must never lose).  Modes: one = everything on the caller's stream, network on two view
streams (the default without overlap_detector); side1 = detector tail on the side stream, network on one stream (round 5's
overlap_detector); side2 = tail on the side stream AND two view streams.  Alternating, three rounds, 12 timed scenes each, new
cameras every scene.  GPU box: python tools/study/r06_overlap_modes.py"""
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
MODES = {"one": (False, 1, 0), "side1": (True, 1, 0), "side2": (True, 2, 0), "side1_lo": (True, 1, 1), "side2_lo": (True, 2, 1)}
print("stream priority range (least, greatest):", torch.cuda.Stream.priority_range(), flush=True)
try:
    LOW = torch.cuda.Stream(device=dev, priority=torch.cuda.Stream.priority_range()[0])
    print("low-priority stream:", LOW.priority, flush=True)
except Exception as exc:  # noqa: BLE001
    LOW = None
    print("no low-priority stream:", exc, flush=True)
NORMAL = torch.cuda.Stream(device=dev)
for name in sys.argv[1:] or ("scannet_ref_40v_12d_60x80", "scannet_test_80v_12d_60x80", "arkit_test_100v_12d_60x80"):
    w = bench.WORKLOADS[name]
    hp = MVSDetHotPath(bench.N_VOXELS, bench.VOXEL_SIZE, list(w["near_far"]), w["D"], topk=3, cost_regularization=net, neck_3d=neck, bbox_head=head)
    pool = [bench.SceneInputs(w, i, dev) for i in range(2)]
    metas = bench.unseen_metas(w, 3, 20)
    res = {m: [] for m in MODES}
    with torch.no_grad():
        for rnd in range(4):
            for mode, (overlap, streams, low) in MODES.items():
                if low and LOW is None:
                    continue
                hp.overlap_detector, hp.overlap_network_streams = overlap, streams
                hp._detector_streams[str(dev)] = LOW if low else NORMAL   # _lo: the detector tail on a stream of the lowest priority
                bench.collect_garbage()
                outs = [hp.forward_scene(pool[i % 2].features, metas[i]) for i in range(4)]
                torch.cuda.synchronize()
                t0 = time.perf_counter()
                for i in range(4, 16):
                    hp.prefetch_scene(metas[i + 1], dev)
                    outs.append(hp.forward_scene(pool[i % 2].features, metas[i]))
                    outs.pop(0)
                torch.cuda.synchronize()
                res[mode].append((time.perf_counter() - t0) / 12 * 1e3)
                del outs
    print(name, {m: [round(v, 3) for v in ts] for m, ts in res.items() if ts}, "min:", {m: round(min(ts), 3) for m, ts in res.items() if ts}, flush=True)
    del pool, hp
    torch.cuda.empty_cache()
