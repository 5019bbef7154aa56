"""Does the HARDWARE QUEUE a stream lands on decide what the side streams buy?  ROCm maps HIP streams onto GPU_MAX_HW_QUEUES (4)
hardware queues in creation order; two streams on one queue run one after the other.  K dummy streams are created first to shift
that order, then the three chain modes of tools/study/r06_overlap_modes.py are timed.  GPU box: python tools/study/r06_queue_phase.py K"""
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import torch  # noqa: E402

K = int(sys.argv[1]) if len(sys.argv) > 1 else 0
dev = torch.device("cuda:0")
dummies = [torch.cuda.Stream(device=dev) for _ in range(K)]
import bench  # noqa: E402
from mvsdet_amd.costreg import CostRegNet3DGS  # noqa: E402
from mvsdet_amd.head import NerfDetHeadConvs  # noqa: E402
from mvsdet_amd.hotpath import MVSDetHotPath  # noqa: E402
from mvsdet_amd.neck import IndoorImVoxelNeck  # noqa: E402

torch.manual_seed(0)
net = CostRegNet3DGS(256).to(dev).eval()
neck = IndoorImVoxelNeck(256, 128, [1, 1, 1]).to(dev).eval()
head = NerfDetHeadConvs(18, 3, 128, 6).to(dev).eval()
MODES = {"one/net1": (False, 1, 1), "one/net2": (False, 1, 2), "side1": (True, 1, 2), "side2": (True, 2, 2)}
for name in ("scannet_ref_40v_12d_60x80", "scannet_test_80v_12d_60x80", "arkit_test_100v_12d_60x80"):
    w = bench.WORKLOADS[name]
    hp = MVSDetHotPath(bench.N_VOXELS, bench.VOXEL_SIZE, list(w["near_far"]), w["D"], topk=3, cost_regularization=net, neck_3d=neck, bbox_head=head)
    pool = [bench.SceneInputs(w, i, dev) for i in range(2)]
    metas = bench.unseen_metas(w, 3, 20)
    res = {m: [] for m in MODES}
    with torch.no_grad():
        for rnd in range(2):
            for mode, (overlap, streams, vs) in MODES.items():
                hp.overlap_detector, hp.overlap_network_streams, net.view_streams = overlap, streams, vs
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
    print(f"K={K} queues={os.environ.get('GPU_MAX_HW_QUEUES', 'default')}", name, {m: round(min(ts), 3) for m, ts in res.items()}, flush=True)
    del pool, hp
    torch.cuda.empty_cache()
