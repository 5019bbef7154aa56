"""Six repetitions of the one-stream chain measurement in one process: per-scene host times, pool size, reserved memory ('sync' as argument: a synchronisation per scene).  Run on the GPU box from the repository root."""
import sys, time
sys.path.insert(0, ".")
import torch
import bench
from mvsdet_amd.costreg import CostRegNet3DGS
from mvsdet_amd.hotpath import MVSDetHotPath
from mvsdet_amd.head import NerfDetHeadConvs
from mvsdet_amd.neck import IndoorImVoxelNeck
dev = torch.device("cuda:0")
wr = bench.WORKLOADS["scannet_ref_40v_12d_60x80"]
sync_each = len(sys.argv) > 1 and sys.argv[1] == "sync"
for rep in range(6):
    torch.manual_seed(0)
    net = CostRegNet3DGS(wr["C"]).to(dev).eval()
    neck = IndoorImVoxelNeck(wr["C"], 128, [1, 1, 1]).to(dev).eval()
    head = NerfDetHeadConvs(18, 3, 128, 6).to(dev).eval()
    hp = MVSDetHotPath(bench.N_VOXELS, bench.VOXEL_SIZE, list(wr["near_far"]), wr["D"], topk=3, cost_regularization=net, neck_3d=neck, bbox_head=head)
    scene = bench.SceneInputs(wr, seed=0, device=dev)
    steps = 10
    metas = bench.unseen_metas(wr, 7, steps + 3)
    with torch.no_grad():
        for i in range(2):
            hp.prefetch_scene(metas[i + 1], dev)
            out = hp.forward_scene(scene.features, metas[i])
        bench.collect_garbage()
        torch.cuda.synchronize(dev)
        n0 = len(net._scl)
        mem0 = torch.cuda.memory_reserved(dev)
        t0 = time.perf_counter()
        per = []
        for i in range(2, steps + 2):
            ta = time.perf_counter()
            hp.prefetch_scene(metas[i + 1], dev)
            out = hp.forward_scene(scene.features, metas[i])
            if sync_each:
                torch.cuda.synchronize(dev)
            per.append(round((time.perf_counter() - ta) * 1e3, 2))
        torch.cuda.synchronize(dev)
        el = time.perf_counter() - t0
    print(f"rep {rep}: {el / steps * 1e3:.2f} ms per scene; pool {n0} -> {len(net._scl)} buffers; reserved {mem0 >> 20} -> {torch.cuda.memory_reserved(dev) >> 20} MiB; per-scene host ms {per}", flush=True)
    del out, hp, net, neck, head, scene
    torch.cuda.empty_cache()
