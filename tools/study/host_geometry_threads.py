"""Host geometry of a scene (a1/a2/a8 + the one upload: MVSDetHotPath.prepare_scene on unseen cameras) under 1 / 4 intra-op threads
(GPU box): 0.57-0.59 ms per scene either way -- it is ~50 small ATen-CPU calls, a third of them the 40 (3x3)@(3x4) products of
compute_projection's per-view loop (the reference's own loop, mvsdet.py:1124-1156)."""
import time, torch, bench
from mvsdet_amd.hotpath import MVSDetHotPath
from mvsdet_amd import synthetic
w = bench.WORKLOADS["scannet_ref_40v_12d_60x80"]
dev = torch.device("cuda:0")
hp = MVSDetHotPath(bench.N_VOXELS, bench.VOXEL_SIZE, list(w["near_far"]), w["D"])
metas = [synthetic.make_img_meta(w["N"], (w["H"], w["W"]), seed=100 + i) for i in range(60)]
for nt in (None, 1, 4):
    if nt: torch.set_num_threads(nt)
    for m in metas[:5]: hp.prepare_scene(m, dev)
    torch.cuda.synchronize()
    t = time.perf_counter()
    for m in metas[5:55]: hp.prepare_scene(m, dev)
    torch.cuda.synchronize()
    print("threads", nt or torch.get_num_threads(), "prepare_scene ms", (time.perf_counter() - t) / 50 * 1e3)
    metas = [synthetic.make_img_meta(w["N"], (w["H"], w["W"]), seed=1000 * (nt or 9) + i) for i in range(60)]
