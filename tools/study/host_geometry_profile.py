"""cProfile of MVSDetHotPath.prepare_scene on unseen cameras: the caller's side only waits for the geometry worker thread."""
import cProfile, pstats, torch, bench
from mvsdet_amd.hotpath import MVSDetHotPath
from mvsdet_amd import synthetic
w = bench.WORKLOADS["scannet_ref_40v_12d_60x80"]
dev = torch.device("cuda:0")
hp = MVSDetHotPath(bench.N_VOXELS, bench.VOXEL_SIZE, list(w["near_far"]), w["D"])
metas = [synthetic.make_img_meta(w["N"], (w["H"], w["W"]), seed=100 + i) for i in range(60)]
for m in metas[:5]: hp.prepare_scene(m, dev)
pr = cProfile.Profile(); pr.enable()
for m in metas[5:55]: hp.prepare_scene(m, dev)
pr.disable()
pstats.Stats(pr).sort_stats("tottime").print_stats(28)
