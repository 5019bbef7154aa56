import os, sys, torch
sys.path.insert(0, "/root/repo")
import bench
from mvsdet_amd import ops, _lib
from mvsdet_amd.hotpath import MVSDetHotPath
dev = torch.device("cuda:0")
for name in ("stress_100v_128d_240x320_c32", "scannet_40v_64d_120x160", "arkit_50v_96d_60x80"):
    w = bench.WORKLOADS[name]
    hp = MVSDetHotPath(bench.N_VOXELS, bench.VOXEL_SIZE, list(w["near_far"]), w["D"])
    meta = bench.SceneInputs(w, 0, dev).meta if False else None
    from mvsdet_amd import synthetic
    meta = synthetic.make_img_meta(w["N"], (w["H"], w["W"]), seed=0)
    geo = hp.prepare_scene(meta, dev)
    for cap in (-1, 0):
        if cap >= 0:
            _lib.check(_lib.load().mvsdet_set_option(b"sweep_boxcap", cap), "opt")
        def run():
            return ops.plane_sweep_table(geo.proj_rel, geo.depth_values, w["H"], w["W"])
        run(); torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(5): run()
        e1.record(); torch.cuda.synchronize()
        print(name, "boxcap", cap, f"{e0.elapsed_time(e1)/5:.3f} ms", flush=True)
    _lib.check(_lib.load().mvsdet_set_option(b"sweep_boxcap", 1 << 20), "opt")
