"""Sweep into a contiguous volume vs a row-pitched one (rows on 128-byte lines, 32x4 tiles) at the 80-wide shapes
(GPU box): python tools/sweep_pitch_timing.py"""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from mvsdet_amd import ops, synthetic  # noqa: E402
from mvsdet_amd.hotpath import MVSDetHotPath  # noqa: E402


def timeit(fn, reps=10):
    fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps):
        fn()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / reps


def main():
    dev = torch.device("cuda:0")
    for name, N, C, D, H, W, nf, pvk in (("scannet_ref 40v 12d 60x80", 40, 256, 12, 60, 80, (0.2, 5.0), False),
                                          ("arkit 50v 96d 60x80", 50, 256, 96, 60, 80, (0.5, 5.5), True),
                                          ("scannet 40v 64d 120x160", 40, 256, 64, 120, 160, (0.2, 5.0), False)):
        hp = MVSDetHotPath([40, 40, 16], [0.16, 0.16, 0.2], list(nf), D)
        feat = synthetic.make_features(N, C, (H, W), seed=1, device=dev)
        geo = hp.prepare_scene(synthetic.make_img_meta(N, (H, W), seed=1, per_view_intrinsics=pvk), dev)
        packed = ops.pack_features(feat)
        wp = ops.sweep_row_pitch(W)
        t0 = ops.plane_sweep_table(geo.proj_rel, geo.depth_values, H, W)
        t1 = ops.plane_sweep_table_pitched(geo.proj_rel, geo.depth_values, H, W, wp)
        a = ops.plane_sweep_variance_tabled(packed, geo.neighbor_ids, t0, C, D, H, W)
        b = ops.plane_sweep_variance_tabled_pitched(packed, geo.neighbor_ids, t1, C, D, H, W, wp)
        same = bool(torch.equal(a, b))
        del a, b
        ta = timeit(lambda: ops.plane_sweep_variance_tabled(packed, geo.neighbor_ids, t0, C, D, H, W))
        tb = timeit(lambda: ops.plane_sweep_variance_tabled_pitched(packed, geo.neighbor_ids, t1, C, D, H, W, wp))
        tt0 = timeit(lambda: ops.plane_sweep_table(geo.proj_rel, geo.depth_values, H, W))
        tt1 = timeit(lambda: ops.plane_sweep_table_pitched(geo.proj_rel, geo.depth_values, H, W, wp))
        gb = (3 * C * H * W * 4 + C * D * H * W * 4) * N / 1e9
        from mvsdet_amd import _lib
        for rep in range(3):
            line = []
            for ds in (0, 1, 2, 3, 4, 6):
                if ds > D:
                    continue
                _lib.set_option("sweep_dsplit", ds)
                tc = timeit(lambda: ops.plane_sweep_variance_tabled(packed, geo.neighbor_ids, t0, C, D, H, W))
                tp = timeit(lambda: ops.plane_sweep_variance_tabled_pitched(packed, geo.neighbor_ids, t1, C, D, H, W, wp))
                line.append(f"ds{ds}: {tc:.3f}/{tp:.3f}")
            print("   contiguous/pitched ms  " + "  ".join(line), flush=True)
        _lib.set_option("sweep_dsplit", 0)
        t0 = ops.plane_sweep_table(geo.proj_rel, geo.depth_values, H, W)
        t1 = ops.plane_sweep_table_pitched(geo.proj_rel, geo.depth_values, H, W, wp)
        print(f"{name}: contiguous {ta:.3f} ms ({gb / ta:.0f} GB/s, frac {gb / ta / 8:.3f}; geometry {tt0:.3f})   pitched {wp}: {tb:.3f} ms "
              f"({gb / tb:.0f} GB/s, frac {gb / tb / 8:.3f}; geometry {tt1:.3f})   bit-identical: {same}", flush=True)


if __name__ == "__main__":
    main()
