#!/usr/bin/env python3
"""Randomised soak of stages 2 and 3 (depth distribution + top-k; depth-weighted lifting with the view mean) against the
oracle (not collected by pytest; run on the GPU box):  python tests/fuzz_stages.py [cases] [first_seed]
Stage 2: prob / off / depth expectation to 2e-6, candidates exact where the probabilities are separated by more than 1e-6.
Stage 3: voxel volume and valid counts bit for bit."""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np  # noqa: E402
import torch  # noqa: E402

from mvsdet_amd import ops, synthetic  # noqa: E402
from mvsdet_amd.hotpath import MVSDetHotPath  # noqa: E402
from oracle import oracle as O  # noqa: E402


def main():
    cases = int(sys.argv[1]) if len(sys.argv) > 1 else 60
    first = int(sys.argv[2]) if len(sys.argv) > 2 else 0
    dev = torch.device("cuda:0")
    O.build()
    bad = 0
    for seed in range(first, first + cases):
        rng = np.random.default_rng(70000 + seed)
        N, D = int(rng.integers(1, 7)), int(rng.integers(3, 70))
        H, W = int(rng.integers(4, 33)), int(rng.integers(4, 49))
        topk = int(rng.integers(1, min(3, D) + 1))
        near, far = 0.2, float(rng.uniform(2.0, 6.0))
        interval = (far - near) / D
        cost = (rng.standard_normal((N, D, H, W)) * rng.uniform(0.5, 4.0)).astype(np.float32)
        offl = rng.standard_normal((N, D, H, W)).astype(np.float32)
        r = O.depth_prob_topk(cost, offl, near, interval, topk)
        prob, off, ed, en, ei, av = ops.depth_prob_topk(torch.from_numpy(cost).to(dev), torch.from_numpy(offl).to(dev), near, interval, topk)
        ok2 = (np.abs(prob.cpu().numpy() - r["prob"]).max() <= 2e-6 and np.abs(off.cpu().numpy() - r["off"]).max() <= 2e-6
               and np.abs(av.cpu().numpy() - r["avg_depth"]).max() <= 2e-5 * far)
        srt = np.sort(r["prob"], axis=1)[:, ::-1]
        sep = (srt[:, :topk] - srt[:, 1:topk + 1]).min(axis=1) > 1e-6 if D > topk else np.ones((N, H, W), bool)
        ok2 = ok2 and np.array_equal(ei.cpu().numpy().transpose(0, 2, 3, 1)[sep], r["est_idx"].transpose(0, 2, 3, 1)[sep])
        # stage 3 on a small scene
        C = int(rng.choice([8, 32, 40]))
        hw = (int(rng.integers(8, 25)), int(rng.integers(8, 33)))
        nv = [int(rng.integers(4, 13)), int(rng.integers(4, 13)), int(rng.integers(2, 7))]
        hp = MVSDetHotPath(nv, [0.4, 0.4, 0.4], [0.2, 5.0], 8)
        meta = synthetic.make_img_meta(N + 1, hw, seed=seed, per_view_intrinsics=bool(rng.integers(0, 2)))
        feat = synthetic.make_features(N + 1, C, hw, seed=seed)
        geo = hp.prepare_scene(meta, dev)
        J = 3
        est_depth = rng.uniform(0.2, 5.0, (N + 1, J) + hw).astype(np.float32)
        est_dens = rng.uniform(0.0, 1.0, (N + 1, J) + hw).astype(np.float32)
        h, w = geo.height, geo.width
        fd = feat.to(dev)
        vol, valid = hp.lift(fd, ops.pack_features(fd), geo, torch.from_numpy(est_depth).to(dev), torch.from_numpy(est_dens).to(dev))
        ref = O.backproject_weigh_mean(feat[:, :, :h, :w], geo.points.cpu().numpy(), geo.projection.cpu().numpy(),
                                       est_depth[:, :, :h, :w], est_dens[:, :, :h, :w], hp.voxel_size[-1])
        ok3 = np.array_equal(vol.cpu().numpy().reshape(ref["volume_mean"].shape), ref["volume_mean"]) and \
            np.array_equal(valid.cpu().numpy().reshape(-1), ref["valid_count"].astype(np.int64))
        if not (ok2 and ok3):
            bad += 1
            print(f"seed {seed}: stage 2 {'ok' if ok2 else 'MISMATCH'} (N={N} D={D} {H}x{W} topk={topk}), stage 3 {'ok' if ok3 else 'MISMATCH'} (C={C} {hw} {nv})", flush=True)
        elif seed % 10 == 0:
            print(f"seed {seed}: ok", flush=True)
    print(f"{cases} cases, {bad} mismatches")
    return 1 if bad else 0


if __name__ == "__main__":
    sys.exit(main())
