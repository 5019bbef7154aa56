#!/usr/bin/env python3
"""Randomised soak of the plane sweep against the oracle (not collected by pytest; run on the GPU box):
    python tests/fuzz_sweep.py [cases] [first_seed]
Random shapes (N, K, C, D, H, W), random projective maps of the kinds tests/test_gpu_parity.py::test_sweep_random uses, random
tuning options.  Forward: bit for bit (NaN where the oracle has NaN).  Backward: 1e-4 of the gradient's scale."""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np  # noqa: E402
import torch  # noqa: E402

from mvsdet_amd import _lib, ops  # noqa: E402
from oracle import oracle as O  # noqa: E402


def make_case(seed):
    rng = np.random.default_rng(50000 + seed)
    N = int(rng.integers(2, 7))
    K = int(rng.integers(0, min(4, N - 1) + 1))
    C = int(rng.choice([1, 5, 16, 31, 32, 33, 64, 70, 96]))
    D = int(rng.integers(1, 14))
    H, W = int(rng.integers(2, 41)), int(rng.integers(2, 101))
    feat = rng.standard_normal((N, C, H, W)).astype(np.float32)
    nbr = (np.stack([rng.permutation([j for j in range(N) if j != n] * 4)[:K] for n in range(N)]).astype(np.int64)
           if K else np.zeros((N, 0), np.int64))
    proj = np.tile(np.eye(4, dtype=np.float32), (N, max(K, 1), 1, 1))[:, :K]
    for n in range(N):
        for j in range(K):
            kind = rng.integers(0, 7)
            A, t = np.eye(3, dtype=np.float32), np.zeros(3, dtype=np.float32)
            if kind == 0:
                A[:2, :2] += rng.normal(0, 0.05, (2, 2)); t[:2] = rng.normal(0, 3.0, 2)
            elif kind == 1:
                t[:2] = rng.choice([-1, 1], 2) * rng.uniform(0.5, 2.5, 2) * np.array([W, H])
            elif kind == 2:
                A[:2, :2] *= rng.uniform(1.5, 6.0)
            elif kind == 3:
                A[2, :2] = rng.normal(0, 0.05, 2); t[2] = rng.normal(0, 0.5)
            elif kind == 4:
                A[:2, :2] *= rng.uniform(0.05, 0.5); t[:2] = rng.uniform(0, 1, 2) * np.array([W, H])
            elif kind == 5:
                a = rng.uniform(-0.6, 0.6)   # in-plane rotation (a rolled camera)
                A[:2, :2] = np.array([[np.cos(a), -np.sin(a)], [np.sin(a), np.cos(a)]], np.float32) * rng.uniform(0.8, 1.25)
                t[:2] = rng.normal(0, 0.3, 2) * np.array([W, H])
            else:
                A[2, 2] = 0.0; t[2] = 0.0 if rng.random() < 0.5 else 1.0
            proj[n, j, :3, :3] = A
            proj[n, j, :3, 3] = t
    depth = np.sort(rng.uniform(0.2, 5.0, (1, D)).astype(np.float32), axis=1).repeat(N, 0)
    opts = {"sweep_tw": int(rng.choice([0, 0, 16, 32])), "sweep_boxcap": int(rng.choice([512, 512, 0, 24, 100, 200])),
            "sweep_xcd": int(rng.choice([0, 1])), "bwd_groups": int(rng.choice([0, 1, 2]))}
    return (N, K, C, D, H, W), feat, nbr, proj, depth, opts


def main():
    cases = int(sys.argv[1]) if len(sys.argv) > 1 else 100
    first = int(sys.argv[2]) if len(sys.argv) > 2 else 0
    dev = torch.device("cuda:0")
    O.build()
    saved = {k: _lib.get_option(k) for k in ("sweep_tw", "sweep_boxcap", "sweep_xcd", "bwd_groups")}
    bad = 0
    for seed in range(first, first + cases):
        shape, feat, nbr, proj, depth, opts = make_case(seed)
        for k, v in opts.items():
            _lib.set_option(k, v)
        args = (torch.from_numpy(nbr).to(dev), torch.from_numpy(proj).to(dev), torch.from_numpy(depth).to(dev))
        ref = O.plane_sweep_variance(torch.from_numpy(feat), nbr, proj, depth, mode=1)
        out = ops.plane_sweep_variance(torch.from_numpy(feat).to(dev), *args).cpu().numpy()
        ok_f = np.array_equal(np.isnan(out), np.isnan(ref)) and np.array_equal(out[~np.isnan(ref)], ref[~np.isnan(ref)])
        ok_b = True
        if not np.isnan(ref).any() and shape[1] > 0:
            g = np.random.default_rng(seed).standard_normal(ref.shape).astype(np.float32)
            gref = O.plane_sweep_variance_bwd(feat, nbr, proj, depth, torch.from_numpy(g))
            got = ops.plane_sweep_variance_backward(torch.from_numpy(feat).to(dev), *args, torch.from_numpy(g).to(dev)).cpu().numpy()
            ok_b = bool(np.abs(got - np.asarray(gref)).max() <= 1e-4 * max(1.0, float(np.abs(np.asarray(gref)).max())))
        if not (ok_f and ok_b):
            bad += 1
            print(f"seed {seed} shape {shape} opts {opts}: forward {'ok' if ok_f else 'MISMATCH'}, backward {'ok' if ok_b else 'MISMATCH'}", flush=True)
        elif seed % 20 == 0:
            print(f"seed {seed} shape {shape} opts {opts}: ok", flush=True)
    for k, v in saved.items():
        _lib.set_option(k, v)
    print(f"{cases} cases, {bad} mismatches")
    return 1 if bad else 0


if __name__ == "__main__":
    sys.exit(main())
