#!/usr/bin/env python3
"""Benchmark of the MVSDet plane-sweep hot path on MI355X.

    python bench.py --gpus N --steps K --warmup W
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 --master-port P \
        bench.py --gpus N --steps K --warmup W

One step = one synthetic ScanNet-shaped scene through the hot path on one GPU: camera algebra on the host
(a1,a2,a8) -> pack -> plane-sweep variance cost volume (a3+a4) -> depth soft-max/top-k/expectation (a5-a7, on
stand-in CostRegNet logits) -> depth-weighted voxel lifting fused with the view mean (a9+a10).  Inputs are
resident in HBM before the timed region.  Scenes are independent, so N ranks each process their own scenes
(weak scaling, no collective on the data path); rank 0 prints ONE JSON line.

metric  = cost volumes / s  (one cost volume = one reference view's (C,D,H,W) variance volume), whole job.
roofline = algorithmic HBM bytes of the plane-sweep kernel / its mean duration from HIP events recorded inside
           the timed region, against the 8 TB/s HBM3E peak.
cpu_baseline = the CPU restatement of the same stage (oracle/) timed on this box's host cores on a bounded
           sample (rank 0, N=1 only).  It is a checker-side measurement: nothing shipped runs on it.
"""
import argparse
import gc
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

import numpy as np  # noqa: E402
import torch  # noqa: E402

WORKLOADS = {
    # BASELINE.json configs[1] as worded (the shape the metric is quoted on): SURVEY.md section 8 "B2"
    "scannet_40v_64d_120x160": dict(N=40, C=256, D=64, H=120, W=160, near_far=(0.2, 5.0), per_view_K=False),
    # what the shipped config mvsdet_res50_2x_low_res.py really runs: "R"
    "scannet_ref_40v_12d_60x80": dict(N=40, C=256, D=12, H=60, W=80, near_far=(0.2, 5.0), per_view_K=False),
    # BASELINE.json configs[3]
    # BASELINE.json configs[3]: per-view intrinsics and the larger voxel grid of SURVEY 8d C4 (the shipped ARKit
    # config keeps 40x40x16: SURVEY D6; 64x64x24 is the builder-defined "larger grid" the config line asks for)
    "arkit_50v_96d_60x80": dict(N=50, C=256, D=96, H=60, W=80, near_far=(0.5, 5.5), per_view_K=True, voxels=[64, 64, 24]),
    # BASELINE.json configs[4] at C=32 (fp32, unchunked: 126 GB cost volume; the C=256 fp16 form needs view chunks)
    "stress_100v_128d_240x320_c32": dict(N=100, C=32, D=128, H=240, W=320, near_far=(0.2, 5.0), per_view_K=False),
    # BASELINE.json configs[4] as worded: C=256, fp16 features and fp16 cost volume (503 GB) produced in chunks of
    # 10 reference views (50 GB each, consumed and released before the next chunk)
    "stress_100v_128d_240x320_c256_f16": dict(N=100, C=256, D=128, H=240, W=320, near_far=(0.2, 5.0), per_view_K=False,
                                              half=True, chunk=10),
    # what tools/test.py really runs (VERDICT r4 missing #1): configs/mvsdet_res50_2x_low_res.py:105-126 evaluates with
    # n_images = 81 (80 source views; fewer after the de-duplication of multiview_pipeline.py:432-441), mvsdet_arkit.py:114 with
    # 101 (100 views, per-view intrinsics, near/far 0.5-5.5); 12 planes, 60 x 80 maps, the 40 x 40 x 16 grid
    "scannet_test_80v_12d_60x80": dict(N=80, C=256, D=12, H=60, W=80, near_far=(0.2, 5.0), per_view_K=False),
    "arkit_test_100v_12d_60x80": dict(N=100, C=256, D=12, H=60, W=80, near_far=(0.5, 5.5), per_view_K=True, arkit_head=True),
    # BASELINE.json configs[0] (plumbing)
    "tiny_3v_8d_48x64": dict(N=3, C=32, D=8, H=48, W=64, near_far=(0.2, 5.0), per_view_K=False),
}
N_VOXELS, VOXEL_SIZE = [40, 40, 16], [0.16, 0.16, 0.2]
SWEEP_KERNEL_NAME = "plane_sweep_variance_kernel<K,TW,FAST,OutT>"   # K neighbours, tile width, unconditional-store form, float or _Float16
HBM_PEAK_GBPS = 8000.0  # MI355X_MICROARCH.md: 8.0 TB/s spec (6.29 TB/s measured float4 copy)


def sweep_bytes_per_cv(w, K=2):
    """(K+1)*C*H*W*b read + C*D*H*W*b written per cost volume (b = 4 fp32, 2 fp16)."""
    b = 2 if w.get("half") else 4
    return (K + 1) * w["C"] * w["H"] * w["W"] * b + w["C"] * w["D"] * w["H"] * w["W"] * b


class SceneInputs:
    def __init__(self, w, seed, device):
        from mvsdet_amd import synthetic
        hw = (w["H"], w["W"])
        self.meta = synthetic.make_img_meta(w["N"], hw, seed=seed, per_view_intrinsics=w["per_view_K"])
        self.features = synthetic.make_features(w["N"], w["C"], hw, seed=seed, device=device)
        if w.get("half"):
            self.features = self.features.half()
        self.cost_logits = synthetic.make_cost_logits(w["N"], w["D"], hw, seed=seed, device=device)


def unseen_metas(w, rank, count):
    """One img_meta per step, each with cameras no earlier step had (own seed): the reference samples the views of a
    scene anew for every item (multiview_pipeline.py:141) and RandomShiftOrigin moves the origin, so a real pipeline
    never presents the same camera bytes twice and the hot path's content cache never hits.  Built before the timed
    region -- in the reference this is the data loader's work, not extract_feat's."""
    from mvsdet_amd import synthetic
    return [synthetic.make_img_meta(w["N"], (w["H"], w["W"]), seed=rank * 100003 + 1000 + i,
                                    per_view_intrinsics=w["per_view_K"]) for i in range(count)]


def run_gpu(args, w, rank, world, device):
    from mvsdet_amd import ops, parallel
    from mvsdet_amd.hotpath import MVSDetHotPath

    hp = MVSDetHotPath(w.get("voxels", N_VOXELS), VOXEL_SIZE, list(w["near_far"]), w["D"], topk=3)
    scenes = [SceneInputs(w, seed=rank * 100 + i, device=device) for i in range(args.scene_pool)]
    # feature maps and stand-in logits cycle through a small resident pool (they are the 2-D backbone's output, resident
    # in HBM when extract_feat's hot block starts); the CAMERAS of every step are new, so the host geometry a1/a2/a8
    # (neighbour selection, projections, voxel points) and its upload run for every timed step.  What may hide them is
    # the one-step-ahead prefetch on the worker thread, as a data loader allows -- never the content cache.
    metas = unseen_metas(w, rank, args.warmup + args.steps + 1)
    torch.cuda.synchronize(device)
    ev = []

    def step_chunked(i, timed):
        """Cost volume produced in chunks of reference views (fp16 storage); every launch is one chunk."""
        s = scenes[i % len(scenes)]
        hp.prefetch_scene(metas[i + 1], device)
        geo = hp.prepare_scene(metas[i], device)
        packed = ops.pack_features(s.features)
        keep = None
        for first, var in hp.cost_volume_chunks(packed, geo, w["C"], w["H"], w["W"], w["chunk"], half_out=w.get("half", False),
                                                events=ev if timed else None):
            keep = var[0, 0, 0].float().abs().sum().reshape(1, 1, 1)  # the consumer stand-in touches the chunk, then drops it
            del var
        prob, off, est_depth, est_dens, est_idx, avg = hp.depth_distribution(s.cost_logits)
        vol, valid = hp.lift_packed(packed, geo, est_depth, est_dens, w["C"], w["H"], w["W"])
        return keep, vol, valid

    def step(i, timed):
        if w.get("chunk"):
            return step_chunked(i, timed)
        s = scenes[i % len(scenes)]
        feat = s.features
        hp.prefetch_scene(metas[i + 1], device)
        geo = hp.prepare_scene(metas[i], device)
        # the product's own sequence (MVSDetHotPath.forward_scene): geometry kernel on a side stream beside the packing.
        # HIP events on the streams the kernels launch on: table = (t0, t1) on the side stream, slab kernel = (em, e1)
        tab = hp.sweep_geometry_async(geo, w["H"], w["W"], events=timed)
        packed = ops.pack_features(feat)
        if timed:
            em, e1 = (torch.cuda.Event(enable_timing=True) for _ in range(2))
            torch.cuda.current_stream(device).wait_event(tab[1])
            em.record()
        var = hp.cost_volume_tabled(packed, geo, tab[:2], w["C"], w["H"], w["W"])
        if timed:
            e1.record()
            ev.append((tab[2], tab[3], em, e1))
        prob, off, est_depth, est_dens, est_idx, avg = hp.depth_distribution(s.cost_logits)
        vol, valid = hp.lift(feat, packed, geo, est_depth, est_dens)
        return var, vol, valid

    barrier = parallel.barrier if world > 1 else (lambda: None)

    out = None
    collect_garbage()   # AHEAD of the warm-up: 100 ms of host work between warm-up and timing let the device clock down (first timed step 13.0 instead of 10.6 ms)
    for i in range(args.warmup):
        out = None
        out = step(i, False)
    del out
    barrier()
    torch.cuda.synchronize(device)
    stats0 = dict(hp._geometry.stats)
    t0 = time.perf_counter()
    out = None
    for i in range(args.warmup, args.warmup + args.steps):
        out = None  # release the previous cost volume before the next one is allocated (126 GB at the stress shape)
        out = step(i, True)
    torch.cuda.synchronize(device)
    barrier()
    elapsed = time.perf_counter() - t0
    hp.geometry_stats = {k: hp._geometry.stats[k] - stats0[k] for k in stats0}   # of the timed region
    checksum = float(out[1].abs().sum().item()) + float(out[0][0, 0, 0].abs().sum().item())
    del out
    # (table ms, slab-kernel ms) per launch; the chunked workload's shard entry point enqueues both behind one pair
    hp.sweep_ms_each = [m.elapsed_time(e) for _, _, m, e in ev] if ev and len(ev[0]) == 4 else [a.elapsed_time(b) for a, b in ev]
    if ev and len(ev[0]) == 4:
        sweep_ms = (float(np.mean([a.elapsed_time(b) for a, b, _, _ in ev])), float(np.mean([m.elapsed_time(e) for _, _, m, e in ev])))
    elif ev:
        sweep_ms = (0.0, float(np.mean([a.elapsed_time(b) for a, b in ev])))
    else:
        sweep_ms = (float("nan"), float("nan"))
    if world > 1:
        elapsed = parallel.max_over_ranks(elapsed, device)
    return elapsed, sweep_ms, checksum, hp, scenes


def stage_breakdown(w, hp, scene, device, reps=3):
    """Per-stage HIP-event timings of one scene (reported as extras; not the headline).  Stage 1 is timed on the SAME entry points
    and streams as the timed loop (`sweep_geometry_async` on its side stream, `cost_volume_tabled` behind its event), the scenes of
    the repetitions enqueued back to back without a host synchronisation in between (as the loop runs them; the first one, which
    grows the allocator's pools and starts from an idle device, is not counted) -- so `plane_sweep_variance` here and
    `roofline.kernel_ms` are one measurement taken twice.  The host part (camera algebra + upload, which needs a synchronisation to
    be timed at all) is timed separately, before."""
    from mvsdet_amd import ops
    names = ["host_prep+h2d (serial here; prefetched one step ahead in the timed loop)", "pack", "plane_sweep_geometry (side stream, beside the packing)",
             "plane_sweep_variance", "depth_prob_topk", "backproject_mean"]
    acc = {n: [] for n in names}
    geos = []
    for rep in range(reps + 1):
        torch.cuda.synchronize(device)
        t0 = time.perf_counter()
        fresh = dict(scene.meta)                              # jittered cameras: not served from the content cache
        fresh["lidar2img"] = dict(scene.meta["lidar2img"], extrinsic=[e + np.float32(1e-6) * (rep + 1) for e in scene.meta["lidar2img"]["extrinsic"]])
        geos.append(hp.prepare_scene(fresh, device))
        torch.cuda.synchronize(device)
        if rep:
            acc[names[0]].append((time.perf_counter() - t0) * 1e3)
    marks = []
    for rep in range(reps + 1):
        geo = geos[rep]
        es = [torch.cuda.Event(enable_timing=True) for _ in range(6)]
        tab = hp.sweep_geometry_async(geo, w["H"], w["W"], events=True)
        es[0].record()
        packed = ops.pack_features(scene.features)
        es[1].record()
        torch.cuda.current_stream(device).wait_event(tab[1])
        es[2].record()
        var = hp.cost_volume_tabled(packed, geo, tab[:2], w["C"], w["H"], w["W"])
        es[3].record()
        prob, off, ed, en, ei, avg = hp.depth_distribution(scene.cost_logits)
        es[4].record()
        vol, valid = hp.lift(scene.features, packed, geo, ed, en)
        es[5].record()
        del var                                               # the next scene's volume takes this one's block (stream order)
        marks.append((es, tab))
    torch.cuda.synchronize(device)
    for es, tab in marks[1:]:                                 # the first pass grows the allocator's pools
        acc[names[1]].append(es[0].elapsed_time(es[1]))
        acc[names[2]].append(tab[2].elapsed_time(tab[3]))
        acc[names[3]].append(es[2].elapsed_time(es[3]))
        acc[names[4]].append(es[3].elapsed_time(es[4]))
        acc[names[5]].append(es[4].elapsed_time(es[5]))
    return {n: round(float(np.median(v)), 4) for n, v in acc.items()}


class PointwiseCostReg(torch.nn.Module):
    """Trainable stand-in for CostRegNet_3DGS: (N,C,D,H,W) -> (N,2,D,H,W) as one thin GEMM per view (2 x C weights), so
    that neither MIOpen's 3-D convolutions nor their tuning runs are part of the measurement."""

    def __init__(self, channels):
        super().__init__()
        self.weight = torch.nn.Parameter(torch.randn(2, channels) / channels ** 0.5)
        self.bias = torch.nn.Parameter(torch.zeros(2))

    def forward(self, var):
        n, c, d, h, w = var.shape
        # bmm, not matmul: matmul folds the batch into the rows of ONE (n*L, C) GEMM, whose backward hands dL/dvar back
        # in (n, L, C) order -- two strided 2.4 GB copies (9.5 ms) before the sweep's backward sees a contiguous gradient
        out = torch.bmm(self.weight.unsqueeze(0).expand(n, -1, -1), var.reshape(n, c, d * h * w)) + self.bias.view(1, 2, 1)
        return out.view(n, 2, d, h, w)


# untimed scenes ahead of a chain measurement.  The caching allocator's pool keeps growing for a few scenes (a block released while
# another stream still reads it cannot be handed out again yet: the next request of that size is a hipMalloc; 4-8 new segments per ten
# scenes after two warm-up scenes, 4-5 after four: `device_allocations_in_timed_loop`), and when the block is the 2.4 GB variance volume
# the one scene that pays for it takes 29.9 ms instead of 10.8 -- 12.7-13.8 ms per scene in the mean of ten, seen in two of a dozen
# runs.  `ms_per_scene_min_median_max`, `ms_per_scene_sequence` and `scenes_per_sec_at_the_median_scene` tell such a run from a slow chain
CHAIN_WARMUP = 4


def collect_garbage():
    """Called once ahead of the WARM-UP of every timed region (between warm-up and timing its 100 ms of host work let the device
    clock down: the first timed step of the headline loop took 13.0 instead of 10.6 ms, 3 % of the value).  CPython's full (generation-2) collection walks every object torch's import
    created: 100-130 ms on this image, measured (MVSDET_BENCH_TRACE=1 prints the per-step host times).  When it falls is a matter
    of allocation counts, so it landed in the first of ten timed training steps of one code version and in the warm-up of
    another: a 5.5 ms step read as 12.8-21.9 ms.  A long-running job pays it once per many thousand steps; a 10-step
    measurement must not.  freeze() then keeps what exists now out of later collections -- after an unfreeze(), or the networks
    of the PREVIOUS measurement (frozen while alive, unreachable since, and cyclic through their hooks) would never be collected:
    3.8 GB of device memory per measurement."""
    gc.unfreeze()
    gc.collect()
    gc.freeze()


def run_train(args, w, rank, world, device):
    """BASELINE.json configs[2]: training step of the hot path -- forward a1..a10, loss, backward through the custom
    ops' autograd, optimiser step -- with a trainable stand-in for CostRegNet_3DGS (PointwiseCostReg) wrapped in DistributedDataParallel when N > 1: the gradient all-reduce
    over RCCL/xGMI is the only collective, exactly as in the reference's DDP training."""
    from mvsdet_amd import parallel
    from mvsdet_amd.hotpath import MVSDetHotPath
    torch.manual_seed(0)
    if args.with_cost_network:   # the real 3-D U-Net (forward / dX / dW of its big layers on our kernels) instead of the stand-in
        from mvsdet_amd.costreg import CostRegNet3DGS
        net = CostRegNet3DGS(w["C"]).to(device).train()
    else:
        net = PointwiseCostReg(w["C"]).to(device)
    # find_unused_parameters=True as the reference sets it (configs/mvsdet_res50_2x_low_res_depth.py:200: three of the four FPN
    # outputs take no part in the loss); here every parameter of the cost network is used, the flag only costs its graph walk
    model = torch.nn.parallel.DistributedDataParallel(net, device_ids=None, find_unused_parameters=True) if world > 1 else net
    # the optimiser of the training config (mvsdet_res50_2x_low_res_depth.py:179-184): AdamW lr 2e-4, weight decay 1e-4, and
    # clip_grad max_norm 35 -- a global-norm reduction over every gradient whose result mmengine's OptimWrapper reads on the HOST
    # every step (`float(grad)` for the 'train/grad_norm' scalar): one device synchronisation per step, kept here
    opt = torch.optim.AdamW(net.parameters(), lr=2e-4, weight_decay=1e-4)
    params = [p for p in net.parameters() if p.requires_grad]
    opt_events = []
    hp = MVSDetHotPath(N_VOXELS, VOXEL_SIZE, list(w["near_far"]), w["D"], topk=3, cost_regularization=model)
    scenes = [SceneInputs(w, seed=rank * 100 + i, device=device) for i in range(args.scene_pool)]
    metas = unseen_metas(w, rank, args.warmup + args.steps + 1)   # new cameras every step; index 0..warmup-1 = warm-up

    def step(i):
        s = scenes[i % len(scenes)]
        feat = s.features.detach().requires_grad_(True)   # the 2-D backbone's output: receives dL/dfeat
        hp.prefetch_scene(metas[args.warmup + i + 1], device)
        out = hp.forward_scene(feat, metas[args.warmup + i])
        loss = out["volume"].square().mean() + out["depth_coding"].mean() + out["est_densities"].mean()
        opt.zero_grad(set_to_none=True)
        loss.backward()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        grad_norm = float(torch.nn.utils.clip_grad_norm_(params, max_norm=35.0, norm_type=2))   # the host read mmengine makes
        opt.step()
        e1.record()
        if i >= 0:
            opt_events.append((e0, e1, grad_norm))
        return float(feat.grad.abs().sum().item()) if i == args.steps - 1 else 0.0

    barrier = parallel.barrier if world > 1 else (lambda: None)
    collect_garbage()
    for i in range(args.warmup):
        step(-1 - i)
    barrier()
    torch.cuda.synchronize(device)
    t0 = time.perf_counter()
    checksum = 0.0
    trace = []
    for i in range(args.steps):
        ta = time.perf_counter()
        checksum += step(i)
        if os.environ.get("MVSDET_BENCH_TRACE") == "2":
            torch.cuda.synchronize(device)
        trace.append(round((time.perf_counter() - ta) * 1e3, 2))
    torch.cuda.synchronize(device)
    barrier()
    elapsed = time.perf_counter() - t0
    if os.environ.get("MVSDET_BENCH_TRACE"):
        print("step host ms:", trace, "total", round(elapsed * 1e3, 1), file=sys.stderr, flush=True)
    if world > 1:
        elapsed = parallel.max_over_ranks(elapsed, device)
    # clip + AdamW between two HIP events per step (the events bracket the host read of the norm too: it is device time waited for)
    args.optimizer_ms = round(float(np.mean([a.elapsed_time(b) for a, b, _ in opt_events])), 4) if opt_events else None
    args.grad_norm_last = opt_events[-1][2] if opt_events else None
    return elapsed, checksum


OPTIMIZER = "AdamW+clip35"   # lr 2e-4, weight decay 1e-4, clip_grad max_norm 35 with its per-step host read (mvsdet_res50_2x_low_res_depth.py:179-184)
# the matrix route of the real cost network under autograd, stated wherever its step time is quoted
BF16X3_GRADIENT_TOLERANCE = ("gradients of the bf16x3 training route against the reference module's: element-wise 1e-3 of each tensor's "
                             "scale on a margin input, 1e-4 in norm (G12b); the fp32 route holds 1e-4 element-wise (G12)")


def backward_rooflines(w, device, reps=5):
    """HIP-event timings of the two kernels a training step of configs[2] spends most of its time in, at workload `w`:
    the backward of the plane sweep (mvsdet.py:439-467 under autograd) against HBM, and the weight gradient of the cost
    network's first layer (mvsnet.py:76) on the bf16 matrix cores against the dense bf16 peak.  Events are recorded on the
    current stream, which is the stream both operators launch on."""
    from mvsdet_amd import ops
    from mvsdet_amd.hotpath import MVSDetHotPath
    hp = MVSDetHotPath(N_VOXELS, VOXEL_SIZE, list(w["near_far"]), w["D"], topk=3)
    s = SceneInputs(w, seed=5, device=device)
    geo = hp.prepare_scene(s.meta, device)
    N, C, D, H, W = w["N"], w["C"], w["D"], w["H"], w["W"]
    gvar = torch.randn((N, C, D, H, W), device=device)

    def timed(fn):
        fn()
        ts = []
        for _ in range(reps):
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            fn()
            e1.record()
            torch.cuda.synchronize(device)
            ts.append(e0.elapsed_time(e1))
        return float(np.mean(ts)), float(np.min(ts))

    # the operator as a training step runs it since round 5: the packed maps and the sweep geometry are the FORWARD pass's
    # (ops.plane_sweep_variance_keep hands them out), the backward is memset + kernel + unpack; the stand-alone entry point that packs
    # and builds the geometry itself is timed beside it
    with torch.no_grad():
        _, packed, table = ops.plane_sweep_variance_keep(s.features, geo.neighbor_ids, geo.proj_rel, geo.depth_values)
    bwd_ms, bwd_min = timed(lambda: ops.plane_sweep_variance_backward_packed(packed, geo.neighbor_ids, table, gvar))
    alone_ms, _ = timed(lambda: ops.plane_sweep_variance_backward(s.features, geo.neighbor_ids, geo.proj_rel, geo.depth_values, gvar))
    del packed, table
    # dL/dvar is read once, the features once, dL/dfeat written once (float atomics on a packed copy, then unpacked)
    bwd_bytes = N * C * D * H * W * 4 + 2 * N * C * H * W * 4
    out = {"backward_sweep": {"bound": "hbm", "achieved": round(bwd_bytes / (bwd_ms * 1e-3) / 1e9, 1), "peak": HBM_PEAK_GBPS,
                              "unit": "GB/s", "frac": round(bwd_bytes / (bwd_ms * 1e-3) / 1e9 / HBM_PEAK_GBPS, 4),
                              "kernel": "plane_sweep_variance_bwd_kernel<K,TW,HALF> (+ memset and unpack of the packed gradient: the whole "
                                        "operator between two HIP events, as the training step runs it -- packed maps and geometry kept "
                                        "from the forward pass)",
                              "kernel_ms": round(bwd_ms, 4), "min_ms": round(bwd_min, 4), "algorithmic_bytes_per_launch": bwd_bytes,
                              "standalone_entry_ms": round(alone_ms, 4),   # packs the features and builds the geometry itself
                              "bytes_formula": "N*C*D*H*W*4 (dL/dvar read once) + 2*N*C*H*W*4 (features read, gradient written)"}}
    x = gvar                                             # the variance-shaped input of conv0 (values do not matter to the timing)
    gy = torch.randn((N, 64, D, H, W), device=device)
    dw_ms, dw_min = timed(lambda: ops.conv3d_k3_dw(x, gy, 0, 1, True))
    fl = 2.0 * 27 * C * 64 * N * D * H * W
    out["weight_gradient_conv0"] = {"bound": "mfma", "achieved": round(3 * fl / (dw_ms * 1e-3) / 1e12, 1), "peak": 2500.0,
                                    "unit": "TFLOP/s", "frac": round(3 * fl / (dw_ms * 1e-3) / 1e12 / 2500.0, 4),
                                    "kernel": "conv3d_k3_dw_bf16x3_kernel (conv0 256 -> 64; 3 bf16 MFMAs per fp32-equivalent product)",
                                    "kernel_ms": round(dw_ms, 4), "min_ms": round(dw_min, 4), "useful_TFLOPs": round(fl / (dw_ms * 1e-3) / 1e12, 1)}
    del gvar, gy, x
    torch.cuda.empty_cache()
    # the stride-2 / transposed layers' weight gradient (mvsnet.py:77,92-100) at conv1's shape: fine 64 channels at full
    # resolution, coarse 128 channels at half of it (conv11 is the same call with the tensors exchanged)
    if D % 2 == 0 and H % 2 == 0 and W % 8 == 0:
        xf = torch.randn((N, 64, D, H, W), device=device)
        gc = torch.randn((N, 128, D // 2, H // 2, W // 2), device=device)
        s2_ms, s2_min = timed(lambda: ops.conv3d_k3_dw(xf, gc, 0, 2, True))
        fl2 = 2.0 * 27 * 64 * 128 * N * (D // 2) * (H // 2) * (W // 2)
        out["weight_gradient_stride2"] = {"bound": "mfma", "achieved": round(3 * fl2 / (s2_ms * 1e-3) / 1e12, 1), "peak": 2500.0,
                                          "unit": "TFLOP/s", "frac": round(3 * fl2 / (s2_ms * 1e-3) / 1e12 / 2500.0, 4),
                                          "kernel": "conv3d_k3_s2_dw_bf16x3_kernel (conv1 64 -> 128 / conv11 128 -> 64; 64 coarse x 16 fine "
                                                    "channels per block on v_mfma_f32_16x16x32_bf16)",
                                          "kernel_ms": round(s2_ms, 4), "min_ms": round(s2_min, 4),
                                          "useful_TFLOPs": round(fl2 / (s2_ms * 1e-3) / 1e12, 1)}
        del xf, gc
        torch.cuda.empty_cache()
    return out


def training_block(device):
    """BASELINE.json configs[2] at N=1 in the driver's default line: a training step of the hot path at the shape the shipped
    config trains on (tools/train.py:76-153; mvsdet_res50_2x_low_res.py:129: one scene per GPU), with the stand-in and with
    the real cost network, new cameras every step, plus the rooflines of its two dominant kernels."""
    wr = WORKLOADS["scannet_ref_40v_12d_60x80"]
    out = {"workload": "scannet_ref_40v_12d_60x80",
           "step": "fwd a1..a10 + loss + bwd through the ops' autograd + clip_grad_norm_(35) with its host read + AdamW step, "
                   "new cameras every step (prefetch only)"}
    for key, real, steps in (("stand_in_cost_network", False, 10), ("real_cost_network", True, 5)):
        a = argparse.Namespace(steps=steps, warmup=2, scene_pool=2, with_cost_network=real)
        el, _ = run_train(a, wr, 0, 1, device)
        out[key] = {"ms_per_step": round(el / steps * 1e3, 3), "scenes_per_sec": round(steps / el, 3), "steps": steps,
                    "optimizer": OPTIMIZER, "optimizer_ms": a.optimizer_ms,
                    "optimizer_share_of_step": round(a.optimizer_ms / (el / steps * 1e3), 4)}
        if real:
            out[key]["gradient_tolerance"] = BF16X3_GRADIENT_TOLERANCE
        torch.cuda.empty_cache()
    out["roofline"] = backward_rooflines(wr, device)
    return out


class ShapeOnlyStages:
    """--launch-check of the view-sharded mode on a machine without a GPU: the device stages replaced by zero tensors of
    the right shapes, so that the rendezvous, the ONE all-gather of the feature shards and the ONE all-reduce of the
    (C+1, X*Y*Z) voxel buffer (parallel.forward_scene_view_sharded) run over the chosen backend with their real sizes."""

    def __init__(self, hp):
        self.hp = hp

    def pack(self, feature):
        return feature

    def cost_volume_shard(self, packed, geo, first, count, n_src, C, H, W):
        return packed.new_zeros((count, 1, 1, 1, 1))

    def depth_distribution(self, cost_logits):
        n, _, d, h, w = cost_logits.shape
        z = cost_logits.new_zeros
        return z((n, d, h, w)), z((n, d, h, w)), z((n, 3, h, w)), z((n, 3, h, w)), None, z((n, h, w))

    def lift_sum_shard(self, packed, geo, est_depth, est_dens, first, count, n_src, C, H, W):
        v = self.hp.n_voxels[0] * self.hp.n_voxels[1] * self.hp.n_voxels[2]
        return packed.new_full((C, v), float(count)), torch.full((v,), count, dtype=torch.int32)


def run_view_sharded(args, w, rank, world, device, dry=False):
    """ONE scene over all ranks (the intra-scene split): every rank holds the feature maps of its own contiguous
    shard of the N reference views (as if its 2-D backbone had produced them), one all-gather gives every rank all N maps,
    stages 1-2 (and the cost network with --with-cost-network) run on the local views, and one all-reduce(SUM) of the
    (C+1, X*Y*Z) voxel buffer completes a9/a10.  Strong scaling of a single scene: value = scenes/s of the whole job."""
    from mvsdet_amd import parallel, synthetic
    from mvsdet_amd.hotpath import MVSDetHotPath
    net = None
    if args.with_cost_network and not dry:
        from mvsdet_amd.costreg import CostRegNet3DGS
        torch.manual_seed(0)
        net = CostRegNet3DGS(w["C"]).to(device).eval()
    hp = MVSDetHotPath(w.get("voxels", N_VOXELS), VOXEL_SIZE, list(w["near_far"]), w["D"], topk=3, cost_regularization=net)
    hw = (w["H"], w["W"])
    first, count = parallel.view_shard(w["N"], rank, world)
    pool = []
    for i in range(args.scene_pool):   # every rank draws the same scene and keeps its own views
        feat = synthetic.make_features(w["N"], w["C"], hw, seed=i, device=device)[first:first + count].contiguous()
        logits = None if net is not None else synthetic.make_cost_logits(w["N"], w["D"], hw, seed=i, device=device)
        pool.append((feat, logits))
    metas = unseen_metas(w, 0, args.warmup + args.steps + 1)   # the same cameras on every rank, new ones every step
    stages = ShapeOnlyStages(hp) if dry else None

    def step(i):
        feat, logits = pool[i % len(pool)]
        if not dry:
            hp.prefetch_scene(metas[i + 1], device)
        with torch.no_grad():
            out = parallel.forward_scene_view_sharded(hp, feat, metas[i], cost_logits=logits, features_are_local=True,
                                                      stages=stages)
        return out

    out = None
    collect_garbage()
    for i in range(args.warmup):
        out = step(i)
    parallel.barrier()
    if device.type == "cuda":
        torch.cuda.synchronize(device)
    t0 = time.perf_counter()
    for i in range(args.warmup, args.warmup + args.steps):
        out = step(i)
    if device.type == "cuda":
        torch.cuda.synchronize(device)
    parallel.barrier()
    elapsed = parallel.max_over_ranks(time.perf_counter() - t0, device if device.type == "cuda" else None)
    return elapsed, float(out["volume"].abs().sum().item()), int(out["valid"].max().item())


def settle_overlap_route(hp, scene, metas, device):
    """`overlap_detector = "auto"`: run scenes until the driver has measured its routes for this shape and kept one (untimed:
    a long-running job pays these ~25 scenes once per scene shape), plus two watch windows of a kept side route.
    -> (route, {route: ms per scene})."""
    hp.overlap_detector = "auto"
    need = len(hp.OVERLAP_ROUTES) * (hp._TUNE_WARM + hp._TUNE_SPAN + 1)
    out = None
    for i in range(need + 2):   # as the timed loops run: the next scene's cameras announced one scene ahead (their upload and geometry
        hp.prefetch_scene(metas[(i + 1) % len(metas)], device)   # kernel take a stream of their own -- one more hardware queue in the
        out = hp.forward_scene(scene.features, metas[i % len(metas)])   # mix the routes are compared in), one result kept alive
    torch.cuda.synchronize(device)
    out = hp.forward_scene(scene.features, metas[0])    # finds every span's events complete: decides
    if hp.overlap_choice(scene.features.shape, device)[0] not in (None, "one"):
        # a kept side route stays under watch (MVSDetHotPath._watch): two windows of it here, so that a route that does not hold
        # what its few tuning scenes promised is already given up when the timed loop starts
        for rep in range(2):
            for i in range(2 * (hp._WATCH_SPAN + 1)):
                hp.prefetch_scene(metas[(i + 1) % len(metas)], device)
                out = hp.forward_scene(scene.features, metas[i % len(metas)])
            torch.cuda.synchronize(device)
    del out
    torch.cuda.synchronize(device)
    return hp.overlap_choice(scene.features.shape, device)


def test_shape_chain_rate(device, name, steps=8):
    """The chain of `full_chain_rate` at a view count the shipped TEST pipelines run (80 / 100 views): scenes per second on one
    stream and with the detector on the side stream, the cost network and the neck alone.  View counts vary per scene in a real
    evaluation; here every scene has the workload's nominal count (the largest the pipeline produces)."""
    from mvsdet_amd.costreg import CostRegNet3DGS
    from mvsdet_amd.hotpath import MVSDetHotPath
    from mvsdet_amd.head import NerfDetHeadConvs
    from mvsdet_amd.neck import IndoorImVoxelNeck
    w = WORKLOADS[name]
    torch.manual_seed(0)
    net = CostRegNet3DGS(w["C"]).to(device).eval()
    neck = IndoorImVoxelNeck(w["C"], 128, [1, 1, 1]).to(device).eval()
    head = (NerfDetHeadConvs(17, 3, 128, 7, arkit_head=True) if w.get("arkit_head") else NerfDetHeadConvs(18, 3, 128, 6)).to(device).eval()
    hp = MVSDetHotPath(N_VOXELS, VOXEL_SIZE, list(w["near_far"]), w["D"], topk=3, cost_regularization=net, neck_3d=neck, bbox_head=head)
    scene = SceneInputs(w, seed=0, device=device)
    metas = unseen_metas(w, 11, steps + 3)
    res = {"workload": name, "views": w["N"], "per_view_intrinsics": bool(w["per_view_K"])}
    with torch.no_grad():
        for overlap, key in ((False, "scenes_per_sec"), (True, "scenes_per_sec_pipelined")):
            hp.overlap_detector = False
            collect_garbage()
            if overlap:   # the route is measured per scene shape (MVSDetHotPath.overlap_detector = "auto"), then kept
                res["pipelined_route"], res["pipelined_route_periods_ms"] = settle_overlap_route(hp, scene, metas, device)
            for i in range(CHAIN_WARMUP):   # (the last one announces metas[2], the first timed scene's)
                hp.prefetch_scene(metas[i % 2 + 1], device)
                out = hp.forward_scene(scene.features, metas[i % 2])
            torch.cuda.synchronize(device)
            t0 = time.perf_counter()
            for i in range(2, steps + 2):
                hp.prefetch_scene(metas[i + 1], device)
                out = hp.forward_scene(scene.features, metas[i])
            torch.cuda.synchronize(device)
            res[key] = round(steps / (time.perf_counter() - t0), 3)
        res["pipelined_route_final"] = hp.overlap_choice(scene.features.shape, device)[0]   # ("one" if the watch gave the side route up)
        res["ms_per_scene"] = round(1e3 / res["scenes_per_sec"], 3)
        res["ms_per_scene_pipelined"] = round(1e3 / res["scenes_per_sec_pipelined"], 3)
        var = out.raw("variance")
        net(var)
        torch.cuda.synchronize(device)
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(3):
            net(var)
        e1.record()
        torch.cuda.synchronize(device)
        res["network_ms"] = round(e0.elapsed_time(e1) / 3, 3)
        res["network_ms_per_view"] = round(res["network_ms"] / w["N"], 4)
        res["network_useful_TFLOPs"] = round(CostRegNet3DGS.flops(w["N"], w["D"], w["H"], w["W"]) / 1e12 / res["network_ms"] * 1e3, 1)
        res["non_empty_voxels"] = int((out["valid"] > 0).sum().item())
    del out, var, scene, hp, net, neck, head
    torch.cuda.empty_cache()
    return res


def full_chain_rate(device, steps=10):
    """a1..a10 with the real CostRegNet_3DGS (random weights, eval mode, no autograd) between a4 and a5 at the shape the
    shipped config runs: what a scene costs end to end on the GPU (the network is ~30x the hot path around it)."""
    from mvsdet_amd.costreg import CostRegNet3DGS
    from mvsdet_amd.hotpath import MVSDetHotPath
    from mvsdet_amd.head import NerfDetHeadConvs
    from mvsdet_amd.neck import IndoorImVoxelNeck
    wr = WORKLOADS["scannet_ref_40v_12d_60x80"]
    torch.manual_seed(0)
    net = CostRegNet3DGS(wr["C"]).to(device).eval()
    neck = IndoorImVoxelNeck(wr["C"], 128, [1, 1, 1]).to(device).eval()   # configs/mvsdet_res50_2x_low_res.py: neck_3d
    head = NerfDetHeadConvs(18, 3, 128, 6).to(device).eval()              # mvsdet_res50_2x_low_res_depth.py:40-43: bbox_head
    hp = MVSDetHotPath(N_VOXELS, VOXEL_SIZE, list(wr["near_far"]), wr["D"], topk=3, cost_regularization=net, neck_3d=neck,
                       bbox_head=head)
    scene = SceneInputs(wr, seed=0, device=device)
    metas = unseen_metas(wr, 7, steps + 3)   # new cameras every scene, announced one scene ahead (as in run_gpu)
    with torch.no_grad():
        collect_garbage()
        for i in range(CHAIN_WARMUP):
            hp.prefetch_scene(metas[i % 2 + 1], device)
            out = hp.forward_scene(scene.features, metas[i % 2])
        torch.cuda.synchronize(device)
        marks = [torch.cuda.Event(enable_timing=True) for _ in range(steps + 1)]
        segs0 = torch.cuda.memory_stats(device).get("segment.all.allocated", 0)
        marks[0].record()
        t0 = time.perf_counter()
        for i in range(2, steps + 2):
            hp.prefetch_scene(metas[i + 1], device)
            out = hp.forward_scene(scene.features, metas[i])
            marks[i - 1].record()
        torch.cuda.synchronize(device)
        el = time.perf_counter() - t0
        new_segments = torch.cuda.memory_stats(device).get("segment.all.allocated", 0) - segs0   # hipMalloc calls of the caching allocator
        scene_sequence = [round(marks[k].elapsed_time(marks[k + 1]), 2) for k in range(steps)]
        per_scene = sorted(scene_sequence)
        # the same loop with the neck and the head of a scene on a stream of their own (MVSDetHotPath.overlap_detector): they run
        # beside the next scene's packing, sweep and first convolution; one synchronisation of the device at the end
        route, route_periods = settle_overlap_route(hp, scene, metas, device)
        for i in range(CHAIN_WARMUP):
            out2 = hp.forward_scene(scene.features, metas[i % 2])
        torch.cuda.synchronize(device)
        t1 = time.perf_counter()
        for i in range(2, steps + 2):
            hp.prefetch_scene(metas[i + 1], device)
            out2 = hp.forward_scene(scene.features, metas[i])
        torch.cuda.synchronize(device)
        el_overlap = time.perf_counter() - t1
        route_final = hp.overlap_choice(scene.features.shape, device)[0]   # ("one" if the watch gave the side route up)
        hp.overlap_detector = False
        del out2
    # the network alone on the variance volume of the last scene.  Its stride-1 layers (79 % of the FLOP) run on the bf16
    # matrix cores with three-term split operands: 3 bf16 MFMAs per fp32-equivalent product, so that share is priced
    # against the DENSE bf16 peak (2.5 PFLOP/s) with 3x its useful FLOP; the stride-2 / transposed layers are fp32 MFMA.
    from mvsdet_amd import ops
    with torch.no_grad():
        var = out["variance"]
        net(var)   # untimed: the loops above left the allocator's pools in another state (the side stream's), the first call re-grows them
        torch.cuda.synchronize(device)
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(3):
            net(var)
        e1.record()
        torch.cuda.synchronize(device)
        net_ms = e0.elapsed_time(e1) / 3
        conv, bn = net.conv0.conv, net.conv0.bn
        sc = torch.ones(conv.out_channels, device=device)
        wq, wmx = ops.split_conv_weight(conv.weight), ops.split_conv_weight_mx(conv.weight)
        routes = {"bf16x3": lambda: ops.conv3d_k3_bf16x3(var, wq, sc, sc, True),      # both read the fp32 variance volume itself
                  "fp16mx": lambda: ops.conv3d_k3_fp16mx(var, wmx, sc, sc, True)}
        c0 = {}
        for name_, fn in routes.items():
            fn()
            e0.record()
            for _ in range(3):
                fn()
            e1.record()
            torch.cuda.synchronize(device)
            c0[name_] = e0.elapsed_time(e1) / 3
    tfl = CostRegNet3DGS.flops(wr["N"], wr["D"], wr["H"], wr["W"]) / 1e12
    c0_tfl = 2.0 * 27 * wr["C"] * 64 * wr["N"] * wr["D"] * wr["H"] * wr["W"] / 1e12
    shipped = net.conv0_precision
    c0_ms = c0[shipped]
    # matrix-pipe work per useful FLOP, in units of the dense bf16 rate: bf16x3 = 3 products; fp16mx = per 8 channels 7 fp16 k-steps + 4
    # block-scaled FP6 instructions of K = 128, each priced as ONE 16x16x32 (MI355X_MICROARCH "MFMA" table: e2m3 at 4x the bf16 rate)
    units = 3.0 if shipped == "bf16x3" else 11.0 / 7.0
    roof = {"bound": "mfma", "achieved": round(units * c0_tfl / c0_ms * 1e3, 1), "peak": 2500.0, "unit": "TFLOP/s",
            "frac": round(units * c0_tfl / c0_ms * 1e3 / 2500.0, 4),
            "kernel": ("conv3d_k3_fp16mx_kernel (conv0 256->64: one fp16 product + one block-scaled FP6 product that carries both correction "
                       "terms per fp32-equivalent product; 11/7 matrix-pipe units per product)" if shipped == "fp16mx" else
                       "conv3d_k3_bf16x3_kernel (conv0 256->64: bf16 MFMA, 3 terms per product)"),
            "note": "peak = dense bf16 at 2.4 GHz; `achieved` = useful FLOP x matrix-pipe units per product / time.  The fp16 + MX kernel runs "
                    "11.6 measured units (an e2m3 K=128 instruction takes 1.16 x a 16x16x32: profiles/r06_mx_mix.txt) and is bound by its LDS "
                    "reads and 8-wave tile, not by the matrix pipes (profiles/r06_conv0_mx_whatif.txt)",
            "kernel_ms": round(c0_ms, 3), "useful_TFLOPs": round(c0_tfl / c0_ms * 1e3, 1), "conv0_precision": shipped,
            "conv0_ms_by_route": {k: round(v, 3) for k, v in c0.items()},
            "network_ms": round(net_ms, 3), "network_useful_TFLOPs": round(tfl / net_ms * 1e3, 1),
            "network_vs_fp32_mfma_peak": round(tfl / net_ms * 1e3 / 157.3, 3), "matrix_precision": net.matrix_precision,
            "view_streams": int(net.view_streams)}   # 2: the second half of the views on a stream of its own (CostRegNet3DGS.view_streams)
    with torch.no_grad():
        vol = out["volume"].unsqueeze(0)
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(3):
            neck(vol)
        e1.record()
        torch.cuda.synchronize(device)
    neck_ms = e0.elapsed_time(e1) / 3
    # the detector on a BATCH of scenes (mvsdet.py:695-696 stacks the volumes; MVSDetHotPath.forward_scenes): one 40 x 40 x 16 volume
    # is 200 blocks for 256 CUs at the neck's largest level, four fill the chip
    det = {}
    with torch.no_grad():
        for bsz in (1, 4):
            vb = vol.expand(bsz, -1, -1, -1, -1).contiguous()
            head(neck(vb))
            torch.cuda.synchronize(device)
            e0.record()
            for _ in range(3):
                head(neck(vb))
            e1.record()
            torch.cuda.synchronize(device)
            det[bsz] = e0.elapsed_time(e1) / 3 / bsz
        del vb
    ntfl = IndoorImVoxelNeck.flops(1, N_VOXELS, wr["C"], 128) / 1e12
    # the 3x3x3 layers (all but ~2 % of the work) run on bf16 MFMA with three terms per product: 3 x useful FLOP
    # against the dense bf16 peak
    neck_roof = {"bound": "mfma", "achieved": round(3 * ntfl / neck_ms * 1e3, 1), "peak": 2500.0, "unit": "TFLOP/s",
                 "frac": round(3 * ntfl / neck_ms * 1e3 / 2500.0, 4), "useful_TFLOPs": round(ntfl / neck_ms * 1e3, 1),
                 "kernel": "IndoorImVoxelNeck forward (3x3x3 layers on bf16x3, the small levels split over the input channels; "
                           "1x1x1 / transposed 2x2x2 layers on the bf16x3 GEMM kernel of csrc/neck_gemm.hip)",
                 "kernel_ms": round(neck_ms, 3)}
    return {"workload": "scannet_ref_40v_12d_60x80", "chain": "a1..a10 + CostRegNet_3DGS + IndoorImVoxelNeck + head convolutions, eval",
            "scenes_per_sec": round(steps / el, 3), "scenes_per_sec_pipelined": round(steps / el_overlap, 3),
            "cost_network_roofline": roof, "neck_roofline": neck_roof,
            "detector_batch1": {"ms_per_scene": round(det[1], 3), "what": "IndoorImVoxelNeck + head convolutions on one volume"},
            "detector_batch4": {"ms_per_scene": round(det[4], 3), "what": "the same on four stacked volumes (forward_scenes), per scene"},
            "ms_per_scene": round(el / steps * 1e3, 3),
            # intervals between events recorded on the caller's stream behind every scene of the one-stream loop: a mean far above the
            # median is one stalled scene (an allocation, the host), not a slow chain
            "ms_per_scene_min_median_max": [round(per_scene[0], 3), round(per_scene[len(per_scene) // 2], 3), round(per_scene[-1], 3)],
            "ms_per_scene_sequence": scene_sequence, "device_allocations_in_timed_loop": int(new_segments),
            "scenes_per_sec_at_the_median_scene": round(1e3 / per_scene[len(per_scene) // 2], 3),
            "detector_on_side_stream": {"scenes_per_sec": round(steps / el_overlap, 3), "ms_per_scene": round(el_overlap / steps * 1e3, 3),
                                        "note": "depth distribution, lifting, neck and head of scene i on their own stream beside scene i+1's packing, sweep and conv0 "
                                                "(MVSDetHotPath.overlap_detector = 'auto': the routes one / side1 / side2 are measured on the first scenes of a "
                                                "shape and the fastest kept); the device is synchronised once, after the last scene",
                                        "route": route, "route_periods_ms": route_periods, "route_final": route_final},
            "cost_network_tflop": round(CostRegNet3DGS.flops(wr["N"], wr["D"], wr["H"], wr["W"]) / 1e12, 3),
            "non_empty_voxels": int((out["valid"] > 0).sum().item())}


def footprint_stats(w, hp, scene, device):
    """How the (tile, plane, neighbour) footprints of the first scene split: out of view (skipped by the sweep, exact),
    staged in the LDS box, or gathered from L2 -- the sweep's speed depends on this mix (tools/box_stats.py)."""
    from mvsdet_amd import ops
    geo = hp.prepare_scene(scene.meta, device)
    N, K, D, H, W = w["N"], geo.neighbor_ids.shape[1], w["D"], w["H"], w["W"]
    if K == 0:
        return None
    from mvsdet_amd import _lib
    tw, th, cap = _lib.sweep_tile_shape(K, D, H, W)
    tiles = ((W + tw - 1) // tw) * ((H + th - 1) // th)
    table = ops.plane_sweep_table(geo.proj_rel, geo.depth_values, H, W)
    nent = N * tiles * D * K
    b = table[:nent * 4].view(torch.int32).view(nent, 4).cpu().numpy().astype(np.int64)  # boxes lead the scratch buffer
    nc, nr = b[:, 1] - b[:, 0] + 1, b[:, 3] - b[:, 2] + 1
    empty = (nc <= 0) | (nr <= 0)
    area = np.where(empty, 0, nc * nr)
    live = ~empty
    fl = table[nent * 4: nent * 4 + N * tiles * D].view(torch.int32).cpu().numpy().astype(np.int64)   # flags follow the boxes
    staged_n = sum(int(((fl >> (4 * j + 1)) & 1).sum()) for j in range(K))
    refill_n = sum(int(((fl >> (4 * j + 2)) & 1).sum()) for j in range(K))
    return {"out_of_view": round(float(empty.mean()), 4), "staged_in_lds": round(float((live & (area <= cap)).mean()), 4),
            "gathered": round(float((live & (area > cap)).mean()), 4), "box_texels": int(cap),
            "box_refills_per_staged": round(refill_n / max(1, staged_n), 4)}


def hbm_copy_ceiling(device, gib=2.0, reps=5):
    from mvsdet_amd import ops
    n = int(gib * (1 << 30) // 4)
    a = torch.empty(n, dtype=torch.float32, device=device).normal_()
    b = torch.empty_like(a)
    ops.device_copy(a, b)
    ts = []
    for _ in range(reps):
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        ops.device_copy(a, b)
        e1.record()
        torch.cuda.synchronize(device)
        ts.append(e0.elapsed_time(e1))
    for _ in range(reps):  # the runtime's own device-to-device copy as a second yardstick; report the better one
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        b.copy_(a)
        e1.record()
        torch.cuda.synchronize(device)
        ts.append(e0.elapsed_time(e1))
    return 2 * n * 4 / (min(ts) * 1e-3) / 1e9


def store_pattern_ceiling(w, device, reps=5):
    """What the sweep's OUTPUT LAYOUT allows on this box: the kernel's store stream alone (same block -> address map, same
    tile width and planes per block, non-temporal 16-byte stores, no taps, no arithmetic: mvsdet_store_pattern_probe_f32)
    timed on a volume of the workload's size.  A device-to-device copy (5 TB/s) is no ceiling of a write-only stream; this is."""
    from mvsdet_amd import _lib, ops
    if w["C"] % 32 or w["W"] % 4:
        return None
    dt, eb = (torch.float16, 2) if w.get("half") else (torch.float32, 4)
    tw, th, _ = _lib.sweep_tile_shape(2, w["D"], w["H"], w["W"])
    n = min(w["N"], w.get("chunk") or w["N"])
    dpb = 0
    if w["D"] <= 16 and tw == 16:
        dpb = (w["D"] + (w["D"] + 3) // 4 - 1) // ((w["D"] + 3) // 4)   # the sweep's "about four planes per block" (planesweep.hip)
    var = torch.empty((n, w["C"], w["D"], w["H"], w["W"]), dtype=dt, device=device)
    ops.store_pattern_probe(var, w["W"], tw, dpb)
    ts = []
    for _ in range(reps):
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        ops.store_pattern_probe(var, w["W"], tw, dpb)
        e1.record()
        torch.cuda.synchronize(device)
        ts.append(e0.elapsed_time(e1))
    nbytes = var.numel() * eb
    del var
    torch.cuda.empty_cache()
    return {"GBps": round(nbytes / (float(np.median(ts)) * 1e-3) / 1e9, 1), "ms": round(float(np.median(ts)), 4), "tile": [tw, th],
            "planes_per_block": dpb or w["D"], "bytes": nbytes}


def cpu_baseline(w, budget_s):
    """Stage 1 (the dominant stage, ~94 % of the reference's CPU time: BASELINE.md section 2) on the host cores,
    bounded sample: a 3-view scene (k=2) at the workload's full C, D, H, W."""
    from mvsdet_amd import synthetic
    from mvsdet_amd.hotpath import MVSDetHotPath
    from oracle import oracle as O
    from oracle import torch_restatement as T
    cores = os.cpu_count() or 1
    ns = 3
    hw = (w["H"], w["W"])
    hp = MVSDetHotPath(N_VOXELS, VOXEL_SIZE, list(w["near_far"]), w["D"])
    meta = synthetic.make_img_meta(ns, hw, seed=0, per_view_intrinsics=w["per_view_K"])
    feat = synthetic.make_features(ns, w["C"], hw, seed=0)
    geo = hp.prepare_scene(meta, "cpu")
    res = {}
    # (a) plain-C oracle, OpenMP over all cores
    O.set_num_threads(cores)
    t0 = time.perf_counter()
    reps = 0
    while True:
        O.plane_sweep_variance(feat, geo.neighbor_ids, geo.proj_rel, geo.depth_values, mode=0)
        reps += 1
        if time.perf_counter() - t0 > budget_s / 2 or reps >= 64:
            break
    dt = time.perf_counter() - t0
    res["c_oracle"] = dict(value=ns * reps / dt, seconds=dt, threads=O.num_threads())
    # (b) the reference's own operator sequence (ATen grid_sample etc.), all cores
    torch.set_num_threads(cores)
    t0 = time.perf_counter()
    reps = 0
    while True:
        T.plane_sweep_variance(feat, geo.neighbor_ids, geo.proj_rel, geo.depth_values, view_chunk=1)
        reps += 1
        if time.perf_counter() - t0 > budget_s / 2 or reps >= 64:
            break
    dt = time.perf_counter() - t0
    res["aten_ops"] = dict(value=ns * reps / dt, seconds=dt, threads=torch.get_num_threads())
    best = max(res, key=lambda k: res[k]["value"])
    return dict(value=round(res[best]["value"], 4), unit="cost volumes/s", cores=cores, kind="port",
                sample=f"stage 1 (warp+variance) of a 3-view k=2 scene at full C={w['C']} D={w['D']} "
                       f"{w['H']}x{w['W']}, {best} restatement, {res[best]['seconds']:.1f} s of CPU work",
                variants={k: round(v["value"], 4) for k, v in res.items()})


def spawn_ranks(n, argv):
    """`bench.py --gpus N` outside a torchrun environment: start the N ranks ourselves.  The parent has not touched
    the GPU (no HIP call, no `torch.cuda.is_available()`), starts `torch.distributed.run` as a CHILD process (never an
    exec: MI355X boxes refuse an exec from a process that initialised the GPU, and we keep the habit), one rank per
    GPU over RCCL, relays rank 0's single JSON line and returns non-zero if any rank failed."""
    import socket
    import subprocess
    with socket.socket() as so:   # a free rendezvous port on the loop-back interface
        so.bind(("127.0.0.1", 0))
        port = so.getsockname()[1]
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", str(n), "--master-addr",
           "127.0.0.1", "--master-port", str(port), os.path.abspath(__file__)] + list(argv)
    env = dict(os.environ)
    env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")   # dmabuf IPC: RCCL needs it on this driver
    env["MVSDET_BENCH_SPAWNED"] = "1"
    print("bench.py: starting %d ranks: %s | %s" % (n, " ".join(cmd[1:8]), " ".join(
        f"{k}={env[k]}" for k in _ENV_KEYS if k in env)), file=sys.stderr, flush=True)
    proc = subprocess.run(cmd, stdout=subprocess.PIPE, text=True, env=env)
    lines = []
    for ln in proc.stdout.splitlines():
        (lines.append(ln) if ln.startswith("{") else print(ln, file=sys.stderr))
    if proc.returncode != 0:
        print(f"bench.py: the {n}-rank launch failed with exit code {proc.returncode}", file=sys.stderr)
        return proc.returncode
    if len(lines) != 1:
        print(f"bench.py: expected ONE JSON line from rank 0, got {len(lines)}", file=sys.stderr)
        return 1
    print(lines[0], flush=True)
    return 0


_ENV_KEYS = ("HSA_ENABLE_IPC_MODE_LEGACY", "NCCL_DEBUG", "RCCL_MSCCL_ENABLE", "NCCL_SOCKET_IFNAME", "HIP_VISIBLE_DEVICES",
             "ROCR_VISIBLE_DEVICES", "MASTER_ADDR", "MASTER_PORT", "MVSDET_DIST_BACKEND", "MVSDET_BENCH_SPAWNED")


def launch_env():
    """The RCCL / HSA environment this rank runs under, for the record: a wrong guess (the IPC mode, a rank that never
    joined) must be visible in the JSON line itself the day the multi-GPU runs happen."""
    env = {k: os.environ[k] for k in _ENV_KEYS if k in os.environ}
    env["backend"] = (torch.distributed.get_backend() if torch.distributed.is_available() and torch.distributed.is_initialized()
                      else None)
    return env


def device_binding(rank, local_rank, world, device):
    """Which physical device every rank REALLY computes on: (rank, local_rank, torch device index, PCI bus id, the rank's
    HIP_VISIBLE_DEVICES) of all ranks, gathered through the process group.  `torch.cuda.set_device(local_rank)` is asserted
    here (current device == the one this rank was given), and on a node with at least `world` devices two ranks on one bus id
    -- a mis-bound rank -- stop the run instead of producing a number."""
    cur = torch.cuda.current_device()
    assert cur == device.index, f"rank {rank}: current device {cur} is not the device it was bound to ({device.index})"
    props = torch.cuda.get_device_properties(device)
    bus = getattr(props, "pci_bus_id", None)
    mine = {"rank": rank, "local_rank": local_rank, "device": int(device.index), "pci_bus_id": bus,
            "HIP_VISIBLE_DEVICES": os.environ.get("HIP_VISIBLE_DEVICES")}
    if world <= 1:
        return [mine]
    allb = [None] * world
    torch.distributed.all_gather_object(allb, mine)
    if torch.cuda.device_count() >= world:
        seen = [(b["device"], b["pci_bus_id"], b["HIP_VISIBLE_DEVICES"]) for b in allb]
        assert len(set(seen)) == world, f"two ranks share a device: {allb}"
    return allb


def count_ranks(world, device=None):
    """How many ranks really take part: every rank adds 1 through the process group (1 without one)."""
    if world <= 1:
        return 1
    from mvsdet_amd import parallel
    on_device = torch.distributed.get_backend() == "nccl"   # RCCL reduces device tensors only
    return int(round(parallel.sum_over_ranks(1.0, device if on_device else None)))


def launch_check(args, rank, world):
    """--launch-check: the rendezvous, barrier and max-over-ranks plumbing of the N-rank launch WITHOUT device work, so
    that the spawn path can be tested on a machine without a GPU (tests/test_bench_launch.py, gloo)."""
    from mvsdet_amd import parallel
    if world > 1:
        parallel.init_distributed(os.environ.get("MVSDET_DIST_BACKEND", "gloo"), None)
    if os.environ.get("MVSDET_BENCH_FAIL_RANK") == str(rank):
        raise SystemExit(3)
    parallel.barrier()
    t = parallel.max_over_ranks(float(rank + 1))
    total = parallel.sum_over_ranks(1.0)
    if rank == 0:
        print(json.dumps({"launch_check": True, "n_gpus": world, "gpus_arg": args.gpus, "max_over_ranks": t,
                          "ranks_seen": int(total)}), flush=True)
    if world > 1:
        import torch.distributed as dist
        dist.destroy_process_group()


def side_workload(name, device, steps=3, warmup=1):
    """Compact line of another BASELINE configuration through the same code (parity-test cases, not the headline)."""
    w = WORKLOADS[name]
    a = argparse.Namespace(steps=steps, warmup=warmup, scene_pool=1)
    el, (tab_ms, slab_ms), _, hp, sc = run_gpu(a, w, 0, 1, device)
    del sc
    torch.cuda.empty_cache()
    nbytes = sweep_bytes_per_cv(w) * (w.get("chunk") or w["N"])
    out = {"sweep_kernel_ms": round(slab_ms, 4), "table_kernel_ms": round(tab_ms, 4),
           "sweep_GBps": round(nbytes / (slab_ms * 1e-3) / 1e9, 1),
           "frac": round(nbytes / (slab_ms * 1e-3) / 1e9 / HBM_PEAK_GBPS, 4),
           "stage1_frac": round(nbytes / ((slab_ms + tab_ms) * 1e-3) / 1e9 / HBM_PEAK_GBPS, 4),
           "scenes_per_sec": round(steps / el, 3), "cost_volumes_per_sec": round(w["N"] * steps / el, 2),
           "geometry_cache": hp.geometry_stats, "voxels": w.get("voxels", N_VOXELS), "dtype": "f32 (f16 storage)" if w.get("half") else "f32"}
    if w.get("chunk"):
        out["views_per_launch"] = w["chunk"]
        out["note"] = "one launch = table + slab kernel of one view chunk"
    sp = store_pattern_ceiling(w, device, reps=3)
    if sp:   # the kernel's store stream alone, same tiles / element size / planes per block
        out["store_pattern_ceiling"] = sp
        out["frac_of_store_pattern_ceiling"] = round(sp["ms"] / slab_ms, 4)
    return out


def committed_traffic(name):
    """HBM-side bytes per launch of the sweep kernel from the committed rocprofv3 PMC passes (bench.py cannot collect
    counters itself).  Only quoted when the profile was taken from THIS library version and workload, else None."""
    from mvsdet_amd import _lib
    try:
        with open(os.path.join(ROOT, "profiles", "sweep_traffic.json")) as fh:
            prof = json.load(fh)
    except (OSError, ValueError):
        return None, None
    if prof.get("lib_version") != int(_lib.load().mvsdet_version()):
        return None, None
    ent = prof.get("workloads", {}).get(name)
    return (None, None) if ent is None else (ent["traffic_bytes"], {"source": prof.get("source"), "measured_in_run": False})


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=10)
    ap.add_argument("--warmup", type=int, default=2)
    ap.add_argument("--workload", default="auto", choices=["auto"] + list(WORKLOADS))
    ap.add_argument("--scene-pool", type=int, default=2, help="distinct resident scenes cycled through")
    ap.add_argument("--cpu-seconds", type=float, default=20.0, help="CPU baseline budget (0 disables)")
    ap.add_argument("--no-extras", action="store_true",
                    help="skip stage breakdown / copy ceiling / R-shape line / other workloads / cost-network chain")
    ap.add_argument("--with-cost-network", action="store_true",
                    help="(default on at N=1 unless --no-extras) scenes/s of a1..a10 with the real cost regularisation "
                         "network in between, reference-true shape, eval mode: our kernels only")
    ap.add_argument("--mode", default="infer", choices=["infer", "train", "view-sharded"],
                    help="train = configs[2]: fwd + bwd + optimiser step with a stand-in cost network under DDP; "
                         "view-sharded = ONE scene over all ranks (one all-gather of the feature shards + one all-reduce "
                         "of the voxel buffer per scene: parallel.forward_scene_view_sharded)")
    ap.add_argument("--launch-check", action="store_true", help=argparse.SUPPRESS)
    args = ap.parse_args()

    if args.gpus > 1 and "WORLD_SIZE" not in os.environ:
        # decided before anything touches the GPU: the parent only launches and relays
        raise SystemExit(spawn_ranks(args.gpus, sys.argv[1:]))

    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    if args.gpus != world:
        raise SystemExit(f"bench.py: --gpus {args.gpus} but WORLD_SIZE={world}: refusing to report a {world}-rank number "
                         f"as a {args.gpus}-GPU one")
    if args.launch_check and args.mode == "view-sharded":
        # the collectives of the intra-scene split with their real sizes, device stages replaced by zero tensors (CPU, gloo)
        from mvsdet_amd import parallel
        if world > 1:
            parallel.init_distributed(os.environ.get("MVSDET_DIST_BACKEND", "gloo"), None)
        wn = args.workload if args.workload != "auto" else "tiny_3v_8d_48x64"
        w = WORKLOADS[wn]
        elapsed, checksum, vmax = run_view_sharded(args, w, rank, world, torch.device("cpu"), dry=True)
        if rank == 0:
            print(json.dumps({"launch_check": True, "mode": "view-sharded", "n_gpus": world, "workload": wn, "steps": args.steps,
                              "views_seen": vmax, "checksum": checksum}), flush=True)
        if world > 1:
            import torch.distributed as dist
            dist.destroy_process_group()
        return
    if args.launch_check:
        return launch_check(args, rank, world)
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs a ROCm device: the hot path has no CPU implementation")
    from mvsdet_amd import _lib
    _lib.load()
    # one rank per GPU; the modulo only matters for dry runs of the N>1 path on a box with fewer GPUs than ranks
    device = torch.device("cuda", local_rank % max(1, torch.cuda.device_count()))
    torch.cuda.set_device(device)
    if world > 1:
        from mvsdet_amd import parallel
        # "nccl" is RCCL on ROCm; MVSDET_DIST_BACKEND=gloo lets two ranks share one GPU in a dry run
        parallel.init_distributed(os.environ.get("MVSDET_DIST_BACKEND", "nccl"), device)
    binding = device_binding(rank, local_rank, world, device)

    name = args.workload
    if name == "auto":
        free = torch.cuda.mem_get_info(device)[0]
        name = "scannet_40v_64d_120x160" if free > 70 * (1 << 30) else "scannet_ref_40v_12d_60x80"
    if args.mode == "view-sharded":
        if args.workload == "auto":
            name = "scannet_ref_40v_12d_60x80"
        w = WORKLOADS[name]
        elapsed, checksum, vmax = run_view_sharded(args, w, rank, world, device)
        ranks_seen = count_ranks(world, device)
        if rank == 0:
            print(json.dumps({
                "metric": "scenes/sec of ONE scene sharded over the ranks by reference views (a1..a10"
                          + (" + cost network)" if args.with_cost_network else ", stand-in logits)"),
                "value": round(args.steps / elapsed, 3), "unit": "scenes/s", "n_gpus": world, "steps": args.steps,
                "warmup": args.warmup, "ms_per_step": round(elapsed / args.steps * 1e3, 4), "higher_is_better": True,
                "scaling": "strong", "vs_baseline": None, "dtype": "f32", "data": "synthetic", "mode": "view-sharded",
                "config": {"workload": name, "views": w["N"], "channels": w["C"], "depth_planes": w["D"],
                           "feat_hw": [w["H"], w["W"]], "voxels": w.get("voxels", N_VOXELS),
                           "parallelism": f"view-sharded x{world}: 1 all-gather (feature shards) + 1 all-reduce (voxel buffer) per scene"},
                "roofline": None, "ranks_seen": ranks_seen, "launch_env": launch_env(), "device_binding": binding, "checksum": checksum,
                "views_in_fullest_voxel": vmax}), flush=True)
        if world > 1:
            import torch.distributed as dist
            dist.destroy_process_group()
        return
    if args.mode == "train":
        if args.workload == "auto":
            name = "scannet_ref_40v_12d_60x80"   # what mvsdet_res50_2x_low_res.py trains on
        w = WORKLOADS[name]
        elapsed, checksum = run_train(args, w, rank, world, device)
        ranks_seen = count_ranks(world, device)
        if rank == 0:
            print(json.dumps({
                "metric": "training scenes/sec through the hot path (fwd a1..a10 + bwd + optimiser step, "
                          + ("real cost network)" if args.with_cost_network else "stand-in cost network)"),
                "value": round(args.steps * world / elapsed, 3), "unit": "scenes/s", "n_gpus": world, "steps": args.steps,
                "warmup": args.warmup, "ms_per_step": round(elapsed / args.steps * 1e3, 4), "higher_is_better": True,
                "scaling": "weak", "vs_baseline": None, "dtype": "f32", "data": "synthetic", "mode": "train",
                "config": {"workload": name, "views": w["N"], "channels": w["C"], "depth_planes": w["D"],
                           "feat_hw": [w["H"], w["W"]], "scenes_per_step_per_gpu": 1,
                           "parallelism": f"ddp x{world}, gradient all-reduce only (find_unused_parameters=True)",
                           "optimizer": OPTIMIZER},
                "optimizer_ms": args.optimizer_ms,
                **({"gradient_tolerance": BF16X3_GRADIENT_TOLERANCE} if args.with_cost_network else {}),
                "roofline": backward_rooflines(w, device)["backward_sweep"], "ranks_seen": ranks_seen,
                "launch_env": launch_env(), "device_binding": binding, "checksum": checksum}), flush=True)
        if world > 1:
            import torch.distributed as dist
            dist.destroy_process_group()
        return
    w = WORKLOADS[name]

    elapsed, (table_ms, sweep_ms), checksum, hp, scenes = run_gpu(args, w, rank, world, device)
    ranks_seen = count_ranks(world, device)
    n_cv = w["N"] * args.steps * world
    value = n_cv / elapsed
    # one launch = the reference views of one scene, or of one view chunk for the chunked workload (there the
    # events bracket the table kernel too: the shard entry point enqueues both)
    bytes_launch = sweep_bytes_per_cv(w) * (w.get("chunk") or w["N"])
    achieved = bytes_launch / (sweep_ms * 1e-3) / 1e9
    stage1 = bytes_launch / ((sweep_ms + table_ms) * 1e-3) / 1e9
    line = {
        "metric": "cost volumes/sec (plane-sweep variance, one per reference view) through the full hot path",
        "value": round(value, 3), "unit": "cost volumes/s", "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
        "ms_per_step": round(elapsed / args.steps * 1e3, 4), "higher_is_better": True, "scaling": "weak",
        "vs_baseline": None, "dtype": "f32 (f16 storage)" if w.get("half") else "f32", "data": "synthetic",
        "config": {"workload": name, "views": w["N"], "neighbors": 2, "channels": w["C"], "depth_planes": w["D"],
                   "feat_hw": [w["H"], w["W"]], "voxels": w.get("voxels", N_VOXELS), "scenes_per_step_per_gpu": 1,
                   "parallelism": f"scene-sharded x{world}, no data-path collective"},
        "scenes_per_sec": round(args.steps * world / elapsed, 4),
        # host geometry (a1, a2, a8) of the timed steps: every step's cameras are new -> content_hits must be 0; the
        # algebra ran `misses` times (on the worker thread, announced one step ahead = prefetch_joins)
        "geometry_cache_hits": hp.geometry_stats["content_hits"], "geometry_cache": hp.geometry_stats,
        "roofline": {"bound": "hbm", "achieved": round(achieved, 2), "peak": HBM_PEAK_GBPS, "unit": "GB/s",
                     "frac": round(achieved / HBM_PEAK_GBPS, 4), "traffic": committed_traffic(name)[0],
                     "traffic_source": committed_traffic(name)[1],
                     "kernel": SWEEP_KERNEL_NAME,
                     "kernel_ms": round(sweep_ms, 4), "table_kernel_ms": round(table_ms, 4),
                     # per launch of the timed region (first = the launch after the warm-up's last): the spread between launches
                     "kernel_ms_min_median_max": [round(float(f(hp.sweep_ms_each)), 4) for f in (np.min, np.median, np.max)],
                     "stage1_frac_incl_table": round(stage1 / HBM_PEAK_GBPS, 4),
                     "algorithmic_bytes_per_launch": bytes_launch},
        "ranks_seen": ranks_seen, "launch_env": launch_env(), "device_binding": binding, "checksum": checksum,
    }
    if w.get("chunk"):
        line["config"]["views_per_launch"] = w["chunk"]
    extras = rank == 0 and world == 1 and not args.no_extras
    if extras and not w.get("chunk"):
        line["stage_ms"] = stage_breakdown(w, hp, scenes[0], device)
        line["roofline"]["footprints"] = footprint_stats(w, hp, scenes[0], device)
    del scenes, hp
    torch.cuda.empty_cache()
    if extras:
        line["hbm_copy_ceiling_GBps"] = round(hbm_copy_ceiling(device), 1)   # a read+write stream: reported, not a yardstick
        sp = store_pattern_ceiling(w, device)
        if sp is not None:
            # the sweep writes 0.955 of its algorithmic bytes (D / (D + K + 1) at D = 64): its write stream against what the
            # same store pattern reaches alone on this box
            line["roofline"]["store_pattern_ceiling"] = sp
            wbytes = w["C"] * w["D"] * w["H"] * w["W"] * 4 * (w.get("chunk") or w["N"])
            line["roofline"]["frac_of_store_pattern_ceiling"] = round(wbytes / (sweep_ms * 1e-3) / 1e9 / sp["GBps"], 4)
        if name != "scannet_ref_40v_12d_60x80":
            # the shape the shipped config really runs, reported beside the headline
            wr = WORKLOADS["scannet_ref_40v_12d_60x80"]
            a2 = argparse.Namespace(steps=20, warmup=3, scene_pool=2)
            el, (tm, sm), _, hp_r, sc_r = run_gpu(a2, wr, 0, 1, device)
            br = sweep_bytes_per_cv(wr) * wr["N"]
            line["reference_true_shape"] = {"workload": "scannet_ref_40v_12d_60x80",
                                            "cost_volumes_per_sec": round(wr["N"] * 20 / el, 2),
                                            "scenes_per_sec": round(20 / el, 3), "sweep_kernel_ms": round(sm, 4),
                                            "table_kernel_ms": round(tm, 4),
                                            "sweep_GBps": round(br / (sm * 1e-3) / 1e9, 1),
                                            "frac": round(br / (sm * 1e-3) / 1e9 / HBM_PEAK_GBPS, 4),
                                            "geometry_cache": hp_r.geometry_stats,
                                            "stage_ms": stage_breakdown(wr, hp_r, sc_r[0], device)}
            spr = store_pattern_ceiling(wr, device)
            if spr is not None:
                line["reference_true_shape"]["store_pattern_ceiling"] = spr
                line["reference_true_shape"]["frac_of_store_pattern_ceiling"] = round(
                    wr["N"] * wr["C"] * wr["D"] * wr["H"] * wr["W"] * 4 / (sm * 1e-3) / 1e9 / spr["GBps"], 4)
            del sc_r, hp_r
            torch.cuda.empty_cache()
        if name == "scannet_40v_64d_120x160":
            # the other BASELINE configurations through the same code: configs[3], configs[4] (fp32 at C=32; as worded
            # in fp16 storage, chunked)
            line["other_workloads"] = {n: side_workload(n, device) for n in
                                       ("arkit_50v_96d_60x80", "stress_100v_128d_240x320_c32",
                                        "stress_100v_128d_240x320_c256_f16")}
    if rank == 0 and world == 1 and (args.with_cost_network or extras):
        line["with_cost_network"] = full_chain_rate(device)
        # the view counts tools/test.py runs (80 / 100 views), the same chain
        line["with_cost_network_test_shapes"] = {n: test_shape_chain_rate(device, n) for n in
                                                 ("scannet_test_80v_12d_60x80", "arkit_test_100v_12d_60x80")}
    if extras:
        line["training"] = training_block(device)   # BASELINE.json configs[2] at N = 1
    if rank == 0 and world == 1 and args.cpu_seconds > 0 and not w.get("half"):
        line["cpu_baseline"] = cpu_baseline(w, args.cpu_seconds)
        line["gpu_over_cpu"] = round(value / line["cpu_baseline"]["value"], 1)
    if rank == 0:
        print(json.dumps(line), flush=True)
    if world > 1:
        import torch.distributed as dist
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
