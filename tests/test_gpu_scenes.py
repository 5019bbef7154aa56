"""Scene-level GPU tests at the view counts the shipped TEST pipelines run (VERDICT r4 missing #1, tasks 3, 4b, 8).

`configs/mvsdet_res50_2x_low_res.py:105-126` evaluates with n_images = 81 (up to 80 source views; the de-duplication of
multiview_pipeline.py:432-441 makes the count VARY per scene), `mvsdet_arkit.py:114` with 101.  Everything here runs the
real modules -- CostRegNet3DGS, IndoorImVoxelNeck, the head convolutions -- on our kernels and demands, scene by scene,
the bits of that scene run alone.
"""
import os
import random

import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu


def _modules(C, neck_out, gpu, seed=0, n_classes=18, n_reg=6):
    from mvsdet_amd.costreg import CostRegNet3DGS
    from mvsdet_amd.head import NerfDetHeadConvs
    from mvsdet_amd.neck import IndoorImVoxelNeck
    torch.manual_seed(seed)
    net = CostRegNet3DGS(C).to(gpu).eval()
    neck = IndoorImVoxelNeck(C, neck_out, [1, 1, 1]).to(gpu).eval()
    head = NerfDetHeadConvs(n_classes, 3, neck_out, n_reg).to(gpu).eval()
    return net, neck, head


def _keep(out):
    """The small results of a scene, cloned on the CURRENT stream (reading through the holder makes it wait for the side
    stream), so that the multi-gigabyte variance volume can go."""
    kept = {k: out[k].clone() for k in ("volume", "valid", "prob_volume", "est_depth", "depth_coding")}
    kept["neck"] = [t.clone() for t in out["neck"]]
    kept["head"] = [[t.clone() for t in part] for part in out["head"]]
    return kept


def _same(a, b, what):
    for k in ("volume", "valid", "prob_volume", "est_depth", "depth_coding"):
        assert torch.equal(a[k], b[k]), f"{what}: {k} differs"
    for i, (x, y) in enumerate(zip(a["neck"], b["neck"])):
        assert torch.equal(x, y), f"{what}: neck level {i} differs"
    for pa, pb in zip(a["head"], b["head"]):
        for i, (x, y) in enumerate(zip(pa, pb)):
            assert torch.equal(x, y), f"{what}: head level {i} differs"


def _fresh_hotpath(net, neck, head, grid, vox, D):
    from mvsdet_amd.hotpath import MVSDetHotPath
    return MVSDetHotPath(grid, vox, [0.2, 5.0], D, topk=3, cost_regularization=net, neck_3d=neck, bbox_head=head)


def _solo(net, neck, head, grid, vox, D, feat, meta, gpu):
    """The scene alone: a new driver, the network's buffer pool emptied, the device idle before and after."""
    torch.cuda.synchronize(gpu)
    net._scl.clear()
    hp = _fresh_hotpath(net, neck, head, grid, vox, D)
    with torch.no_grad():
        out = hp.forward_scene(feat, meta)
        kept = _keep(out)
    torch.cuda.synchronize(gpu)
    return kept


@pytest.mark.parametrize("per_view_K", [False, True])
def test_test_time_view_counts_in_sequence(gpu, per_view_K):
    """79, 80, 63, 80 views back to back (the shipped ScanNet test pipeline: <= 80 source views, fewer after the de-duplication) at
    the reference-true shape -- 256 channels, 12 planes, 60 x 80 maps, 40 x 40 x 16 grid --, with the real cost network, neck and
    head: the same bits as each scene alone, on one stream (the network on two halves of the views) and with the detector on the
    side stream; no synchronisation between the scenes.  `per_view_K`: the ARKit form (a list of intrinsics)."""
    from mvsdet_amd import synthetic
    C, D, hw, grid, vox = 256, 12, (60, 80), [40, 40, 16], [0.16, 0.16, 0.2]
    net, neck, head = _modules(C, 128, gpu)
    counts = (79, 80, 63, 80)
    big = synthetic.make_features(80, C, hw, seed=5).to(gpu)
    scenes = [(big[:n] if i % 2 == 0 else big[80 - n:], synthetic.make_img_meta(n, hw, seed=40 + i, per_view_intrinsics=per_view_K))
              for i, n in enumerate(counts)]
    solo = [_solo(net, neck, head, grid, vox, D, f, m, gpu) for f, m in scenes]
    assert all(int((s["valid"] > 0).sum()) > 0 for s in solo)
    for overlap in (False, True):
        hp = _fresh_hotpath(net, neck, head, grid, vox, D)
        hp.overlap_detector = overlap
        with torch.no_grad():
            got = []
            for f, m in scenes:                      # no synchronisation in between
                out = hp.forward_scene(f, m)
                got.append(_keep(out))
                del out
        torch.cuda.synchronize(gpu)
        for i, (a, b) in enumerate(zip(solo, got)):
            _same(a, b, f"scene {i} ({counts[i]} views), overlap_detector={overlap}")


def test_hundred_views_arkit_test_pipeline(gpu):
    """mvsdet_arkit.py:114: 101 images = 100 source views with per-view intrinsics and the ARKit head: two such scenes and an
    87-view one in sequence, detector on the side stream, against each scene alone."""
    from mvsdet_amd import synthetic
    from mvsdet_amd.head import NerfDetHeadConvs
    C, D, hw, grid, vox = 256, 12, (60, 80), [40, 40, 16], [0.16, 0.16, 0.2]
    net, neck, _ = _modules(C, 128, gpu)
    torch.manual_seed(1)
    head = NerfDetHeadConvs(17, 3, 128, 7, arkit_head=True).to(gpu).eval()
    big = synthetic.make_features(100, C, hw, seed=6).to(gpu)
    scenes = [(big[:n], synthetic.make_img_meta(n, hw, seed=60 + i, per_view_intrinsics=True)) for i, n in enumerate((100, 87, 100))]
    solo = [_solo(net, neck, head, grid, vox, D, f, m, gpu) for f, m in scenes]
    hp = _fresh_hotpath(net, neck, head, grid, vox, D)
    hp.overlap_detector = True
    with torch.no_grad():
        got = [_keep(hp.forward_scene(f, m)) for f, m in scenes]
    torch.cuda.synchronize(gpu)
    for i, (a, b) in enumerate(zip(solo, got)):
        _same(a, b, f"scene {i}")


def test_soak_random_view_counts_streams_and_cache_flushes(gpu):
    """200 scenes, view counts drawn from [40, 80], detector on the side stream, the network on one or two streams, the CALLER
    alternating between the default stream and two streams of its own, results read on yet another stream through the
    SceneOutputs holder, `torch.cuda.empty_cache()` every 17 scenes: every scene's volume, depth distribution, neck and head
    bit-equal to that scene alone.  (Buffers kept across calls -- the network's SCL / PSCL pool -- are ordered by events, not
    by stream ids: mvsdet_amd/scratch.py.)"""
    from mvsdet_amd import synthetic
    C, D, hw, grid, vox = 64, 8, (24, 32), [16, 16, 8], [0.4, 0.4, 0.4]
    net, neck, head = _modules(C, 64, gpu, seed=2)
    rng = random.Random(7)
    big = synthetic.make_features(80, C, hw, seed=9).to(gpu)
    distinct = []
    for j in range(14):
        n = rng.randint(40, 80)
        lo = rng.randint(0, 80 - n)
        distinct.append((big[lo:lo + n], synthetic.make_img_meta(n, hw, seed=100 + j)))
    solo = [_solo(net, neck, head, grid, vox, D, f, m, gpu) for f, m in distinct]
    hp = _fresh_hotpath(net, neck, head, grid, vox, D)
    hp.overlap_detector = True
    callers = [torch.cuda.default_stream(gpu), torch.cuda.Stream(device=gpu), torch.cuda.Stream(device=gpu)]
    reader = torch.cuda.Stream(device=gpu)
    torch.cuda.synchronize(gpu)
    results = []
    with torch.no_grad():
        for i in range(int(os.environ.get("MVSDET_SOAK_SCENES", "200"))):   # 200 in the suite; a long run: MVSDET_SOAK_SCENES=3000
            j = rng.randrange(len(distinct))
            net.view_streams = rng.choice((1, 2))
            hp.overlap_detector = rng.random() < 0.8
            f, m = distinct[j]
            with torch.cuda.stream(callers[i % 3]):
                out = hp.forward_scene(f, m)
            # with the side stream the holder orders ANY reading stream; without it the outputs belong to the caller's stream
            read_on = (reader if i % 2 else callers[(i + 1) % 3]) if hp.overlap_detector else callers[i % 3]
            with torch.cuda.stream(read_on):
                kept = _keep(out)
            results.append((j, kept))
            del out
            if i % 17 == 16:
                torch.cuda.empty_cache()
    torch.cuda.synchronize(gpu)
    for i, (j, kept) in enumerate(results):
        _same(solo[j], kept, f"soak scene {i} (distinct scene {j}, {distinct[j][0].shape[0]} views)")
    assert len(net._scl) <= net._scl.capacity


def test_forward_scenes_runs_the_detector_once_on_the_batch(gpu):
    """mvsdet.py:681-698 stacks the scenes' volumes and calls neck_3d once.  `forward_scenes`: every scene's own outputs (volume,
    depth distribution) bit-equal to `forward_scene` of that scene; the batched neck / head rows equal to the per-scene
    detector's within the summation order of the split levels (1e-5 of each tensor's scale), with and without the side stream."""
    from mvsdet_amd import synthetic
    C, D, hw, grid, vox = 64, 8, (24, 32), [16, 16, 8], [0.4, 0.4, 0.4]
    net, neck, head = _modules(C, 64, gpu, seed=3)
    scenes = [(synthetic.make_features(n, C, hw, seed=20 + i).to(gpu), synthetic.make_img_meta(n, hw, seed=20 + i))
              for i, n in enumerate((5, 9, 6, 8))]
    solo = [_solo(net, neck, head, grid, vox, D, f, m, gpu) for f, m in scenes]
    for overlap in (False, True):
        hp = _fresh_hotpath(net, neck, head, grid, vox, D)
        hp.overlap_detector = overlap
        with torch.no_grad():
            res = hp.forward_scenes([f for f, _ in scenes], [m for _, m in scenes])
            assert tuple(res["volume"].shape) == (4, C, *grid) and len(res["scenes"]) == 4
            got = [_keep(o) for o in res["scenes"]]
            stacked = [t.clone() for t in res["neck"]]
        torch.cuda.synchronize(gpu)
        for i, (a, b) in enumerate(zip(solo, got)):
            for k in ("volume", "valid", "prob_volume", "est_depth", "depth_coding"):
                assert torch.equal(a[k], b[k]), f"scene {i}: {k}"
            for lvl, (x, y) in enumerate(zip(a["neck"], b["neck"])):
                assert torch.equal(stacked[lvl][i:i + 1], y)
                s = float(x.abs().max())
                assert float((x - y).abs().max()) <= 1e-5 * max(1.0, s), f"scene {i} neck level {lvl}"
            for pa, pb in zip(a["head"], b["head"]):
                for lvl, (x, y) in enumerate(zip(pa, pb)):
                    s = float(x.abs().max())
                    assert float((x - y).abs().max()) <= 1e-5 * max(1.0, s), f"scene {i} head level {lvl}"


def test_measured_overlap_route(gpu):
    """`overlap_detector = "auto"` (the pipelined route must never lose): the first scenes of a shape run every
    route of MVSDetHotPath.OVERLAP_ROUTES for a few scenes, the period between consecutive scenes' cost networks is measured with
    HIP events (no host wait), and the fastest route is kept -- a side route only if it beats the one-stream route by 1 %.  Every
    scene on the way, whatever route it ran on, gives the bits of that scene alone; two shapes are tuned independently."""
    from mvsdet_amd import synthetic
    from mvsdet_amd.hotpath import MVSDetHotPath
    C, D, hw, grid, vox = 64, 8, (24, 32), [16, 16, 8], [0.4, 0.4, 0.4]
    net, neck, head = _modules(C, 64, gpu, seed=4)
    shapes = {n: (synthetic.make_features(n, C, hw, seed=60 + n).to(gpu), synthetic.make_img_meta(n, hw, seed=60 + n)) for n in (8, 12)}
    solo = {n: _solo(net, neck, head, grid, vox, D, f, m, gpu) for n, (f, m) in shapes.items()}
    hp = _fresh_hotpath(net, neck, head, grid, vox, D)
    hp.overlap_detector = "auto"
    need = len(MVSDetHotPath.OVERLAP_ROUTES) * (MVSDetHotPath._TUNE_WARM + MVSDetHotPath._TUNE_SPAN + 1)
    assert hp.overlap_choice(shapes[8][0].shape) == (None, None)
    with torch.no_grad():
        for i in range(need + 6):
            for n, (f, m) in shapes.items():
                got = _keep(hp.forward_scene(f, m))
                if i % 5 == 0 or i >= need:
                    torch.cuda.synchronize(gpu)
                    _same(solo[n], got, f"scene {i} of the {n}-view shape while the route is being measured")
    torch.cuda.synchronize(gpu)
    for n, (f, m) in shapes.items():
        with torch.no_grad():
            hp.forward_scene(f, m)                   # the call that finds every span's events complete decides
        choice, periods = hp.overlap_choice(f.shape)
        assert choice in MVSDetHotPath.OVERLAP_ROUTES and set(periods) == set(MVSDetHotPath.OVERLAP_ROUTES), (choice, periods)
        assert all(v > 0 for v in periods.values())
        assert choice == "one" or periods[choice] < 0.99 * periods["one"]
        print(f"{n} views: route {choice}, periods {periods}")
    # under autograd the tail stays on the caller's stream whatever was measured
    f, m = shapes[8]
    out = hp.forward_scene(f.clone().requires_grad_(True), m)
    assert "ready" not in out


def test_kept_side_route_is_watched_and_given_up(gpu):
    """`overlap_detector = "auto"` after its decision: a kept side route is re-measured every `_WATCH_SPAN` scenes from events (no
    host wait) against the period `one` was tuned at; two windows in a row more than 3 % behind it and the shape goes back to
    `one`.  Forced here by handing the watch a tuned period nobody can meet: the demotion happens, `overlap_choice` reports it, and
    every scene on the way has the bits of that scene alone."""
    from mvsdet_amd import synthetic
    from mvsdet_amd.hotpath import MVSDetHotPath
    C, D, hw, grid, vox = 64, 8, (24, 32), [16, 16, 8], [0.4, 0.4, 0.4]
    net, neck, head = _modules(C, 64, gpu, seed=4)
    f, m = synthetic.make_features(8, C, hw, seed=68).to(gpu), synthetic.make_img_meta(8, hw, seed=68)
    solo = _solo(net, neck, head, grid, vox, D, f, m, gpu)
    hp = _fresh_hotpath(net, neck, head, grid, vox, D)
    hp.overlap_detector = "auto"
    key = (tuple(f.shape), hp.num_depth, str(f.device))
    hp._overlap_tuning[key] = {"cand": 3, "marks": [], "side_marks": [], "spans": {}, "choice": "side1",
                               "periods_ms": {"one": 1e-6, "side1": 5e-7, "side2": 1e-6}}
    with torch.no_grad():
        for i in range(4 * (MVSDetHotPath._WATCH_SPAN + 1) + 2):
            got = _keep(hp.forward_scene(f, m))
            torch.cuda.synchronize(gpu)
            _same(solo, got, f"scene {i} under watch")
            if hp.overlap_choice(f.shape)[0] == "one":
                break
    st = hp._overlap_tuning[key]
    assert hp.overlap_choice(f.shape)[0] == "one" and st["demoted_from"] == "side1" and st["watched_period_ms"] > 1e-6, st
    with torch.no_grad():
        out = hp.forward_scene(f, m)
    assert "ready" not in out            # back on the caller's stream
    # and a route that keeps what it promised stays: the same watch against a period it beats easily
    hp2 = _fresh_hotpath(net, neck, head, grid, vox, D)
    hp2.overlap_detector = "auto"
    hp2._overlap_tuning[key] = {"cand": 3, "marks": [], "side_marks": [], "spans": {}, "choice": "side1",
                                "periods_ms": {"one": 1e6, "side1": 1.0, "side2": 1e6}}
    with torch.no_grad():
        for i in range(3 * (MVSDetHotPath._WATCH_SPAN + 1)):
            got = _keep(hp2.forward_scene(f, m))
    torch.cuda.synchronize(gpu)
    _same(solo, got, "last watched scene")
    assert hp2.overlap_choice(f.shape)[0] == "side1" and hp2._overlap_tuning[key].get("watched_period_ms", 0) > 0


def test_event_pool_orders_foreign_streams(gpu):
    """mvsdet_amd/scratch.EventPool: (1) a key at its buffer limit hands a buffer that another stream is still working on to the
    next stream, which then WAITS for that work (a long queue of adds followed by an overwrite from the other stream must not be
    overtaken); (2) below the limit a busy buffer is left alone and a new one is made -- two streams running the same layers side
    by side must not serialise on one buffer (the halves of CostRegNet3DGS.view_streams: 8.1 against 7.3 ms when they did)."""
    from mvsdet_amd.scratch import EventPool
    s1, s2 = torch.cuda.Stream(device=gpu), torch.cuda.Stream(device=gpu)
    n = 1 << 26
    make = lambda: torch.zeros(n, device=gpu)   # noqa: E731
    pool = EventPool(4, max_per_key=1)
    for rep in range(3):
        with torch.cuda.stream(s1):
            lease = pool.acquire("k", make, lambda b: (b,), gpu)
            buf = lease.buf
            for _ in range(20):
                buf.add_(1.0)              # a long queue of work on s1
            snap1 = buf.sum(dtype=torch.float64)
            pool.release([lease], gpu)
        with torch.cuda.stream(s2):
            lease2 = pool.acquire("k", make, lambda b: (b,), gpu)
            assert lease2.buf is buf       # the same memory, handed to the other stream
            lease2.buf.fill_(-5.0)
            snap2 = lease2.buf.sum(dtype=torch.float64)
            lease2.buf.zero_()
            pool.release([lease2], gpu)
        torch.cuda.synchronize(gpu)
        assert float(snap1) == 20.0 * n and float(snap2) == -5.0 * n
    pool = EventPool(8)                    # default: up to four free buffers per key before a busy one is waited for
    with torch.cuda.stream(s1):
        a = pool.acquire("k", make, lambda b: (b,), gpu)
        for _ in range(20):
            a.buf.add_(1.0)
        pool.release([a], gpu)
    with torch.cuda.stream(s2):
        b = pool.acquire("k", make, lambda b_: (b_,), gpu)
        assert b.buf is not a.buf          # s1 is still adding: s2 gets a buffer of its own instead of waiting
        pool.release([b], gpu)
    torch.cuda.synchronize(gpu)
    with torch.cuda.stream(s2):
        c = pool.acquire("k", make, lambda b_: (b_,), gpu)
        assert c.buf is a.buf or c.buf is b.buf    # everything finished: free buffers are reused, none made
        pool.release([c], gpu)
    assert len(pool) == 2


def test_first_call_on_two_streams_waits_for_the_derived_tensors(gpu):
    """The BatchNorm affines (and every other tensor derived from parameters) are computed by whichever stream first needs them.  On
    the FIRST call of a fresh CostRegNet3DGS with the views on two streams that is the side stream's chain -- and the main stream's
    chain, running layers ahead or behind, read them before the kernels that fill them had run (stale small blocks recycled by the
    allocator: wrong logits by ~1e-2, found in round 5 by the 100-view test in a full-suite run; latent since the halves exist).
    Every derived tensor now carries the event behind its computation (neck._mark_made / _await_made).  Here the allocator's small
    blocks are poisoned first, so that a premature read cannot go unnoticed, and a long kernel is enqueued in front, so that both
    chains are enqueued before either starts."""
    from mvsdet_amd.costreg import CostRegNet3DGS
    x = torch.rand(40, 256, 12, 60, 80, device=gpu)
    for it in range(6):
        torch.manual_seed(it)
        net = CostRegNet3DGS(256).to(gpu).eval()
        junk = [torch.full((n,), float("nan"), device=gpu) for n in (64, 128, 256, 64, 128, 256) * 16]
        torch.cuda.synchronize(gpu)
        del junk
        busy = torch.rand(8, 256, 32, 120, 160, device=gpu) * 2.0     # noqa: F841 -- the kernel both chains queue up behind
        with torch.no_grad():
            net.view_streams = 2
            first = net(x)
            torch.cuda.synchronize(gpu)
            net.view_streams = 1
            again = net(x)
        torch.cuda.synchronize(gpu)
        assert torch.equal(first, again), f"fresh module {it}: the first two-stream call differs by {float((first - again).abs().max()):.2e}"
