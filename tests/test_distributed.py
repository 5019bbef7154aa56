"""N>1 path on CPU: two gloo processes exercise the scene sharding, the max-over-ranks timing reduction,
the result gather and a DDP gradient all-reduce of a stand-in cost-regularisation module (what the
training configuration does over RCCL).  No GPU needed."""
import os
import socket
import sys

import numpy as np
import pytest
import torch
import torch.multiprocessing as mp

from conftest import ROOT


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _worker(rank, world, port, q):
    sys.path.insert(0, ROOT)
    os.environ.update(RANK=str(rank), LOCAL_RANK=str(rank), WORLD_SIZE=str(world), MASTER_ADDR="127.0.0.1",
                      MASTER_PORT=str(port))
    import torch.distributed as dist
    from mvsdet_amd import parallel
    r, lr, w = parallel.init_distributed("gloo")
    assert (r, w) == (rank, world)
    mine = parallel.shard_scenes(7, rank, world)
    # each rank "processes" its scenes: payload = scene id squared; times differ per rank
    local = [(s, s * s) for s in mine]
    elapsed = parallel.max_over_ranks(0.5 + rank)
    total = parallel.sum_over_ranks(len(mine))
    allres = parallel.gather_scene_results(local, world)
    parallel.barrier()
    # DDP: gradients of a tiny stand-in network are averaged over ranks
    torch.manual_seed(0)
    net = torch.nn.Conv3d(4, 2, 3, padding=1)
    ddp = torch.nn.parallel.DistributedDataParallel(net)
    x = torch.full((1, 4, 4, 6, 6), float(rank + 1))
    ddp(x).sum().backward()
    g = net.weight.grad.clone()
    # reference: mean of the two per-rank gradients
    ref = torch.zeros_like(g)
    for rr in range(world):
        n2 = torch.nn.Conv3d(4, 2, 3, padding=1)
        n2.load_state_dict(net.state_dict())
        n2(torch.full((1, 4, 4, 6, 6), float(rr + 1))).sum().backward()
        ref += n2.weight.grad / world
    q.put((rank, mine, elapsed, total, allres, bool(torch.allclose(g, ref, atol=1e-5))))
    dist.destroy_process_group()


@pytest.mark.timeout(180)
def test_two_rank_gloo():
    world = 2
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_worker, args=(r, world, port, q)) for r in range(world)]
    for p in procs:
        p.start()
    res = sorted(q.get(timeout=150) for _ in range(world))
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    (r0, s0, e0, t0, a0, ok0), (r1, s1, e1, t1, a1, ok1) = res
    assert s0 == [0, 2, 4, 6] and s1 == [1, 3, 5]          # disjoint, complete
    assert e0 == e1 == 1.5                                   # max over ranks
    assert t0 == t1 == 7
    assert a0 == a1 == [(s, s * s) for s in range(7)]        # gathered in scene order
    assert ok0 and ok1                                       # DDP averaged the gradients


def test_shard_scenes_properties():
    from mvsdet_amd import parallel
    for world in (1, 2, 4, 8):
        for n in (0, 1, 7, 8, 40):
            parts = [parallel.shard_scenes(n, r, world) for r in range(world)]
            flat = sorted(s for p in parts for s in p)
            assert flat == list(range(n))
            assert max(len(p) for p in parts) - min(len(p) for p in parts) <= 1
    with pytest.raises(ValueError):
        parallel.shard_scenes(4, 2, 2)
    assert parallel.max_over_ranks(3.0) == 3.0  # no process group: identity


# ------------------------------------------------------------------------------------------------------------
# Intra-scene view sharding (parallel.forward_scene_view_sharded): the collective wiring -- all-gather of the
# per-rank feature maps, contiguous view shards, ONE all-reduce of [sum ; count], division -- runs here on two
# gloo ranks.  The device stages are injected: on CPU they are backed by the oracle (test infrastructure), on a
# GPU box the default HipStages run the kernels (tests/test_gpu_parity.py covers those against the same oracle).
# ------------------------------------------------------------------------------------------------------------
class _OracleStages:
    def __init__(self, hp):
        if ROOT not in sys.path:
            sys.path.insert(0, ROOT)
        from oracle import oracle as orc
        self.orc, self.hp = orc, hp

    def pack(self, feature):
        return feature

    def cost_volume_shard(self, packed, geo, first, count, n_src, C, H, W):
        full = self.orc.plane_sweep_variance(packed.numpy(), geo.neighbor_ids.numpy(), geo.proj_rel.numpy(),
                                             geo.depth_values.numpy(), mode=1)
        return torch.from_numpy(full[first:first + count])

    def depth_distribution(self, logits):
        o = self.orc.depth_prob_topk(logits[:, 0].numpy(), logits[:, 1].numpy(), self.hp.near_far_range[0],
                                     self.hp.depth_interval, self.hp.topk)
        t = {k: torch.from_numpy(v) for k, v in o.items()}
        return t["prob"], t["off"], t["est_depth"], t["est_dens"], t["est_idx"], t["avg_depth"]

    def lift_sum_shard(self, packed, geo, est_depth, est_dens, first, count, n_src, C, H, W):
        h, w = geo.height, geo.width
        o = self.orc.backproject_weigh(packed[first:first + count, :, :h, :w].numpy(), geo.points.reshape(3, -1).numpy(),
                                       geo.projection[first:first + count].numpy(), est_depth.numpy(), est_dens.numpy(),
                                       self.hp.voxel_size[-1])
        return torch.from_numpy(o["volume"].sum(0)), torch.from_numpy(o["valid"].sum(0).astype("int32"))


def _scene(n_views=5, C=8, D=8, hw=(24, 32)):
    from mvsdet_amd import synthetic
    from mvsdet_amd.hotpath import MVSDetHotPath
    hp = MVSDetHotPath([16, 16, 8], [.4, .4, .4], [0.2, 5.0], D)
    meta = synthetic.make_img_meta(n_views, feat_hw=hw, seed=3)
    feat = synthetic.make_features(n_views, C, feat_hw=hw, seed=3)
    logits = synthetic.make_cost_logits(n_views, D, feat_hw=hw, seed=3)
    return hp, meta, feat, logits


def _view_worker(rank, world, port, q):
    sys.path.insert(0, ROOT)
    os.environ.update(RANK=str(rank), LOCAL_RANK=str(rank), WORLD_SIZE=str(world), MASTER_ADDR="127.0.0.1",
                      MASTER_PORT=str(port))
    import torch.distributed as dist
    from mvsdet_amd import parallel
    parallel.init_distributed("gloo")
    torch.set_num_threads(2)
    hp, meta, feat, logits = _scene()
    first, count = parallel.view_shard(feat.shape[0], rank, world)
    out = parallel.forward_scene_view_sharded(hp, feat[first:first + count], meta, cost_logits=logits,
                                              features_are_local=True, stages=_OracleStages(hp))
    q.put((rank, out["view_range"], out["volume"].numpy(), out["valid"].numpy(), out["variance"].numpy(),
           out["est_depth"].numpy()))
    dist.destroy_process_group()


@pytest.mark.timeout(300)
def test_view_sharded_scene_two_rank_gloo():
    world = 2
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_view_worker, args=(r, world, port, q)) for r in range(world)]
    for p in procs:
        p.start()
    res = sorted((q.get(timeout=240) for _ in range(world)), key=lambda t: t[0])
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    # single-process result of the same scene with the same (oracle-backed) stages
    from mvsdet_amd import parallel
    hp, meta, feat, logits = _scene()
    one = parallel.forward_scene_view_sharded(hp, feat, meta, cost_logits=logits, stages=_OracleStages(hp))
    assert one["view_range"] == (0, 5) and int((one["valid"] > 0).sum()) > 50       # the scene is not empty
    assert [r[1] for r in res] == [(0, 3), (3, 5)]                                   # contiguous, complete shards
    for rank, (a, b), volume, valid, variance, est_depth in res:
        np.testing.assert_array_equal(valid, one["valid"].numpy())                   # counts: exact
        np.testing.assert_allclose(volume, one["volume"].numpy(), rtol=0, atol=2e-6)  # re-associated view sum
        np.testing.assert_array_equal(variance, one["variance"].numpy()[a:b])        # per-view stages: exact
        np.testing.assert_array_equal(est_depth, one["est_depth"].numpy()[a:b])
    np.testing.assert_array_equal(res[0][2], res[1][2])                              # every rank holds the same volume


def test_view_shard_properties():
    from mvsdet_amd import parallel
    for world in (1, 2, 3, 8):
        for n in (1, 2, 5, 40, 41):
            shards = [parallel.view_shard(n, r, world) for r in range(world)]
            assert shards[0][0] == 0 and sum(c for _, c in shards) == n
            assert all(shards[i][0] + shards[i][1] == shards[i + 1][0] for i in range(world - 1))
            assert max(c for _, c in shards) - min(c for _, c in shards) <= 1
    with pytest.raises(ValueError):
        parallel.view_shard(4, 2, 2)
