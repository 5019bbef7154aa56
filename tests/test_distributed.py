"""N>1 path on CPU: two gloo processes exercise the scene sharding, the max-over-ranks timing reduction,
the result gather and a DDP gradient all-reduce of a stand-in cost-regularisation module (what the
training configuration does over RCCL).  No GPU needed."""
import os
import socket
import sys

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
