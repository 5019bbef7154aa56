"""One process per GPU over torch.distributed (backend "nccl" = RCCL on ROCm; "gloo" in CPU tests).

The hot path shards by independent units: scenes are independent, so N ranks each take
every N-th scene and nothing is exchanged on the data path.  The only collectives are (i) the max-over-ranks
of the timed region in bench.py and (ii), for the training configuration, the gradient all-reduce that
torch's DistributedDataParallel issues over RCCL/xGMI for the trainable modules around the ops (the ops
themselves hold no parameters).
"""
from __future__ import annotations

import os
from typing import List, Sequence

import torch
import torch.distributed as dist


def env_rank_world():
    return int(os.environ.get("RANK", "0")), int(os.environ.get("LOCAL_RANK", "0")), int(os.environ.get("WORLD_SIZE", "1"))


def init_distributed(backend: str | None = None, device: torch.device | None = None) -> tuple:
    """Initialise the default process group from the torchrun environment (no-op for WORLD_SIZE=1).
    Rendezvous on 127.0.0.1 unless MASTER_ADDR says otherwise (container hostnames may not resolve)."""
    rank, local_rank, world = env_rank_world()
    if world > 1 and not dist.is_initialized():
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("MASTER_PORT", "29500")
        if backend is None:
            backend = "nccl" if torch.cuda.is_available() else "gloo"
        kw = {}
        if backend == "nccl" and device is not None:
            kw["device_id"] = device
        dist.init_process_group(backend, rank=rank, world_size=world, **kw)
    return rank, local_rank, world


def shard_scenes(n_scenes: int, rank: int, world: int) -> List[int]:
    """Round-robin assignment of independent scenes to ranks: rank r takes scenes r, r+world, ..."""
    if not (0 <= rank < world):
        raise ValueError(f"rank {rank} outside world of {world}")
    return list(range(rank, n_scenes, world))


def barrier():
    if dist.is_available() and dist.is_initialized() and dist.get_world_size() > 1:
        dist.barrier()


def max_over_ranks(value: float, device=None) -> float:
    """Slowest rank's value (bench.py: the job takes as long as its slowest GPU)."""
    if not (dist.is_available() and dist.is_initialized()) or dist.get_world_size() == 1:
        return float(value)
    t = torch.tensor([float(value)], dtype=torch.float64, device=device if device is not None else "cpu")
    dist.all_reduce(t, op=dist.ReduceOp.MAX)
    return float(t.item())


def sum_over_ranks(value: float, device=None) -> float:
    if not (dist.is_available() and dist.is_initialized()) or dist.get_world_size() == 1:
        return float(value)
    t = torch.tensor([float(value)], dtype=torch.float64, device=device if device is not None else "cpu")
    dist.all_reduce(t, op=dist.ReduceOp.SUM)
    return float(t.item())


def gather_scene_results(local: Sequence[tuple], world: int) -> list:
    """All ranks' (scene_id, payload) pairs, ordered by scene id (evaluation-style gather, CPU objects)."""
    if not (dist.is_available() and dist.is_initialized()) or world == 1:
        return sorted(local, key=lambda kv: kv[0])
    out = [None] * world
    dist.all_gather_object(out, list(local))
    return sorted((kv for part in out for kv in part), key=lambda kv: kv[0])


# ---------------------------------------------------------------------------------------------------------
# Intra-scene split ("optional intra-scene split"): one scene over several GPUs.
#   * the N reference views are dealt to the ranks as contiguous shards (stages 1-2 are per reference view);
#   * every rank needs ALL N feature maps, because neighbours are arbitrary views: one all-gather of the
#     per-rank (M,C,H,W) maps (197 MB at the reference-true shape) when each rank ran the 2-D backbone on its
#     own images only;
#   * stage 3's per-voxel mean is the one real exchange step: each rank sums its views' contributions and
#     counts, one all-reduce(SUM) of a (C+1, X*Y*Z) fp32 buffer (26 MB) follows, then the a10 division.
# The sum over views is re-associated (rank partial sums), so `volume` agrees with the single-GPU result to
# fp32 rounding, not bit for bit; `variance`, `prob_volume`, `est_*` of the local views are bit-identical.
# Inference only (the shard ops carry no autograd).
# ---------------------------------------------------------------------------------------------------------
def view_shard(n_views: int, rank: int, world: int) -> tuple:
    """(first, count) of the contiguous block of reference views of `rank`; earlier ranks take the remainder."""
    if not (0 <= rank < world):
        raise ValueError(f"rank {rank} outside world of {world}")
    base, extra = divmod(n_views, world)
    first = rank * base + min(rank, extra)
    return first, base + (1 if rank < extra else 0)


class HipStages:
    """The three device stages of the sharded scene on the HIP operators (what runs on a GPU box)."""

    def __init__(self, hp):
        self.hp = hp

    def pack(self, feature):
        from . import ops
        return ops.pack_features(feature)

    def cost_volume_shard(self, packed, geo, first, count, n_src, C, H, W):
        from . import ops
        sl = slice(first, first + count)
        return ops.plane_sweep_variance_shard(packed, geo.neighbor_ids[sl], geo.proj_rel[sl], geo.depth_values[sl],
                                              n_src, first, C, H, W)

    def depth_distribution(self, cost_logits):
        return self.hp.depth_distribution(cost_logits)

    def lift_sum_shard(self, packed, geo, est_depth, est_dens, first, count, n_src, C, H, W):
        from . import ops
        h, w = geo.height, geo.width
        return ops.backproject_weigh_sum_shard(packed, geo.points, geo.projection[first:first + count],
                                               est_depth[:, :, :h, :w], est_dens[:, :, :h, :w], n_src, first, C, H, W,
                                               float(self.hp.voxel_size[-1]))


def all_gather_view_features(local: torch.Tensor, n_views: int, group=None) -> torch.Tensor:
    """Per-rank (M_r,C,H,W) maps of contiguous view shards -> the full (N,C,H,W) tensor on every rank.
    Shards may differ by one view: every rank pads to the largest shard so ONE all_gather suffices."""
    world = dist.get_world_size(group)
    rank = dist.get_rank(group)
    first, count = view_shard(n_views, rank, world)
    if local.shape[0] != count:
        raise ValueError(f"rank {rank} holds {local.shape[0]} views, its shard of {n_views} views is {count}")
    biggest = view_shard(n_views, 0, world)[1]
    send = local.contiguous()
    if count < biggest:
        send = torch.cat([send, send.new_zeros((biggest - count,) + tuple(send.shape[1:]))], 0)
    recv = send.new_empty((world * biggest,) + tuple(send.shape[1:]))
    dist.all_gather_into_tensor(recv, send, group=group)
    parts = [recv[r * biggest: r * biggest + view_shard(n_views, r, world)[1]] for r in range(world)]
    return torch.cat(parts, 0)


def forward_scene_view_sharded(hp, feature: torch.Tensor, img_meta: dict, cost_logits=None, group=None,
                               features_are_local: bool = False, stages=None) -> dict:
    """One scene through a1..a10 on all ranks of `group` (see the block comment above).

    feature: (N,C,H,W) on every rank, or this rank's (M,C,H,W) shard with `features_are_local=True`.
    cost_logits: (N,2,D,H,W) stand-in for the cost regularisation network (only this rank's rows are read); with
    `hp.cost_regularization` set, that module runs on the local variance rows instead.
    Returns forward_scene's dictionary with `volume` / `valid` complete on every rank and `variance`,
    `prob_volume`, `off_pred`, `est_depth`, `est_densities`, `depth_coding` for the local views `view_range`."""
    stages = stages or HipStages(hp)
    sharded = dist.is_available() and dist.is_initialized() and dist.get_world_size(group) > 1
    rank = dist.get_rank(group) if sharded else 0
    world = dist.get_world_size(group) if sharded else 1
    n_src = len(img_meta["lidar2img"]["extrinsic"])
    if features_are_local and sharded:
        feature = all_gather_view_features(feature, n_src, group)
    if feature.shape[0] != n_src:
        raise ValueError(f"feature holds {feature.shape[0]} views, img_meta describes {n_src}")
    _, C, H, W = feature.shape
    first, count = view_shard(n_src, rank, world)
    geo = hp.prepare_scene(img_meta, feature.device)       # N 4x4 matrices: every rank recomputes them
    packed = stages.pack(feature)
    nx, ny, nz = hp.n_voxels
    V = nx * ny * nz
    buf = feature.new_zeros((C + 1, V))                     # [sum over local views ; count] -> one all-reduce
    out = dict(geometry=geo, view_range=(first, first + count))
    if count > 0:
        variance = stages.cost_volume_shard(packed, geo, first, count, n_src, C, H, W)
        if hp.cost_regularization is not None:
            logits = hp.cost_regularization(variance)
        elif cost_logits is None:
            raise ValueError("forward_scene_view_sharded needs `cost_logits` when no cost_regularization module is set")
        else:
            logits = cost_logits[first:first + count] if cost_logits.shape[0] == n_src else cost_logits
        prob, off, est_depth, est_dens, _, avg_depth = stages.depth_distribution(logits)
        total, cnt = stages.lift_sum_shard(packed, geo, est_depth, est_dens, first, count, n_src, C, H, W)
        buf[:C] = total
        buf[C] = cnt.to(buf.dtype)                          # exact: counts are far below 2^24
        h, w = geo.height, geo.width
        out.update(variance=variance, prob_volume=prob, off_pred=off, est_depth=est_depth[:, :, :h, :w],
                   est_densities=est_dens[:, :, :h, :w], depth_coding=avg_depth[:, :h, :w].unsqueeze(1),
                   opacity=est_dens[:, 0])
    if sharded:
        dist.all_reduce(buf, op=dist.ReduceOp.SUM, group=group)
    cnt = buf[C]
    volume = torch.where(cnt > 0, buf[:C] / (cnt + 1e-8), torch.zeros((), dtype=buf.dtype, device=buf.device))  # mvsdet.py:511-515
    out.update(volume=volume.view(C, nx, ny, nz), valid=cnt.round().long().view(1, nx, ny, nz))
    return out
