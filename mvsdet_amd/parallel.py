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
