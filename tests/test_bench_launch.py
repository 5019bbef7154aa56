"""`python bench.py --gpus N` must start N ranks itself (VERDICT r1 #6 / ADVICE r1): the parent spawns
`torch.distributed.run` as a child before anything touches the GPU, relays rank 0's single JSON line and fails when a
rank fails.  Exercised here without a GPU through `--launch-check` (rendezvous + barrier + max-over-ranks over gloo;
on the 8-GPU node the children use backend nccl = RCCL)."""
import json
import os
import subprocess
import sys

import pytest

from conftest import ROOT

BENCH = os.path.join(ROOT, "bench.py")


def _run(extra_env, *argv):
    env = {k: v for k, v in os.environ.items() if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK")}
    env.update(MVSDET_DIST_BACKEND="gloo", **extra_env)
    return subprocess.run([sys.executable, BENCH, *argv], capture_output=True, text=True, timeout=300, cwd=ROOT, env=env)


@pytest.mark.timeout(400)
def test_gpus_flag_spawns_that_many_ranks():
    out = _run({}, "--gpus", "2", "--launch-check")
    assert out.returncode == 0, out.stderr[-2000:]
    lines = [l for l in out.stdout.splitlines() if l.startswith("{")]
    assert len(lines) == 1, out.stdout
    d = json.loads(lines[0])
    assert d["n_gpus"] == 2 and d["ranks_seen"] == 2 and d["max_over_ranks"] == 2.0 and d["gpus_arg"] == 2


@pytest.mark.timeout(400)
def test_failing_rank_fails_the_launch():
    out = _run({"MVSDET_BENCH_FAIL_RANK": "1"}, "--gpus", "2", "--launch-check")
    assert out.returncode != 0
    assert not [l for l in out.stdout.splitlines() if l.startswith("{")]


@pytest.mark.timeout(400)
def test_view_sharded_mode_two_ranks():
    """--mode view-sharded: one scene over the ranks (one all-gather of the feature shards + one all-reduce of the voxel
    buffer per scene, mvsdet_amd.parallel.forward_scene_view_sharded) launched as two gloo ranks; device stages replaced by
    shape-only stand-ins under --launch-check, so what runs is the launch, the sharding and the two collectives."""
    out = _run({}, "--gpus", "2", "--mode", "view-sharded", "--launch-check", "--steps", "2", "--warmup", "1",
               "--workload", "tiny_3v_8d_48x64")
    assert out.returncode == 0, out.stderr[-2000:]
    lines = [l for l in out.stdout.splitlines() if l.startswith("{")]
    assert len(lines) == 1, out.stdout
    d = json.loads(lines[0])
    # the stand-in lifting reports `count` views per voxel on every rank: the all-reduce must add them up to all 3 views
    assert d["mode"] == "view-sharded" and d["n_gpus"] == 2 and d["views_seen"] == 3


def test_mismatched_world_size_is_refused():
    out = _run({"WORLD_SIZE": "1", "RANK": "0", "LOCAL_RANK": "0"}, "--gpus", "2", "--launch-check")
    assert out.returncode != 0 and "refusing" in out.stderr


def test_parent_decides_before_touching_the_gpu():
    """The spawn decision must precede every device call (a process that initialised the GPU must not start the ranks
    by exec; we do not exec at all): in main(), `spawn_ranks` is reached before `torch.cuda` or `_lib.load()`."""
    src = open(BENCH).read()
    body = src[src.index("def main():"):]
    assert body.index("spawn_ranks(") < body.index("torch.cuda.") and body.index("spawn_ranks(") < body.index("_lib.load()")
    assert "os.exec" not in src
