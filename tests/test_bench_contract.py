"""bench.py's contract on the GPU box: one JSON line with the agreed keys at N=1, and the N>1 launch path
(torch.distributed.run, one process per rank, barrier + max-over-ranks timing) as a two-rank dry run that shares the
one GPU of the box over gloo -- on the 8-GPU node the same code runs with backend nccl (= RCCL)."""
import json
import os
import subprocess
import sys

import pytest

from conftest import ROOT

pytestmark = pytest.mark.gpu

KEYS = {"metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "higher_is_better", "scaling", "vs_baseline",
        "dtype", "data", "config", "roofline"}


def _last_json(out):
    lines = [l for l in out.strip().splitlines() if l.startswith("{")]
    assert len(lines) == 1, out
    return json.loads(lines[0])


@pytest.mark.timeout(600)
def test_single_gpu_line(gpu):
    out = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--workload", "tiny_3v_8d_48x64", "--steps", "3",
                          "--warmup", "1", "--cpu-seconds", "2"], capture_output=True, text=True, timeout=500, cwd=ROOT)
    assert out.returncode == 0, out.stderr[-2000:]
    d = _last_json(out.stdout)
    assert KEYS <= set(d) and d["n_gpus"] == 1 and d["steps"] == 3 and d["warmup"] == 1 and d["value"] > 0
    r = d["roofline"]
    assert r["bound"] == "hbm" and r["unit"] == "GB/s" and abs(r["frac"] - r["achieved"] / r["peak"]) < 1e-3
    assert r["table_kernel_ms"] >= 0 and (r["traffic"] is None or (r["traffic"] > 0 and r["traffic_source"]["measured_in_run"] is False))
    c = d["cpu_baseline"]
    assert c["kind"] == "port" and c["cores"] >= 1 and c["value"] > 0 and "sample" in c
    assert d["config"]["workload"] == "tiny_3v_8d_48x64" and "model" not in d["config"]


@pytest.mark.timeout(600)
def test_two_rank_launch_dry_run(gpu):
    # `bench.py --gpus 2` alone: the parent starts the two ranks itself (no torchrun on the command line)
    env = {k: v for k, v in os.environ.items() if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK")}
    env["MVSDET_DIST_BACKEND"] = "gloo"
    cmd = [sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--steps", "3", "--warmup", "1",
           "--workload", "tiny_3v_8d_48x64"]
    out = subprocess.run(cmd, capture_output=True, text=True, timeout=500, cwd=ROOT, env=env)
    assert out.returncode == 0, out.stderr[-2000:]
    d = _last_json(out.stdout)                      # rank 0 prints, once
    assert d["n_gpus"] == 2 and d["scaling"] == "weak" and d["value"] > 0 and "cpu_baseline" not in d
    assert d["config"]["parallelism"].startswith("scene-sharded x2")


@pytest.mark.timeout(600)
def test_training_mode_two_rank_ddp_dry_run(gpu):
    """configs[2]: fwd + bwd + optimiser step of the hot path with a trainable stand-in network under
    DistributedDataParallel, two ranks sharing the GPU over gloo (nccl = RCCL on the 8-GPU node)."""
    env = dict(os.environ, MVSDET_DIST_BACKEND="gloo")
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2", "--master-addr", "127.0.0.1",
           "--master-port", "29537", os.path.join(ROOT, "bench.py"), "--gpus", "2", "--mode", "train", "--steps", "2",
           "--warmup", "1", "--workload", "tiny_3v_8d_48x64"]
    out = subprocess.run(cmd, capture_output=True, text=True, timeout=500, cwd=ROOT, env=env)
    assert out.returncode == 0, out.stderr[-2000:]
    d = _last_json(out.stdout)
    assert d["mode"] == "train" and d["n_gpus"] == 2 and d["value"] > 0 and d["checksum"] > 0
    assert d["config"]["parallelism"].startswith("ddp x2") and "find_unused_parameters=True" in d["config"]["parallelism"]
    # the training config's optimiser (mvsdet_res50_2x_low_res_depth.py:179-184), its share of the step measured
    assert d["config"]["optimizer"] == "AdamW+clip35" and 0 < d["optimizer_ms"] < d["ms_per_step"]


@pytest.mark.timeout(600)
def test_training_mode_with_the_real_cost_network_two_rank_dry_run(gpu):
    """`--mode train --with-cost-network`: the real CostRegNet3DGS under DistributedDataParallel with
    find_unused_parameters=True (configs/mvsdet_res50_2x_low_res_depth.py:200), two ranks over gloo; the record names the
    device every rank was bound to."""
    env = dict(os.environ, MVSDET_DIST_BACKEND="gloo")
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2", "--master-addr", "127.0.0.1",
           "--master-port", "29541", os.path.join(ROOT, "bench.py"), "--gpus", "2", "--mode", "train", "--with-cost-network",
           "--steps", "2", "--warmup", "1", "--workload", "tiny_3v_8d_48x64"]
    out = subprocess.run(cmd, capture_output=True, text=True, timeout=500, cwd=ROOT, env=env)
    assert out.returncode == 0, out.stderr[-2000:]
    d = _last_json(out.stdout)
    assert d["mode"] == "train" and d["n_gpus"] == 2 and d["value"] > 0 and d["checksum"] > 0 and "real cost network" in d["metric"]
    assert d["config"]["optimizer"] == "AdamW+clip35" and d["optimizer_ms"] > 0 and "1e-3" in d["gradient_tolerance"]
    b = d["device_binding"]
    assert [x["rank"] for x in b] == [0, 1] and [x["local_rank"] for x in b] == [0, 1] and all(x["device"] == 0 for x in b)   # one GPU on this box


@pytest.mark.timeout(600)
def test_view_sharded_mode_two_rank_dry_run(gpu):
    """--mode view-sharded on the device stages: two ranks share the GPU over gloo, each sweeping its half of the views;
    one all-gather + one all-reduce per scene."""
    env = {k: v for k, v in os.environ.items() if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK")}
    env["MVSDET_DIST_BACKEND"] = "gloo"
    cmd = [sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--mode", "view-sharded", "--steps", "2", "--warmup", "1",
           "--workload", "tiny_3v_8d_48x64"]
    out = subprocess.run(cmd, capture_output=True, text=True, timeout=500, cwd=ROOT, env=env)
    assert out.returncode == 0, out.stderr[-2000:]
    d = _last_json(out.stdout)
    assert d["mode"] == "view-sharded" and d["n_gpus"] == 2 and d["scaling"] == "strong" and d["value"] > 0
    one = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--mode", "view-sharded", "--steps", "2", "--warmup", "1",
                          "--workload", "tiny_3v_8d_48x64"], capture_output=True, text=True, timeout=500, cwd=ROOT, env=env)
    assert one.returncode == 0, one.stderr[-2000:]
    d1 = _last_json(one.stdout)
    assert abs(d1["checksum"] - d["checksum"]) <= 2e-6 * abs(d1["checksum"]) + 1e-6   # the view sum is re-associated
