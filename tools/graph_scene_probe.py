#!/usr/bin/env python3
"""Probe (GPU box): forward_scene of the reference-true shape captured into one HIP graph (torch.cuda.CUDAGraph) and replayed
with new cameras copied into the captured buffers, against the eager launches."""
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa: E402

import bench  # noqa: E402
from mvsdet_amd.hotpath import MVSDetHotPath, SceneGeometry  # noqa: E402

name = sys.argv[1] if len(sys.argv) > 1 else "scannet_ref_40v_12d_60x80"
w = bench.WORKLOADS[name]
dev = torch.device("cuda:0")
hp = MVSDetHotPath(bench.N_VOXELS, bench.VOXEL_SIZE, list(w["near_far"]), w["D"])
scenes = [bench.SceneInputs(w, i, dev) for i in range(4)]
geos = [hp.prepare_scene(s.meta, dev) for s in scenes]
torch.cuda.synchronize()


def eager(i):
    s = scenes[i % 4]
    return hp.forward_scene(s.features, s.meta, s.cost_logits, geo=geos[i % 4])


def timeit(fn, n=50):
    for i in range(5):
        fn(i)
    torch.cuda.synchronize()
    t = time.perf_counter()
    for i in range(n):
        fn(i)
    torch.cuda.synchronize()
    return (time.perf_counter() - t) / n * 1e3


print(f"{name}: eager forward_scene {timeit(eager):.3f} ms per scene", flush=True)

# static buffers
st = scenes[0]
feat = st.features.clone()
logits = st.cost_logits.clone()
g0 = geos[0]
sgeo = SceneGeometry(g0.neighbor_ids.clone(), g0.proj_rel.clone(), g0.depth_values.clone(), g0.projection.clone(), g0.points.clone(),
                     g0.height, g0.width)
side = torch.cuda.Stream()
side.wait_stream(torch.cuda.current_stream())
with torch.cuda.stream(side):
    for _ in range(3):
        hp.forward_scene(feat, None, logits, geo=sgeo)
torch.cuda.current_stream().wait_stream(side)
torch.cuda.synchronize()
graph = torch.cuda.CUDAGraph()
with torch.cuda.graph(graph):
    out = hp.forward_scene(feat, None, logits, geo=sgeo)
torch.cuda.synchronize()


def replay(i):
    s, g = scenes[i % 4], geos[i % 4]
    feat.copy_(s.features, non_blocking=True)
    logits.copy_(s.cost_logits, non_blocking=True)
    for a, b in ((sgeo.neighbor_ids, g.neighbor_ids), (sgeo.proj_rel, g.proj_rel), (sgeo.depth_values, g.depth_values),
                 (sgeo.projection, g.projection), (sgeo.points, g.points)):
        a.copy_(b, non_blocking=True)
    graph.replay()
    return out


print(f"{name}: graph replay (incl. copies of the inputs into the captured buffers) {timeit(replay):.3f} ms per scene", flush=True)
ref = eager(1)
got = replay(1)
torch.cuda.synchronize()
print("volume equal:", torch.equal(ref["volume"], got["volume"]), " variance equal:", torch.equal(ref["variance"], got["variance"]))
