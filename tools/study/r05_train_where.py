"""Host time of a stand-in training step by phase (forward_scene, loss, backward, optimiser): how the generation-2 garbage collection inside bench.py's timed region was found.  Run on the GPU box from the repository root."""
import sys, time
sys.path.insert(0, ".")
import torch
import bench
from mvsdet_amd.hotpath import MVSDetHotPath
w = bench.WORKLOADS["scannet_ref_40v_12d_60x80"]
dev = torch.device("cuda:0")
torch.manual_seed(0)
net = bench.PointwiseCostReg(w["C"]).to(dev)
opt = torch.optim.SGD(net.parameters(), lr=1e-4)
hp = MVSDetHotPath(bench.N_VOXELS, bench.VOXEL_SIZE, list(w["near_far"]), w["D"], topk=3, cost_regularization=net)
scenes = [bench.SceneInputs(w, seed=i, device=dev) for i in range(2)]
metas = bench.unseen_metas(w, 0, 20)
def step(i, t):
    s = scenes[i % 2]
    feat = s.features.detach().requires_grad_(True)
    t0 = time.perf_counter()
    hp.prefetch_scene(metas[i + 1], dev)
    out = hp.forward_scene(feat, metas[i])
    t1 = time.perf_counter()
    loss = out["volume"].square().mean() + out["depth_coding"].mean() + out["est_densities"].mean()
    opt.zero_grad(set_to_none=True)
    t2 = time.perf_counter()
    loss.backward()
    t3 = time.perf_counter()
    opt.step()
    t4 = time.perf_counter()
    t.append((t1 - t0, t2 - t1, t3 - t2, t4 - t3))
ts = []
for i in range(3):
    step(i, [])
torch.cuda.synchronize()
T0 = time.perf_counter()
for i in range(3, 13):
    step(i, ts)
torch.cuda.synchronize()
print(f"wall per step {(time.perf_counter() - T0) / 10 * 1e3:.2f} ms; host time per step: forward_scene {sum(t[0] for t in ts) / 10 * 1e3:.2f}, loss {sum(t[1] for t in ts) / 10 * 1e3:.2f}, backward {sum(t[2] for t in ts) / 10 * 1e3:.2f}, opt {sum(t[3] for t in ts) / 10 * 1e3:.2f} ms", flush=True)
print("per-step host ms:", [round(sum(t) * 1e3, 1) for t in ts])
