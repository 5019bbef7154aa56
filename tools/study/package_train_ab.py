"""A/B of two checkouts of the python package on one box: python <this> <root of a checkout>  (e.g. `git archive <commit> mvsdet_amd bench.py | tar -x -C .exp/old`\nwith the current libmvsdet_hip.so copied in) -- how the regressions of the event-ordered buffer pool were found in round 5."""
import sys
root = sys.argv[1]
sys.path.insert(0, root)
import torch
import mvsdet_amd
from mvsdet_amd.costreg import CostRegNet3DGS
from mvsdet_amd.neck import IndoorImVoxelNeck
from mvsdet_amd.head import NerfDetHeadConvs
dev = torch.device("cuda:0")
torch.manual_seed(0)
net = CostRegNet3DGS(256).to(dev).train()
x = torch.randn(40, 256, 12, 60, 80, device=dev, requires_grad=True)
def step():
    net.zero_grad(set_to_none=True); x.grad = None
    net(x).sum().backward()
tag = mvsdet_amd.__file__.split("/")[-3]
for rnd in range(2):
    step(); torch.cuda.synchronize()
    ts = []
    for _ in range(4):
        a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        a.record(); step(); b.record(); torch.cuda.synchronize(); ts.append(a.elapsed_time(b))
    print(tag, f"cost network training step (bf16x3): min {min(ts):.2f} ms median {sorted(ts)[2]:.2f} ms  peak mem {torch.cuda.max_memory_allocated() / 2**30:.2f} GiB", flush=True)
neck = IndoorImVoxelNeck(256, 128, [1, 1, 1]).to(dev).eval()
head = NerfDetHeadConvs(18, 3, 128, 6).to(dev).eval()
v = torch.randn(1, 256, 40, 40, 16, device=dev)
with torch.no_grad():
    for rnd in range(2):
        for _ in range(3):
            head(neck(v))
        torch.cuda.synchronize()
        a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        a.record()
        for _ in range(10):
            head(neck(v))
        b.record(); torch.cuda.synchronize()
        print(tag, f"neck + head: {a.elapsed_time(b) / 10:.3f} ms", flush=True)
