"""A/B of two checkouts of the python package on one box: python <this> <root of a checkout>  (e.g. `git archive <commit> mvsdet_amd bench.py | tar -x -C .exp/old`\nwith the current libmvsdet_hip.so copied in) -- how the regressions of the event-ordered buffer pool were found in round 5."""
import os, sys, time
root = sys.argv[1]
sys.path.insert(0, root)
import torch
import bench
from mvsdet_amd.costreg import CostRegNet3DGS
import mvsdet_amd
dev = torch.device("cuda:0")
torch.manual_seed(0)
net = CostRegNet3DGS(256).to(dev).eval()
x = torch.rand(40, 256, 12, 60, 80, device=dev)
with torch.no_grad():
    for rnd in range(3):
        for k in (1, 2):
            net.view_streams = k
            net(x); net(x)
            torch.cuda.synchronize()
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            for _ in range(6):
                net(x)
            e1.record()
            torch.cuda.synchronize()
            print(os.path.basename(os.path.dirname(mvsdet_amd.__file__ + "/")), mvsdet_amd.__file__.split("/")[-3], f"view_streams={k}: {e0.elapsed_time(e1)/6:.3f} ms", flush=True)
