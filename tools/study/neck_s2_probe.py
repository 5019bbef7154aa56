import os, sys, torch
sys.path.insert(0, "/root/repo")
from mvsdet_amd import neck as NK
def timeit(fn, reps=20):
    fn(); torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / reps
dev = torch.device("cuda:0"); torch.manual_seed(0)
net = NK.IndoorImVoxelNeck(256, 128, [1, 1, 1]).to(dev).eval()
x = torch.randn(1, 256, 40, 40, 16, device=dev).relu()
with torch.no_grad():
    for flag in (False, True, False, True):
        NK.S2_BF16X3 = flag
        NK.drop_derived_tensors(net)
        net(x)
        print("S2_BF16X3", flag, f"{timeit(lambda: net(x)):.3f} ms", flush=True)
