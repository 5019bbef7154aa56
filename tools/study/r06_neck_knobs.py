"""The 3-D neck (IndoorImVoxelNeck 256 -> 128, 40x40x16) under the split knobs of the bf16x3 convolutions, one process, alternating."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
from mvsdet_amd import _lib
from mvsdet_amd.neck import IndoorImVoxelNeck
dev = torch.device("cuda:0")
torch.manual_seed(0)
m = IndoorImVoxelNeck(256, 128, [1, 1, 1]).to(dev).eval()
x = torch.randn(1, 256, 40, 40, 16, device=dev)
def timed(reps=12):
    with torch.no_grad():
        m(x); m(x); torch.cuda.synchronize()
        ts = []
        for _ in range(reps):
            a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            a.record(); y = m(x); b.record(); torch.cuda.synchronize(); ts.append(a.elapsed_time(b))
    return min(ts), sorted(ts)[len(ts) // 2], float(sum(float(t.double().sum()) for t in y))
for rnd in range(2):
    for blocks, groups in ((768, 2), (512, 2), (1024, 2), (1536, 2), (768, 1), (768, 4), (384, 2), (256, 2)):
        _lib.set_option("conv_split_blocks", blocks)
        _lib.set_option("conv_split_min_groups", groups)
        lo, med, chk = timed()
        print(f"split_blocks {blocks:5d} min_groups {groups}: min {lo:.3f} median {med:.3f} ms  checksum {chk:.6f}", flush=True)
_lib.set_option("conv_split_blocks", 768); _lib.set_option("conv_split_min_groups", 2)
