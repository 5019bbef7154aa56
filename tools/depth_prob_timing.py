"""depth_prob_topk at the headline shape (GPU box).  The register-resident form (D <= 64) takes 0.27 ms, the form that re-reads
the logits from L2 0.30."""
import torch
from mvsdet_amd import ops
dev = torch.device("cuda:0")
N, D, H, W = 40, 64, 120, 160
lg = torch.randn(N, 2, D, H, W, device=dev)
lg[:, 0] *= 3
def run(): return ops.depth_prob_topk(lg[:, 0], lg[:, 1], 0.2, 0.075, 3)
for _ in range(3): run()
torch.cuda.synchronize()
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
e0.record()
for _ in range(20): run()
e1.record(); torch.cuda.synchronize()
print(f"depth_prob_topk {N}x{D}x{H}x{W}: {e0.elapsed_time(e1) / 20:.3f} ms")
