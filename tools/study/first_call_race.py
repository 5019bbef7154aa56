"""First call of a fresh CostRegNet3DGS with the views on two streams: the derived tensors (BatchNorm affines) are computed by the
side stream's chain; does the main stream's chain read them too early?  Compares against the one-stream result of the same module
(computed afterwards), many fresh modules; with and without the wait on pending derived tensors."""
import sys
sys.path.insert(0, ".")
import torch
from mvsdet_amd import neck as NK
from mvsdet_amd.costreg import CostRegNet3DGS
dev = torch.device("cuda:0")
x = torch.rand(100, 256, 12, 60, 80, device=dev)
real_await = NK._await_made
for mode in ("no wait", "wait"):
    NK._await_made = real_await if mode == "wait" else (lambda *a: None)
    bad = 0
    for it in range(12):
        torch.manual_seed(it)
        net = CostRegNet3DGS(256).to(dev).eval()
        junk = [torch.full((n,), float('nan'), device=dev) for n in (64, 128, 256, 64, 128, 256) * 16]   # the small blocks the affines will recycle
        torch.cuda.synchronize(); del junk
        w = torch.rand(8, 256, 64, 120, 160, device=dev) * 2.0   # a long kernel in front: both chains are enqueued before either starts
        with torch.no_grad():
            net.view_streams = 2
            a = net(x)                      # the first call ever: derived tensors are made inside it
            torch.cuda.synchronize()
            net.view_streams = 1
            b = net(x)
            torch.cuda.synchronize()
        if not torch.equal(a, b):
            bad += 1
            print(f"   {mode}: module {it}: first two-stream call differs from the one-stream result, max |d| {float((a - b).abs().max()):.3e}, NaN {bool(torch.isnan(a).any())}", flush=True)
        del net
    print(f"{mode}: {bad} of 12 fresh modules differ", flush=True)
