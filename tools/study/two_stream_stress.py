"""The cost network on two halves of the views on two streams against one batch on one stream, the same module, many times, bitwise."""
import sys
sys.path.insert(0, ".")
import torch
from mvsdet_amd.costreg import CostRegNet3DGS
dev = torch.device("cuda:0")
torch.manual_seed(0)
net = CostRegNet3DGS(256).to(dev).eval()
for n in (100, 79, 40):
    x = torch.rand(n, 256, 12, 60, 80, device=dev)
    with torch.no_grad():
        net.view_streams = 1
        ref = net(x).clone()
        torch.cuda.synchronize()
        bad = 0
        for it in range(60):
            net.view_streams = 2 if it % 3 else 3
            if it % 7 == 0:
                net._scl.clear()
            a = net(x)
            if it % 2:
                torch.cuda.synchronize()
            if not torch.equal(a, ref):
                bad += 1
                d = (a - ref).abs().amax(dim=(1, 2, 3, 4))
                print(f"   n={n} iteration {it} (k={net.view_streams}): views that differ {torch.nonzero(d > 0).flatten().tolist()[:12]}... max {float(d.max()):.3e}", flush=True)
        print(f"n={n}: {bad} of 60 multi-stream calls differ from the one-stream result", flush=True)
    del x
