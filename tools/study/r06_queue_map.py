"""Which HIP streams wait behind another stream's BACKLOG?  ROCm maps streams onto GPU_MAX_HW_QUEUES (default 4) hardware queues;
packets of one queue are dispatched in order, so a kernel of stream B that shares its queue with stream A starts only after
everything A had enqueued before it -- even though the two streams are independent.  (Two kernels put on an EMPTY shared queue do
overlap: a pairwise test without backlog shows nothing.)  Streams are created and used once in order; then 300 medium kernels
(~20 ms) are enqueued on stream A and ONE small kernel on stream B: the time from A's start to B's completion tells whether B
ran beside A's backlog (early) or behind it (late).  GPU box: python tools/study/r06_queue_map.py [n_streams]"""
import os
import sys

import torch

n = int(sys.argv[1]) if len(sys.argv) > 1 else 9
dev = torch.device("cuda:0")
x = torch.randn(2048, 2048, device=dev)
null = torch.cuda.default_stream(dev)
streams = [torch.cuda.Stream(device=dev) for _ in range(n)]
for s in streams:                     # first use in creation order
    with torch.cuda.stream(s):
        torch.zeros(1, device=dev)
torch.cuda.synchronize()


def behind(a, b):
    """fraction of A's backlog that had run when B's small kernel completed"""
    torch.cuda.synchronize()
    e0, e1, eb = (torch.cuda.Event(enable_timing=True) for _ in range(3))
    with torch.cuda.stream(a):
        e0.record()
        y = x
        for _ in range(300):
            y = y * 1.0001
        e1.record()
    with torch.cuda.stream(b):
        z = torch.ones(64, device=dev) * 2
        eb.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(eb) / e0.elapsed_time(e1)


names = ["null"] + [f"s{i + 1}" for i in range(n)]
allst = [null] + streams
print(f"GPU_MAX_HW_QUEUES={os.environ.get('GPU_MAX_HW_QUEUES', 'default')}: share of stream A's backlog (rows) that ran before stream B's (columns) small kernel completed")
print("      " + " ".join(f"{v:>5s}" for v in names))
for i, a in enumerate(allst):
    row = []
    for j, b in enumerate(allst):
        row.append("    ." if i == j else f"{min(behind(a, b) for _ in range(2)):5.2f}")
    print(f"{names[i]:>5s} " + " ".join(row), flush=True)
