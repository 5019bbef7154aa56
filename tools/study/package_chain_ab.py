"""A/B of two checkouts of the python package on one box: python <this> <root of a checkout>  (e.g. `git archive <commit> mvsdet_amd bench.py | tar -x -C .exp/old`\nwith the current libmvsdet_hip.so copied in) -- how the regressions of the event-ordered buffer pool were found in round 5."""
import sys, json
root = sys.argv[1]
sys.path.insert(0, root)
import torch
import mvsdet_amd, bench
tag = mvsdet_amd.__file__.split("/")[-3]
r = bench.full_chain_rate(torch.device("cuda:0"))
print(tag, {k: r[k] for k in ("scenes_per_sec", "scenes_per_sec_pipelined", "ms_per_scene")}, r["cost_network_roofline"]["network_ms"], r["cost_network_roofline"]["kernel_ms"], r["neck_roofline"]["kernel_ms"], flush=True)
