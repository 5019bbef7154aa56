"""bench.full_chain_rate with two and four warm-up scenes: per-scene sequence and the caching allocator's hipMalloc count inside the timed loop.  Run on the GPU box from the repository root."""
import sys, json, subprocess
# run bench's chain measurement alone several times in fresh processes with 2 and 4 warm-up scenes; print sequence + allocations
import sys
sys.path.insert(0, ".")
import bench, torch
dev = torch.device("cuda:0")
torch.cuda.set_device(dev)
for warm in (2, 4, 2, 4):
    bench.CHAIN_WARMUP = warm
    r = bench.full_chain_rate(dev, steps=10)
    print("warm-up", warm, {k: r[k] for k in ("ms_per_scene", "ms_per_scene_min_median_max", "ms_per_scene_sequence", "device_allocations_in_timed_loop")}, flush=True)
    torch.cuda.empty_cache()
