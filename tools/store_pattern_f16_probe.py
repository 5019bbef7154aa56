"""What the sweep's store stream reaches with fp16 storage (BASELINE configs[4]: 240x320 maps, 128 planes, C=256, ten views per
launch): the kernel's own pattern (32x4 tiles, 8 bytes per lane: 64-byte runs per channel row), the same with lanes of adjacent
pixel quads storing 16 bytes of two channel rows ("probe_f16_pair"), other tile shapes, and fp32 on 32x4 beside them.
Measured: 3.24 TB/s, paired 3.64, 64x2 paired 3.31, 16x8 0.72; fp32 5.67 -- half-line runs cap the fp16 stream near 0.4-0.45 of HBM."""
import numpy as np
import torch
from mvsdet_amd import _lib, ops

dev = torch.device("cuda:0")
N, C, D, H, W = 10, 256, 128, 240, 320
for dt, tw, pair in ((torch.float16, 32, 0), (torch.float16, 32, 1), (torch.float16, 64, 1), (torch.float16, 16, 1), (torch.float32, 32, 0)):
    _lib.set_option("probe_f16_pair", pair)
    var = torch.empty((N, C, D, H, W), dtype=dt, device=dev)
    ops.store_pattern_probe(var, W, tw, 0)
    ts = []
    for _ in range(5):
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        ops.store_pattern_probe(var, W, tw, 0)
        e1.record()
        torch.cuda.synchronize()
        ts.append(e0.elapsed_time(e1))
    nb = var.numel() * var.element_size()
    print(f"{str(dt):14s} tile {tw:2d}x{128 // tw} paired={pair}: {np.median(ts):7.3f} ms  {nb / np.median(ts) / 1e6:7.1f} GB/s", flush=True)
    del var
    torch.cuda.empty_cache()
_lib.set_option("probe_f16_pair", 0)
