"""What bounds the persistent transposed kernel (csrc/convt_persist.h): conv11 / conv9 at 40 views with parts of its drain path sent
beyond the buffer descriptors (the instructions are issued, the memory system drops them), and on fewer blocks.  The what-if forms are BUILDS
(`-DMVS_CONVT_WHATIF=n` on costreg_bf16.hip, bit 0 loads, bit 1 stores, bit 2 one halo tile, bit 3 one weights stage; libraries under build_ab/,
tools/ab_libs.sh): this script times whatever library MVSDET_HIP_LIB names, with the per-tile kernel, the persistent one and 128 blocks of it."""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from mvsdet_amd import _lib, ops  # noqa: E402


def time_it(fn, reps=20):
    fn()
    torch.cuda.synchronize()
    ev = [torch.cuda.Event(enable_timing=True) for _ in range(reps + 1)]
    for i in range(reps):
        ev[i].record()
        fn()
    ev[reps].record()
    torch.cuda.synchronize()
    ts = sorted(ev[i].elapsed_time(ev[i + 1]) for i in range(reps))
    return ts[0], ts[len(ts) // 2]


def main():
    dev = torch.device("cuda")
    for tag, (N, Cin, Cout, D, H, W), outputs in (("conv11", (40, 128, 64, 6, 30, 40), ("f32",)), ("conv9", (40, 256, 128, 3, 15, 20), ("scl",))):
        g = torch.Generator(device="cpu").manual_seed(5)
        x = torch.randn((N, Cin, D, H, W), generator=g).to(dev)
        w = (torch.randn((Cin, Cout, 3, 3, 3), generator=g) * 0.05).to(dev)
        sc = (torch.rand(Cout, generator=g) + 0.5).to(dev)
        sh = torch.randn(Cout, generator=g).to(dev)
        res = torch.randn((N, Cout, 2 * D, 2 * H, 2 * W), generator=g).to(dev)
        xs, wq = ops.scl_pack(x), ops.split_conv_weight(w, 2)
        y = ops.convT3d_k3_s2_bf16x3(xs, wq, sc, sh, res, True, outputs=outputs)
        fn = lambda: ops.convT3d_k3_s2_bf16x3(xs, wq, sc, sh, res, True, outputs=outputs, scl_out=y if "scl" in outputs else None)  # noqa: E731
        for label, persist in (("per-tile kernel", 0), ("persistent", 1), ("persistent on 128 blocks", 128)):
            _lib.set_option("convT_persist", persist)
            lo, med = time_it(fn)
            print(f"{tag}: {label:36s} min {lo:.3f} median {med:.3f} ms", flush=True)
        _lib.set_option("convT_persist", 1)


if __name__ == "__main__":
    main()
