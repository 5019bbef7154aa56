"""Run one of the reference's entry scripts with the HIP hot path patched in, config and source unmodified:

    python -m mvsdet_amd.launch tools/test.py projects/NeRF-Det/configs/mvsdet_res50_2x_low_res.py CKPT
    python -m torch.distributed.run --nproc-per-node 8 --master-addr 127.0.0.1 -m mvsdet_amd.launch tools/train.py CFG --launcher pytorch

The import hook is installed before anything of the reference (or the GPU runtime) is loaded; the script then
runs in this same process as `__main__` (no exec), with its own argv.
"""
import runpy
import sys


def main(argv=None) -> None:
    argv = list(sys.argv[1:] if argv is None else argv)
    if not argv:
        raise SystemExit("usage: python -m mvsdet_amd.launch <script.py> [script arguments ...]")
    from . import _lib, autopatch
    _lib.load()  # fail now, loudly, if libmvsdet_hip.so is not built -- not at the first patched call
    autopatch.install_import_hook()
    sys.argv = argv
    runpy.run_path(argv[0], run_name="__main__")


if __name__ == "__main__":
    main()
