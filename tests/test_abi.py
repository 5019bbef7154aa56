"""The C-ABI library loads on a machine without a GPU and exports every symbol include/mvsdet_hip.h declares
(no compute calls here); argument checks fail cleanly before anything is launched."""
import ctypes
import os
import re

import pytest

from conftest import ROOT


def declared_functions():
    text = open(os.path.join(ROOT, "include", "mvsdet_hip.h")).read()
    text = re.sub(r"/\*.*?\*/", "", text, flags=re.S)
    return sorted(set(re.findall(r"\b(mvsdet_[A-Za-z0-9_]+)\s*\(", text)))


@pytest.fixture(scope="module")
def lib():
    from mvsdet_amd import _lib
    if not os.path.exists(_lib.LIB_PATH):
        _lib.build()
    return _lib.load()


def test_header_declares_entry_points():
    names = declared_functions()
    assert "mvsdet_plane_sweep_variance_f32" in names and "mvsdet_backproject_weigh_mean_packed_f32" in names
    assert len(names) >= 16


def test_library_exports_every_declared_symbol(lib):
    for name in declared_functions():
        assert hasattr(lib, name), f"libmvsdet_hip.so does not export {name}"


def test_binding_table_matches_header():
    from mvsdet_amd import _lib
    assert sorted(_lib.SIGNATURES) == declared_functions()
    # parameter counts of the ctypes table agree with the header prototypes
    text = open(os.path.join(ROOT, "include", "mvsdet_hip.h")).read()
    text = re.sub(r"/\*.*?\*/", "", text, flags=re.S)
    for name, argtypes in _lib.SIGNATURES.items():
        m = re.search(r"\b%s\s*\(([^;]*?)\)\s*;" % name, text, flags=re.S)
        assert m, name
        params = m.group(1).strip()
        n = 0 if params in ("", "void") else params.count(",") + 1
        assert n == len(argtypes), (name, n, len(argtypes))


def test_host_only_entry_points(lib):
    assert lib.mvsdet_version() == 6001
    # packed layout: ceil(C/32) slabs of 32 floats (128 B) per pixel
    assert lib.mvsdet_packed_bytes(40, 256, 60, 80) == 40 * 60 * 80 * 256 * 4
    assert lib.mvsdet_packed_bytes(3, 5, 4, 4) == 3 * 16 * 32 * 4
    assert lib.mvsdet_packed_bytes(2, 33, 4, 4) == 2 * 2 * 16 * 32 * 4
    assert lib.mvsdet_packed_bytes(0, 5, 4, 4) == 0


def test_argument_checks_do_not_launch(lib):
    # NULL pointers / bad shapes are rejected on the host with a message; nothing touches a device
    assert lib.mvsdet_homo_warp_f32(None, None, None, None, 1, 1, 1, 2, 2, None) == 1
    assert b"NULL" in lib.mvsdet_last_error()
    one = ctypes.c_void_p(16)
    assert lib.mvsdet_homo_warp_f32(one, one, one, one, 1, 1, 1, 1, 1, None) == 1  # H, W must be > 1
    assert b"bad shape" in lib.mvsdet_last_error()
    assert lib.mvsdet_plane_sweep_variance_packed_f32(one, one, one, one, one, one, 1 << 30, 2, 9, 4, 3, 8, 8, None) == 1
    assert b"K=9" in lib.mvsdet_last_error()
    assert lib.mvsdet_depth_prob_topk_f32(one, one, one, one, one, one, None, one, 1, 4, 2, 2, 5, 0.2, 0.4, None) == 1
    assert b"topk" in lib.mvsdet_last_error()
    assert lib.mvsdet_plane_sweep_variance_f32(one, one, one, one, one, one, 8, 2, 2, 4, 3, 8, 8, None) == 2
    assert b"workspace" in lib.mvsdet_last_error()
    sixteen = ctypes.c_void_p(4096)
    assert lib.mvsdet_plane_sweep_variance_packed_f32(one, one, one, one, one, sixteen, 64, 2, 2, 4, 3, 8, 8, None) == 2
    assert b"scratch" in lib.mvsdet_last_error()
    # scratch: one 16-B box per (view, tile, plane, neighbour) + one flags word per (view, tile, plane) + proj / depth copies
    # + 7 plane-group boundaries (u16) per (view, tile)
    assert lib.mvsdet_plane_sweep_scratch_bytes(40, 2, 64, 120, 160) == \
        16 + 40 * 150 * 64 * (2 * 16 + 4) + 40 * 2 * 64 + 40 * 64 * 4 + 40 * 150 * 7 * 2   # 16: the geometry header
    assert lib.mvsdet_plane_sweep_scratch_bytes(40, 0, 64, 120, 160) == 0
    assert lib.mvsdet_plane_sweep_workspace_bytes(40, 2, 256, 64, 120, 160) == \
        lib.mvsdet_packed_bytes(40, 256, 120, 160) + lib.mvsdet_plane_sweep_scratch_bytes(40, 2, 64, 120, 160)


def test_ops_have_no_cpu_path():
    import torch
    from mvsdet_amd import ops
    with pytest.raises((RuntimeError, NotImplementedError)):
        ops.homo_warp(torch.zeros(1, 1, 4, 4), torch.eye(4)[None], torch.ones(1, 2))
    with pytest.raises((RuntimeError, NotImplementedError)):
        ops.pack_features(torch.zeros(1, 4, 4, 4))


def test_missing_library_fails_loudly(monkeypatch, tmp_path):
    from mvsdet_amd import _lib
    monkeypatch.setattr(_lib, "_lib", None)
    monkeypatch.setattr(_lib, "LIB_PATH", str(tmp_path / "nope.so"))
    with pytest.raises(RuntimeError, match="no CPU fallback"):
        _lib.load()


def test_product_never_imports_oracle():
    """oracle/ is test infrastructure: nothing under mvsdet_amd/ may import, load or execute it."""
    pkg = os.path.join(ROOT, "mvsdet_amd")
    for dirpath, _, files in os.walk(pkg):
        for f in files:
            if f.endswith((".py", ".hip", ".h", ".cpp")):
                src = open(os.path.join(dirpath, f)).read()
                code = "\n".join(l for l in src.splitlines() if not l.strip().startswith(("#", "//", "*", "/*")))
                assert "import oracle" not in code and "from oracle" not in code and "libplanesweep_oracle" not in code, f


def test_sweep_tile_shape_query():
    """mvsdet_plane_sweep_tile_shape: 32x4 tiles unless the width is a multiple of 16 but not of 32 AND the block has few
    planes; the box capacity is what K resident boxes leave of the block's LDS budget."""
    from mvsdet_amd import _lib
    assert _lib.sweep_tile_shape(2, 64, 120, 160) == (32, 4, 312)
    assert _lib.sweep_tile_shape(2, 12, 60, 80) == (16, 8, 200)     # reference-true shape
    assert _lib.sweep_tile_shape(2, 96, 60, 80) == (32, 4, 312)     # ARKit: 96 planes per block
    assert _lib.sweep_tile_shape(2, 12, 33, 47)[:2] == (32, 4)
    tw, th, cap = _lib.sweep_tile_shape(4, 64, 120, 160)
    assert (tw, th) == (32, 4) and 4 * (cap + 8) * 128 <= 80 * 1024
