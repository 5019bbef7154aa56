"""Config reader (mvsdet_amd.config) and the import hook / launcher (mvsdet_amd.autopatch, .launch) on CPU.
The configs and the stand-in "reference" package used here are written by the test itself; the real reference
config is read only by the refcheck test at the bottom (build container only)."""
import importlib
import os
import subprocess
import sys
import textwrap

import pytest

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _write(path, text):
    os.makedirs(os.path.dirname(path), exist_ok=True)
    with open(path, "w") as fh:
        fh.write(textwrap.dedent(text))


@pytest.fixture()
def cfg_tree(tmp_path):
    _write(tmp_path / "base" / "det.py", """
        _base_ = ['../runtime/not_shipped.py']
        prior = dict(ranges=[[-3.2, -3.2, -1.28, 3.2, 3.2, 1.28]])
        model = dict(type='MVSDet', voxel_size=[.16, .16, .2], n_voxels=[40, 40, 16], near_far_range=[0.2, 8.0],
                     neck_3d=dict(type='IndoorImVoxelNeck', in_channels=256), prior_generator=prior)
        """)
    _write(tmp_path / "cfg" / "low_res.py", """
        import os.path as osp
        _base_ = ['../base/det.py']
        model = dict(type='MVSDet', near_far_range=[0.2, 5.0], gs_cfg=dict(num_monocular_samples=12, d_feature=256),
                     neck_3d=dict(_delete_=True, type='Other'), topk=3)
        def helper(): return 1
        """)
    return tmp_path


def test_load_config_merges_base_chain(cfg_tree):
    from mvsdet_amd import config
    cfg = config.load_config(str(cfg_tree / "cfg" / "low_res.py"))
    m = cfg["model"]
    assert m["near_far_range"] == [0.2, 5.0] and m["n_voxels"] == [40, 40, 16]       # child wins, base kept
    assert m["neck_3d"] == {"type": "Other"}                                          # _delete_ drops inherited keys
    assert m["prior_generator"]["ranges"][0][0] == -3.2
    assert "osp" not in cfg and "helper" not in cfg and "_base_" not in cfg
    kw = config.hotpath_kwargs(cfg)
    assert kw == dict(n_voxels=[40, 40, 16], voxel_size=[.16, .16, .2], near_far_range=[0.2, 5.0],
                      num_monocular_samples=12, topk=3)
    hp = config.hotpath_from_config(str(cfg_tree / "cfg" / "low_res.py"))
    assert hp.num_depth == 12 and abs(hp.depth_interval - 0.4) < 1e-12 and hp.depth_values.shape == (12,)


def test_load_config_errors(cfg_tree):
    from mvsdet_amd import config
    with pytest.raises(FileNotFoundError):
        config.load_config(str(cfg_tree / "cfg" / "low_res.py"), missing_base_ok=False)
    with pytest.raises(KeyError):
        config.hotpath_kwargs({"model": {"type": "MVSDet", "n_voxels": [1, 1, 1]}})
    with pytest.raises(ValueError):
        config.hotpath_kwargs({"model": {"type": "NerfDet"}})
    _write(cfg_tree / "loop" / "a.py", "_base_ = './b.py'\n")
    _write(cfg_tree / "loop" / "b.py", "_base_ = './a.py'\n")
    with pytest.raises(ValueError):
        config.load_config(str(cfg_tree / "loop" / "a.py"))


FAKE_REFERENCE = """
    def homo_warping(src_fea, src_proj, ref_proj, depth_values): return 'reference'
    def backproject_Weigh(*a, **k): return 'reference'
    def get_points(n_voxels, voxel_size, origin): return 'reference'
    class MVSDet:
        def sample_depth_prob(self, prob_volume, off_pred, topk=3): return 'reference'
    """


def test_import_hook_patches_the_reference_module_on_import(tmp_path, monkeypatch):
    _write(tmp_path / "fakeproj" / "__init__.py", "")
    _write(tmp_path / "fakeproj" / "nerfdet" / "__init__.py", "")
    _write(tmp_path / "fakeproj" / "nerfdet" / "mvsdet.py", FAKE_REFERENCE)
    _write(tmp_path / "fakeproj" / "nerfdet" / "other.py", "def homo_warping(): return 'untouched'\n")
    monkeypatch.syspath_prepend(str(tmp_path))
    from mvsdet_amd import autopatch, functional as F_, integration
    hook = autopatch.install_import_hook()
    try:
        assert autopatch.install_import_hook() is hook                     # idempotent
        mod = importlib.import_module("fakeproj.nerfdet.mvsdet")
        other = importlib.import_module("fakeproj.nerfdet.other")
        assert mod.homo_warping is F_.homo_warping and mod.backproject_Weigh is F_.backproject_Weigh
        assert mod.MVSDet.sample_depth_prob is integration.PATCHED_METHODS["sample_depth_prob"]
        assert other.homo_warping() == "untouched"
        assert mod.__mvsdet_amd_originals__["homo_warping"](0, 0, 0, 0) == "reference"
        integration.unpatch_reference(mod, mod.__mvsdet_amd_originals__)
        assert mod.homo_warping(0, 0, 0, 0) == "reference"
    finally:
        autopatch.remove_import_hook()
        for name in [n for n in sys.modules if n.startswith("fakeproj")]:
            del sys.modules[name]
    assert not any(isinstance(f, autopatch.ReferenceImportHook) for f in sys.meta_path)


def test_launcher_runs_a_script_with_the_hook_installed(tmp_path):
    _write(tmp_path / "projects" / "__init__.py", "")
    _write(tmp_path / "projects" / "NeRF-Det" / "__init__.py", "")
    _write(tmp_path / "projects" / "NeRF-Det" / "nerfdet" / "__init__.py", "")
    _write(tmp_path / "projects" / "NeRF-Det" / "nerfdet" / "mvsdet.py", FAKE_REFERENCE)
    _write(tmp_path / "tools" / "entry.py", """
        import importlib, sys
        m = importlib.import_module('projects.NeRF-Det.nerfdet.mvsdet')     # what mmengine's custom_imports does
        print('ARGV', sys.argv[1:], 'PATCHED', m.homo_warping.__module__, __name__)
        """)
    env = dict(os.environ, PYTHONPATH=os.pathsep.join([REPO, str(tmp_path)]))
    out = subprocess.run([sys.executable, "-m", "mvsdet_amd.launch", str(tmp_path / "tools" / "entry.py"), "cfg.py", "--x"],
                         cwd=str(tmp_path), env=env, capture_output=True, text=True, timeout=300)
    assert out.returncode == 0, out.stderr
    assert "ARGV ['cfg.py', '--x'] PATCHED mvsdet_amd.functional __main__" in out.stdout
    bad = subprocess.run([sys.executable, "-m", "mvsdet_amd.launch"], env=env, capture_output=True, text=True, timeout=300)
    assert bad.returncode != 0 and "usage" in bad.stderr


@pytest.mark.refcheck
@pytest.mark.parametrize("name,near_far,planes", [("mvsdet_res50_2x_low_res.py", [0.2, 5.0], 12),
                                                  ("mvsdet_arkit.py", [0.5, 5.5], 12)])
def test_reads_the_shipped_reference_configs(name, near_far, planes):
    path = os.path.join("/root/reference/projects/NeRF-Det/configs", name)
    if not os.path.exists(path):
        pytest.skip("reference tree not mounted")
    from mvsdet_amd import config
    kw = config.hotpath_kwargs(config.load_config(path))
    assert kw["n_voxels"] == [40, 40, 16] and kw["voxel_size"] == [.16, .16, .2]
    assert kw["near_far_range"] == near_far and kw["num_monocular_samples"] == planes and kw["topk"] == 3
