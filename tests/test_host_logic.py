"""Host-side mirror of the reference interface (mvsdet_amd.functional / hotpath.prepare_scene) against the
golden vectors.  CPU only: camera algebra on N 4x4 matrices, exactly what the reference does on the host."""
import numpy as np
import pytest
import torch

from conftest import load_golden


def meta_from(g, prefix=""):
    intr = g[prefix + "intrinsic"]
    return {"lidar2img": {"extrinsic": list(g[prefix + "extrinsic"]),
                          "intrinsic": list(intr) if intr.ndim == 3 else intr,
                          "origin": g[prefix + "origin"] if (prefix + "origin") in g.files else np.zeros(3, np.float32)},
            "img_shape": tuple(int(v) for v in g[prefix + "img_shape"]),
            "ori_shape": tuple(int(v) for v in g[prefix + "ori_shape"])}


def test_knn_and_nearest_pose_ids():
    from mvsdet_amd import functional as F_
    g = load_golden("g3_knn")
    for n in (2, 3, 40):
        c2w = torch.tensor(g[f"c2w_{n}"])
        ids = F_.get_nearest_pose_ids(c2w, c2w, 2, maskself=True)
        assert ids.dtype == torch.int64 and ids.shape == (n, min(2, n - 1))
        np.testing.assert_array_equal(ids.numpy(), g[f"ids_{n}"])
    c2w = torch.tensor(g["c2w_dup"])
    np.testing.assert_array_equal(F_.get_nearest_pose_ids(c2w, c2w, 2, maskself=True).numpy(), g["ids_dup"])
    c2w = torch.tensor(g["c2w_40"])
    np.testing.assert_array_equal(F_.get_nearest_pose_ids(c2w, c2w, 3, maskself=False).numpy(), g["ids_40_k3_noself"])
    with pytest.raises(NotImplementedError):
        F_.get_nearest_pose_ids(c2w, c2w, 2, angular_dist_method="matrix")


@pytest.mark.parametrize("tag", ["n3_d8", "n2_k1", "n6_d12_arkit"])
def test_collect_proj_and_relative_projection(tag):
    from mvsdet_amd import functional as F_
    g = load_golden("g2_variance_" + tag)
    nbr = torch.tensor(g["neighbor_ids"])
    proj, nei = F_.collect_proj(torch.tensor(g["w2c"]), torch.tensor(g["K_feat"]), nbr)
    # matmul results: bit-identical on the golden-generating host, last-bit BLAS differences elsewhere
    np.testing.assert_allclose(proj.numpy(), g["ref_proj"], rtol=2e-6, atol=1e-5)
    assert isinstance(nei, tuple) and len(nei) == nbr.shape[1]
    np.testing.assert_allclose(torch.stack(nei, 1).numpy(), g["nei_projs"], rtol=2e-6, atol=1e-5)
    rel = torch.stack([F_.relative_projection(p, proj) for p in nei], 1)
    # same ATen-CPU ops as the reference; LAPACK builds may differ in the last bits between machines
    np.testing.assert_allclose(rel.numpy(), g["proj_rel"], rtol=1e-5, atol=1e-4)


@pytest.mark.parametrize("tag", ["scannet", "arkit"])
def test_projection_and_points(tag):
    from mvsdet_amd import functional as F_
    g = load_golden("g5_backproject_" + tag)
    meta = meta_from(g)
    np.testing.assert_allclose(F_.compute_projection(meta, 4).numpy(), g["projection"], rtol=2e-6, atol=1e-5)
    pts = F_.get_points(torch.tensor(g["n_voxels"]), torch.tensor(g["voxel_size"], dtype=torch.float32), torch.tensor(g["origin"]))
    assert pts.shape == (3, 40, 40, 16)
    np.testing.assert_array_equal(pts.numpy(), g["points"])
    with pytest.raises(NotImplementedError):
        F_.compute_projection(meta, 4, angles=[0.1])


@pytest.mark.parametrize("tag", ["n3_d8", "n2_k1", "n6_d12_arkit"])
def test_prepare_scene(tag):
    from mvsdet_amd.hotpath import MVSDetHotPath
    g = load_golden("g2_variance_" + tag)
    meta = meta_from(g)
    D = g["depth_values"].shape[1]
    hp = MVSDetHotPath([40, 40, 16], [0.16, 0.16, 0.2], list(g["near_far"]), D)
    geo = hp.prepare_scene(meta, "cpu")
    np.testing.assert_array_equal(geo.neighbor_ids.numpy(), g["neighbor_ids"])
    np.testing.assert_array_equal(geo.depth_values.numpy(), g["depth_values"])
    np.testing.assert_allclose(geo.proj_rel.numpy(), g["proj_rel"], rtol=1e-5, atol=1e-4)
    assert geo.height == meta["img_shape"][0] // 4 and geo.width == meta["img_shape"][1] // 4
    assert geo.projection.shape == (g["feature"].shape[0], 3, 4) and geo.points.shape == (3, 40, 40, 16)


def test_depth_planes_match_reference_configs():
    from mvsdet_amd.hotpath import MVSDetHotPath
    # both shipped configs (12 planes) and the BASELINE scale-ups: arange yields exactly D planes
    for nf, D in (((0.2, 5.0), 12), ((0.5, 5.5), 12), ((0.2, 5.0), 64), ((0.5, 5.5), 96), ((0.2, 5.0), 128), ((0.2, 5.0), 8)):
        hp = MVSDetHotPath([40, 40, 16], [0.16, 0.16, 0.2], list(nf), D)
        assert len(hp.depth_values) == D and hp.depth_values.dtype == np.float32
        assert abs(hp.depth_values[0] - nf[0]) < 1e-7


def test_synthetic_scene_is_reproducible_and_overlapping():
    from mvsdet_amd import synthetic
    a = synthetic.make_img_meta(40, (60, 80), seed=3)
    b = synthetic.make_img_meta(40, (60, 80), seed=3)
    np.testing.assert_array_equal(np.array(a["lidar2img"]["extrinsic"]), np.array(b["lidar2img"]["extrinsic"]))
    assert a["img_shape"] == (239, 320) and a["ori_shape"] == (968, 1296)
    f1 = synthetic.make_features(2, 4, (6, 8), seed=1)
    f2 = synthetic.make_features(2, 4, (6, 8), seed=1)
    assert torch.equal(f1, f2)
    pv = synthetic.make_img_meta(5, (60, 80), seed=3, per_view_intrinsics=True)
    assert isinstance(pv["lidar2img"]["intrinsic"], list) and len(pv["lidar2img"]["intrinsic"]) == 5


def test_neighbor_ids_are_validated_on_the_host():
    """mvsdet.py:434-440 feeds the ids to an index gather, which raises on an id outside [0, N): the C ABI's host check
    (mvsdet_validate_neighbors) does the same before anything is uploaded."""
    import torch
    from mvsdet_amd import ops
    ops.validate_neighbors(torch.tensor([[1, 2], [0, 2], [0, 1]]), 3)
    for bad in ([[1, 3]], [[-1, 0]]):
        with pytest.raises(ValueError, match="outside"):
            ops.validate_neighbors(torch.tensor(bad), 3)


def test_scene_geometry_worker_cpu():
    """prepare_scene: camera algebra on the worker thread, results kept by the content of the camera data; prefetch
    returns at once; a moved origin changes the voxel points only; errors surface in the caller."""
    import torch
    from mvsdet_amd import synthetic
    from mvsdet_amd.hotpath import MVSDetHotPath
    hp = MVSDetHotPath([40, 40, 16], [0.16, 0.16, 0.2], [0.2, 5.0], 12)
    meta = synthetic.make_img_meta(6, (60, 80), seed=3)
    other = synthetic.make_img_meta(4, (60, 80), seed=4, per_view_intrinsics=True)
    hp.prefetch_scene(other, "cpu")
    g1 = hp.prepare_scene(meta, "cpu")
    assert hp.prepare_scene(meta, "cpu") is g1
    assert hp.prepare_scene(dict(meta), "cpu") is g1          # same cameras in a new dict: served by content
    moved = dict(meta)
    moved["lidar2img"] = dict(meta["lidar2img"], origin=np.array([0.1, 0.0, 0.5], np.float32))   # RandomShiftOrigin
    g2 = hp.prepare_scene(moved, "cpu")
    assert g2 is not g1 and torch.equal(g2.proj_rel, g1.proj_rel) and torch.equal(g2.neighbor_ids, g1.neighbor_ids)
    assert not torch.equal(g2.points, g1.points)
    g3 = hp.prepare_scene(other, "cpu")
    assert g3.neighbor_ids.shape == (4, 2) and g3.points.shape == (3, 40, 40, 16)
    n0 = torch.get_num_threads()
    assert torch.get_num_threads() == n0          # the caller's thread settings are never touched
    broken = dict(meta)
    broken["lidar2img"] = dict(meta["lidar2img"], extrinsic=[])
    with pytest.raises(Exception):
        hp.prepare_scene(broken, "cpu")
