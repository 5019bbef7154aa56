"""Signature-level drop-in check of mvsdet_amd.integration against the reference module itself.
Needs /root/reference (build container); skipped elsewhere.  CPU only: nothing is launched."""
import inspect
import os
import sys

import pytest

from conftest import GOLDEN

pytestmark = pytest.mark.refcheck


@pytest.fixture(scope="module")
def reference():
    sys.path.insert(0, GOLDEN)
    try:
        from _ref_loader import load_reference
        return load_reference()
    except FileNotFoundError:
        pytest.skip("reference tree not mounted")


def _params(fn):
    return [(p.name, p.default if p.default is not inspect._empty else "<req>") for p in inspect.signature(fn).parameters.values()]


def test_mirrors_have_the_reference_signatures(reference):
    ref, ref_module = reference
    from mvsdet_amd import functional as F_, integration
    assert _params(F_.homo_warping) == _params(ref_module.homo_warping)
    assert _params(F_.backproject_Weigh) == _params(ref.backproject_Weigh)
    assert _params(F_.get_nearest_pose_ids) == _params(ref.get_nearest_pose_ids)
    assert _params(F_.knn) == _params(ref.knn)
    assert _params(F_.get_points) == _params(ref.get_points)
    for name, fn in integration.PATCHED_METHODS.items():
        assert _params(fn) == _params(getattr(ref.MVSDet, name)), name


def test_patch_and_unpatch(reference):
    ref, _ = reference
    from mvsdet_amd import functional as F_, integration
    orig = integration.patch_reference(ref)
    try:
        assert ref.homo_warping is F_.homo_warping and ref.backproject_Weigh is F_.backproject_Weigh
        assert ref.MVSDet.sample_depth_prob is integration.PATCHED_METHODS["sample_depth_prob"]
        from mvsdet_amd.costreg import CostRegNet3DGS
        assert ref.CostRegNet_3DGS is CostRegNet3DGS          # MVSDet.__init__ builds the HIP-routed network
        assert set(orig) >= {"homo_warping", "backproject_Weigh", "MVSDet.sample_depth_prob", "MVSDet.compute_avg_depth"}
        assert not integration.apply_on_import()  # already patched: nothing left to do
    finally:
        integration.unpatch_reference(ref, orig)
    assert ref.homo_warping is orig["homo_warping"]


def test_cost_regularisation_network_matches_the_reference_module(reference):
    """mvsdet_amd.costreg.CostRegNet3DGS against mvs_models/mvsnet.py:CostRegNet_3DGS: same parameter names and
    shapes (a reference checkpoint loads), same outputs in eval and in train mode (CPU, fp32)."""
    import torch
    from mvsdet_amd.costreg import CostRegNet3DGS
    ref_net = sys.modules["refpkg.mvs_models.mvsnet"].CostRegNet_3DGS()
    torch.manual_seed(0)
    for m in ref_net.modules():            # non-trivial BatchNorm statistics and affine parameters
        if isinstance(m, torch.nn.BatchNorm3d):
            m.running_mean.normal_(0, 0.1); m.running_var.uniform_(0.5, 1.5)
            m.weight.data.uniform_(0.5, 1.5); m.bias.data.normal_(0, 0.1)
    ours = CostRegNet3DGS()
    sd = ref_net.state_dict()
    assert {k: tuple(v.shape) for k, v in sd.items()} == {k: tuple(v.shape) for k, v in ours.state_dict().items()}
    ours.load_state_dict(sd)
    x = torch.rand(1, 256, 8, 12, 16)
    for mode in ("eval", "train"):
        getattr(ref_net, mode)(); getattr(ours, mode)()
        with torch.no_grad():
            a, b = ref_net(x.clone()), ours(x.clone())
        assert a.shape == (1, 2, 8, 12, 16)
        torch.testing.assert_close(b, a, rtol=1e-5, atol=1e-5)
    # 2.8 TFLOP per scene at the reference-true shape
    assert 2.5e12 < CostRegNet3DGS.flops(40, 12, 60, 80) < 3.2e12
    with pytest.raises(ValueError):
        ours(torch.rand(1, 256, 6, 12, 16))


def test_lazy_route_is_pinned_to_the_reference_source_text(reference, oracle, monkeypatch):
    """The statements of MVSDet.extract_feat that the deferred-evaluation route must recognise, taken from the REFERENCE
    FILE ITSELF (mvsdet.py:416-467: neighbour selection .. variance; :511-515: the sum over the views) and executed
    against the patched module with the fused launches replaced by the CPU oracle: the variance loop (both its training
    and its eval form) collapses into ONE fused plane-sweep call and the view sum into one fused lifting call, nothing is
    materialised, and the results equal those of the same statements on the unpatched reference."""
    import textwrap
    from types import SimpleNamespace
    import numpy as np
    import torch
    from mvsdet_amd import functional as F_, integration, lazywarp, ops, synthetic
    ref, _ = reference
    src = open(ref.__file__).read().splitlines()
    block = textwrap.dedent("\n".join(src[415:467]))          # lines 416..467
    assert block.lstrip().startswith("height = img_meta['img_shape'][0] // stride") and "volume_variance = volume_sq_sum.div_(k+1)" in block
    tail = textwrap.dedent("\n".join(src[510:515]))           # lines 511..515
    assert tail.lstrip().startswith("volume_sum = volume.sum(dim=0)") and "volume_mean[:, valid[0] == 0] = .0" in tail

    N, C, D, hw = 4, 6, 8, (60, 80)
    meta = synthetic.make_img_meta(N, hw, seed=21)
    feature = synthetic.make_features(N, C, hw, seed=21)
    depth_values = np.arange(0.2, 5.0, 4.8 / D, dtype=np.float32)

    def run(training):
        me = SimpleNamespace(gs_cfg=SimpleNamespace(num_monocular_samples=D), depth_values=depth_values, training=training)
        me.collect_proj = lambda *a: ref.MVSDet.collect_proj(me, *a)
        ns = dict(ref.__dict__)
        ns.update(self=me, x=feature, feature=feature.clone(), img_meta=meta, stride=4, num_src=N)
        exec(compile(block, "mvsdet.py:416-467", "exec"), ns)
        return ns

    plain = {t: run(t)["volume_variance"] for t in (True, False)}          # the unpatched reference, ATen-CPU

    calls = {"sweep": 0, "lift": 0}

    def sweep_stub(feat, ids, proj, depth):
        calls["sweep"] += 1
        return torch.from_numpy(oracle.plane_sweep_variance(feat, ids, proj, depth, mode=0))

    def lift_stub(packed, points, projection, est_depth, est_dens, n, first, c, h, w, vz):
        calls["lift"] += 1
        o = oracle.backproject_weigh(packed.numpy(), points.reshape(3, -1).numpy(), projection.numpy(), est_depth.numpy(),
                                     est_dens.numpy(), vz)
        return torch.from_numpy(o["volume"].sum(0)), torch.from_numpy(o["valid"].sum(0).astype(np.int32))

    monkeypatch.setattr(ops, "plane_sweep_variance", sweep_stub)
    monkeypatch.setattr(ops, "backproject_weigh_sum_shard", lift_stub)
    monkeypatch.setattr(ops, "pack_features", lambda f: f)
    monkeypatch.setattr(F_, "LAZY_ANY_DEVICE", True)
    orig = integration.patch_reference(ref)
    try:
        before = dict(lazywarp.stats)
        for t in (True, False):
            ns = run(t)
            assert not isinstance(ns["volume_variance"], lazywarp.LazyVolume)
            np.testing.assert_allclose(ns["volume_variance"].numpy(), plain[t].numpy(), rtol=0, atol=1e-4)
        assert lazywarp.stats["fused"] == before["fused"] + 2 and lazywarp.stats["materialized"] == before["materialized"]
        assert calls["sweep"] == 2
        # the lifting statements (mvsdet.py:499-515) on the deferred (volume, valid) of the patched backproject_Weigh
        height, width = ns["height"], ns["width"]
        r = oracle.depth_prob_topk(synthetic.make_cost_logits(N, D, hw, seed=21, sharp=2.0)[:, 0].numpy(),
                                   synthetic.make_cost_logits(N, D, hw, seed=21, sharp=2.0)[:, 1].numpy(), 0.2, 4.8 / D, 3)
        ed = torch.from_numpy(r["est_depth"][:, :, :height, :width]).reshape(N, 3, -1).transpose(2, 1).unsqueeze(2)
        en = torch.from_numpy(r["est_dens"][:, :, :height, :width]).reshape(N, 3, -1).transpose(2, 1).unsqueeze(2)
        projection = ref.MVSDet._compute_projection(meta, 4, None)
        points = ref.get_points(n_voxels=torch.tensor([40, 40, 16]), voxel_size=torch.tensor([0.16, 0.16, 0.2]),
                                origin=torch.tensor(meta["lidar2img"]["origin"]))
        volume, valid, _, _ = ref.backproject_Weigh(feature[:, :, :height, :width], points, projection, ed, [0.16, 0.16, 0.2], en)
        assert isinstance(volume, lazywarp.LazyVolume)
        ns2 = dict(volume=volume, valid=valid)
        exec(compile(tail, "mvsdet.py:511-515", "exec"), ns2)
        assert calls["lift"] == 1 and lazywarp.stats["materialized"] == before["materialized"]
    finally:
        integration.unpatch_reference(ref, orig)
    vol_ref, val_ref, _, _ = ref.backproject_Weigh(feature[:, :, :height, :width], points, projection, ed, [0.16, 0.16, 0.2], en)
    mean_ref = vol_ref.sum(dim=0) / (val_ref.sum(dim=0) + 1e-8)
    mean_ref[:, val_ref.sum(dim=0)[0] == 0] = .0
    np.testing.assert_allclose(ns2["volume_mean"].numpy(), mean_ref.numpy(), rtol=0, atol=1e-6)
    assert int((ns2["valid"] > 0).sum()) > 100


@pytest.mark.parametrize("cls_name,cfg,arkit", [("NerfDetHead", (128, 6, 18, 3), False), ("ImVoxelHead_ARKit", (128, 7, 17, 3), True)])
def test_head_convolutions_are_pinned_to_the_reference_source_text(cls_name, cfg, arkit):
    """`_init_layers`, `_forward_single` and `forward` of NerfDetHead (nerfdet_head.py:90-118) and of the 7-DoF
    ImVoxelHead_ARKit (:663-700), taken from the REFERENCE FILE ITSELF and executed as methods of a bare nn.Module -- with
    mmcv's `Scale` (a learnable scalar factor) and mmdet's `multi_apply` (map + transpose of the result tuples) spelled out,
    neither package being installed here: the same parameter names and shapes as mvsdet_amd.head.NerfDetHeadConvs, and
    after load_state_dict the same outputs."""
    import os
    import textwrap
    import torch
    from torch import nn, Tensor
    from mvsdet_amd.head import NerfDetHeadConvs
    path = "/root/reference/projects/NeRF-Det/nerfdet/nerfdet_head.py"
    if not os.path.isfile(path):
        pytest.skip("reference tree not mounted")
    src = open(path).read().splitlines()
    cls_line = next(i for i, l in enumerate(src) if l.startswith(f"class {cls_name}("))
    first = next(i for i in range(cls_line, len(src)) if src[i].startswith("    def _init_layers"))
    last = next(i for i in range(first, len(src)) if src[i].startswith("    def loss("))
    block = textwrap.dedent("\n".join(src[first:last]))
    assert "self.conv_center = nn.Conv3d(n_channels, 1, 3, padding=1, bias=False)" in block
    assert "multi_apply(self._forward_single, x, self.scales)" in block
    assert ("reg_final[:, 6:]" in block) == arkit

    class Scale(nn.Module):                      # mmcv.cnn.Scale
        def __init__(self, scale=1.0):
            super().__init__()
            self.scale = nn.Parameter(torch.tensor(scale, dtype=torch.float))

        def forward(self, x):
            return x * self.scale

    def multi_apply(func, *args):                # mmdet.models.utils.multi_apply without its kwargs
        return tuple(map(list, zip(*map(func, *args))))

    ns = dict(nn=nn, torch=torch, Tensor=Tensor, Scale=Scale, multi_apply=multi_apply, normal_init=lambda *a, **k: None,
              bias_init_with_prob=lambda p: 0.0)
    exec(block, ns)
    n_channels, n_reg_outs, n_classes, n_levels = cfg   # configs: mvsdet_res50_2x_low_res_depth.py:40-43, mvsdet_arkit_base.py:41-44

    class RefHead(nn.Module):
        _init_layers, _forward_single, forward = ns["_init_layers"], ns["_forward_single"], ns["forward"]

        def __init__(self):
            super().__init__()
            self._init_layers(n_channels, n_reg_outs, n_classes, n_levels)

    torch.manual_seed(0)
    ref, ours = RefHead().eval(), NerfDetHeadConvs(n_classes, n_levels, n_channels, n_reg_outs, arkit_head=arkit).eval()
    assert {k: tuple(v.shape) for k, v in ref.state_dict().items()} == {k: tuple(v.shape) for k, v in ours.state_dict().items()}
    with torch.no_grad():
        for p in ref.parameters():
            p.copy_(torch.randn_like(p) * 0.05)
    ours.load_state_dict(ref.state_dict())
    xs = [torch.randn(1, n_channels, 8 >> i, 8 >> i, 4 >> i) for i in range(n_levels)]
    with torch.no_grad():
        a, b = ref(xs), ours(xs)
    assert len(a) == len(b) == 3
    for la, lb in zip(a, b):
        for ta, tb in zip(la, lb):
            assert torch.equal(ta, tb)


def test_neck_is_pinned_to_the_reference_source_text():
    """`IndoorImVoxelNeck` and `ResModule`, taken from the REFERENCE FILE ITSELF (mmdet3d/models/necks/imvoxel_neck.py:68-231)
    and executed with mmcv's `ConvModule` spelled out (Conv3d without bias -> BatchNorm3d -> optional ReLU, sub-modules `conv`,
    `bn`, `activate`), `BaseModule` = nn.Module and a no-op registry, none of those packages being installed here: the same
    parameter / buffer names and shapes as mvsdet_amd.neck.IndoorImVoxelNeck, and after load_state_dict the same outputs in
    eval and in train mode."""
    import os
    import textwrap
    import torch
    from torch import nn
    from mvsdet_amd.neck import IndoorImVoxelNeck
    path = "/root/reference/mmdet3d/models/necks/imvoxel_neck.py"
    if not os.path.isfile(path):
        pytest.skip("reference tree not mounted")
    src = open(path).read().splitlines()
    first = next(i for i, l in enumerate(src) if l.startswith("class IndoorImVoxelNeck("))
    block = textwrap.dedent("\n".join(src[first:]))
    assert "class ResModule(nn.Module):" in block and "x = down_outs[i] + x" in block and "x = x + identity" in block

    class ConvModule(nn.Module):                 # mmcv.cnn.ConvModule for conv_cfg Conv3d, norm_cfg BN3d, act_cfg ReLU | None
        def __init__(self, in_channels, out_channels, kernel_size, stride=1, padding=0, conv_cfg=None, norm_cfg=None, act_cfg=None):
            super().__init__()
            assert conv_cfg == dict(type='Conv3d') and norm_cfg == dict(type='BN3d')
            self.conv = nn.Conv3d(in_channels, out_channels, kernel_size, stride=stride, padding=padding, bias=False)
            self.bn = nn.BatchNorm3d(out_channels)
            self.with_activation = act_cfg is not None
            if self.with_activation:
                self.activate = nn.ReLU(inplace=act_cfg.get('inplace', True))

        def forward(self, x):
            x = self.bn(self.conv(x))
            return self.activate(x) if self.with_activation else x

    ns = dict(nn=nn, ConvModule=ConvModule, BaseModule=nn.Module)
    exec(block, ns)
    torch.manual_seed(1)
    ref = ns["IndoorImVoxelNeck"](8, 4, [1, 1, 1])
    ours = IndoorImVoxelNeck(8, 4, [1, 1, 1])
    assert {k: tuple(v.shape) for k, v in ref.state_dict().items()} == {k: tuple(v.shape) for k, v in ours.state_dict().items()}
    with torch.no_grad():
        for k, v in ref.state_dict().items():
            if v.dtype.is_floating_point:
                v.copy_(torch.rand_like(v) + 0.5 if "running_var" in k else torch.randn_like(v) * 0.2)
    ours.load_state_dict(ref.state_dict())
    x = torch.randn(2, 8, 8, 8, 4)
    for training in (False, True):
        ref.train(training); ours.train(training)
        with torch.no_grad():
            a, b = ref(x.clone()), ours(x.clone())
        assert [tuple(t.shape) for t in a] == [(2, 4, 8, 8, 4), (2, 4, 4, 4, 2), (2, 4, 2, 2, 1)]
        for ta, tb in zip(a, b):
            assert torch.equal(ta, tb)


def _oracle_backed_ops(monkeypatch, oracle):
    """The device operators of mvsdet_amd.ops replaced by the CPU oracle / ATen restatements (this container has no GPU):
    what the patched reference and MVSDetHotPath.forward_scene launch is counted, and computed by the checker."""
    import numpy as np
    import torch
    from mvsdet_amd import functional as F_, ops
    calls = {"sweep": 0, "lift": 0, "depth": 0}

    def sweep(feat, ids, proj, depth):
        calls["sweep"] += 1
        return torch.from_numpy(oracle.plane_sweep_variance(feat.detach(), ids, proj, depth, mode=0))

    def sweep_packed(packed, ids, proj, depth, c, h, w):
        return sweep(packed, ids, proj, depth)

    def sample(prob, off, near, interval, topk):      # a6 + a7 as SURVEY appendix A states them (bit-exact there)
        calls["depth"] += 1
        dens, idx = prob.topk(topk, dim=1)
        est = idx.float() * np.float32(interval) + np.float32(near) + torch.gather(off, 1, idx) * np.float32(interval)
        d = torch.arange(prob.shape[1], dtype=torch.float32).view(1, -1, 1, 1)
        avg = (prob * (np.float32(near) + (d + off) * np.float32(interval))).sum(1)
        return est, dens, idx.int(), avg

    def depth_prob_topk(cost_reg, off_logit, near, interval, topk):
        prob, off = torch.softmax(cost_reg, 1), torch.sigmoid(off_logit)
        est, dens, idx, avg = sample(prob, off, near, interval, topk)
        return prob, off, est, dens, idx, avg

    def lift_sum(packed, points, projection, est_depth, est_dens, n, first, c, h, w, vz):
        calls["lift"] += 1
        o = oracle.backproject_weigh(packed.numpy()[:, :, :est_depth.shape[2], :est_depth.shape[3]], points.reshape(3, -1).numpy(),
                                     projection.numpy(), est_depth.numpy(), est_dens.numpy(), vz)
        return torch.from_numpy(o["volume"].sum(0)), torch.from_numpy(o["valid"].sum(0).astype(np.int32))

    def lift_mean(feat, packed, points, projection, est_depth, est_dens, hf, wf, vz):
        calls["lift"] += 1
        o = oracle.backproject_weigh_mean(feat.numpy(), points.reshape(3, -1).numpy(), projection.numpy(), est_depth.numpy(),
                                          est_dens.numpy(), vz)
        return torch.from_numpy(o["volume_mean"]), torch.from_numpy(o["valid_count"].astype(np.int32))

    monkeypatch.setattr(ops, "plane_sweep_variance", sweep)
    monkeypatch.setattr(ops, "plane_sweep_variance_packed", sweep_packed)
    monkeypatch.setattr(ops, "sample_depth_prob", sample)
    monkeypatch.setattr(ops, "depth_prob_topk", depth_prob_topk)
    monkeypatch.setattr(ops, "backproject_weigh_sum_shard", lift_sum)
    monkeypatch.setattr(ops, "backproject_weigh_mean", lift_mean)
    monkeypatch.setattr(ops, "pack_features", lambda f: f)
    monkeypatch.setattr(F_, "LAZY_ANY_DEVICE", True)
    return calls


@pytest.mark.parametrize("training", [False, True])
def test_reference_extract_feat_itself_runs_through_the_patch(reference, oracle, monkeypatch, training):
    """The REAL thing once: an instance of the reference's own `MVSDet` class, its own `extract_feat`
    (mvsdet.py:336-698) called as it stands -- 2-D backbone / neck, 3-D neck and NVS branch out of the way (stub modules,
    ray_batch=None), `.cuda()` of the depth-scale helpers (:1283-1313) made a no-op -- first unpatched (pure reference,
    ATen-CPU), then under `integration.patch_reference` with the device operators backed by the oracle.  The patched run
    must launch exactly ONE fused sweep and ONE fused lifting per scene and materialise nothing (lazywarp.stats), agree
    with the pure reference, and equal MVSDetHotPath.forward_scene on the same scene."""
    from types import SimpleNamespace
    import numpy as np
    import torch
    from mvsdet_amd import integration, lazywarp, synthetic
    from mvsdet_amd.costreg import CostRegNet3DGS
    from mvsdet_amd.hotpath import MVSDetHotPath
    ref, _ = reference
    N, C, D, hw = 5, 8, 8, (60, 80)
    meta = synthetic.make_img_meta(N, hw, seed=31)
    feature = synthetic.make_features(N, C, hw, seed=31)
    torch.manual_seed(3)
    net = CostRegNet3DGS(C, base=8).eval()
    with torch.no_grad():
        net.prob.weight.mul_(40.0)          # a soft-max with a clear ranking: the top-3 must not hinge on 1e-6
    monkeypatch.setattr(torch.Tensor, "cuda", lambda self, *a, **k: self)

    def make_detector():
        det = ref.MVSDet.__new__(ref.MVSDet)                     # the reference class; __init__ needs mmengine's registry
        torch.nn.Module.__init__(det)
        det.backbone = lambda img: feature                        # the 2-D stages are not ours: hand out the feature maps
        det.neck = lambda x: [x]
        det.neck_3d = lambda x: x
        det.head_2d = None
        det.n_voxels, det.voxel_size, det.near_far_range, det.topk = [40, 40, 16], [0.16, 0.16, 0.2], [0.2, 5.0], 3
        det.gs_cfg = SimpleNamespace(num_monocular_samples=D)
        det.depth_interval = (5.0 - 0.2) / D                      # mvsdet.py:221-225
        det.depth_values = np.arange(0.2, 5.0, det.depth_interval, dtype=np.float32)
        det.cost_regularization = net
        det.train(training)
        net.eval()                                                # same BatchNorm behaviour in both runs
        return det

    inputs = {"imgs": torch.zeros(1, N, 3, 240, 320)}
    samples = [SimpleNamespace(metainfo=meta)]

    def run(det):
        with torch.no_grad():
            return det.extract_feat(inputs, samples, "test" if not training else "train")

    plain = run(make_detector())                                  # the unpatched reference
    calls = _oracle_backed_ops(monkeypatch, oracle)
    orig = integration.patch_reference(ref)
    try:
        before = dict(lazywarp.stats)
        got = run(make_detector())
        assert lazywarp.stats["fused"] == before["fused"] + 2, lazywarp.stats            # one sweep + one lifting
        assert lazywarp.stats["materialized"] == before["materialized"], lazywarp.stats   # nothing fell back
        assert calls["sweep"] == 1 and calls["lift"] == 1
    finally:
        integration.unpatch_reference(ref, orig)
    x0, v0 = plain[0][0], plain[1][0]
    x1, v1 = got[0][0], got[1][0]
    assert x1.shape == x0.shape == (C, 40, 40, 16) and v1.shape == v0.shape
    same = (v0 == v1)[0]
    assert float((~same).float().mean()) < 1e-3 and int((v0 > 0).sum()) > 500           # counts: bit-equal but for near-ties
    np.testing.assert_allclose(x1[:, same].numpy(), x0[:, same].numpy(), rtol=0, atol=1e-4)
    # the scene driver on the same operators: the same volume and counts
    hp = MVSDetHotPath([40, 40, 16], [0.16, 0.16, 0.2], [0.2, 5.0], D, topk=3, cost_regularization=net)
    with torch.no_grad():
        out = hp.forward_scene(feature, meta)
    assert torch.equal(out["valid"].float(), v1)
    np.testing.assert_allclose(out["volume"].numpy(), x1.numpy(), rtol=0, atol=2e-6)
