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
