"""`--amp` (tools/train.py:24-28; mmengine wraps the forward pass in torch.autocast): the hot path CASTS, on purpose.

Under autocast the 2-D backbone hands out float16 maps.  torch's own autocast rule sends the reference's `grid_sampler`
(module.py:142) and `softmax` (mvsdet.py:472) to float32; this package does the same for the whole path: the operators and the
reference-named mirrors cast low-precision floating inputs to float32 while autocast is on and return float32, the package's
modules (cost network, neck, head) switch autocast off for their call.  Outside autocast a non-float32 tensor is still refused
with a TypeError that names the dtype (the fp16-STORAGE sweep is a separate, explicit entry point).  INTEGRATION.md section 2.
"""
import numpy as np
import pytest
import torch


# --------------------------------------------------------------------------------------------------------------- CPU
def test_amp_fp32_casts_only_while_autocast_is_on(monkeypatch):
    from mvsdet_amd import functional as F_
    h, f, i = torch.zeros(3, dtype=torch.float16), torch.zeros(3), torch.zeros(3, dtype=torch.int64)
    assert F_.amp_fp32(h) is h and F_.amp_fp32(h, f)[0] is h                 # autocast off: untouched
    monkeypatch.setattr(torch, "is_autocast_enabled", lambda *a: True)
    a, b, c, d = F_.amp_fp32(h, f, i, torch.zeros(2, dtype=torch.bfloat16))
    assert a.dtype == torch.float32 and b is f and c is i and d.dtype == torch.float32
    assert F_.amp_fp32(h).dtype == torch.float32


def test_mirrors_hand_float32_to_the_operators_under_autocast(monkeypatch):
    """functional.homo_warping / backproject_Weigh and the patched sample_depth_prob / compute_avg_depth with float16 inputs while
    autocast is on: what reaches the device operators is float32 (the operators are recorded, not run: no GPU here)."""
    from types import SimpleNamespace
    from mvsdet_amd import functional as F_, integration, ops
    seen = {}

    def rec(name, ret):
        def fn(*args, **kw):
            seen[name] = [a.dtype for a in args if isinstance(a, torch.Tensor)]
            return ret
        return fn
    monkeypatch.setattr(torch, "is_autocast_enabled", lambda *a: True)
    monkeypatch.setattr(ops, "homo_warp", rec("homo_warp", torch.zeros(1)))
    monkeypatch.setattr(ops, "backproject_weigh", rec("backproject_weigh", (torch.zeros(2, 4, 8), torch.zeros(2, 1, 8, dtype=torch.bool))))
    monkeypatch.setattr(ops, "sample_depth_prob", rec("sample_depth_prob", (0, 0, 0, 0)))
    eye = torch.eye(4).repeat(2, 1, 1)
    F_.homo_warping(torch.zeros(2, 4, 6, 8, dtype=torch.float16), eye, eye, torch.ones(2, 3, dtype=torch.float16))
    assert seen["homo_warp"] == [torch.float32] * 3
    F_.backproject_Weigh(torch.zeros(2, 4, 6, 8, dtype=torch.float16), torch.zeros(3, 2, 2, 2), torch.zeros(2, 3, 4),
                         torch.zeros(2, 48, 1, 3, dtype=torch.float16), [0.1, 0.1, 0.2], torch.zeros(2, 48, 1, 3, dtype=torch.float16))
    assert seen["backproject_weigh"] == [torch.float32] * 5
    ns = SimpleNamespace(near_far_range=[0.2, 5.0], depth_interval=0.4)
    integration.PATCHED_METHODS["sample_depth_prob"](ns, torch.zeros(2, 12, 6, 8, dtype=torch.float16), torch.zeros(2, 12, 6, 8, dtype=torch.bfloat16))
    assert seen["sample_depth_prob"] == [torch.float32] * 2
    integration.PATCHED_METHODS["compute_avg_depth"](ns, torch.zeros(2, 12, 6, 8, dtype=torch.float16), torch.zeros(2, 12, 6, 8))
    assert seen["sample_depth_prob"] == [torch.float32] * 2


def test_every_forward_operator_of_the_path_has_the_fp32_autocast_rule():
    from mvsdet_amd import ops
    names = {op._qualname.split("::")[1] for op in ops.AUTOCAST_FP32_OPS}
    assert names == {"homo_warp", "plane_sweep_variance", "plane_sweep_variance_keep", "depth_prob_topk", "sample_depth_prob",
                     "backproject_weigh", "backproject_weigh_mean"}


# --------------------------------------------------------------------------------------------------------------- GPU
@pytest.mark.gpu
def test_hot_path_under_autocast_computes_in_float32(gpu):
    """The operators, the mirrors, the scene driver and the package's modules under torch.autocast('cuda', float16) with float16
    feature maps: float32 results, bit-equal to the float32 call on the same (float16-representable) values; gradients reach
    the float16 leaf.  Without autocast the same float16 tensor is refused."""
    from mvsdet_amd import functional as F_, ops, synthetic
    from mvsdet_amd.costreg import CostRegNet3DGS
    from mvsdet_amd.hotpath import MVSDetHotPath
    N, C, D, hw = 4, 32, 8, (24, 32)
    torch.manual_seed(0)
    net = CostRegNet3DGS(C, base=8).to(gpu).eval()
    hp = MVSDetHotPath([16, 16, 8], [.4, .4, .4], [0.2, 5.0], D, cost_regularization=net)
    meta = synthetic.make_img_meta(N, hw, seed=9)
    geo = hp.prepare_scene(meta, gpu)
    half = synthetic.make_features(N, C, hw, seed=9).to(gpu).half()
    full = half.float()
    with pytest.raises(TypeError, match="float32"):
        ops.plane_sweep_variance(half, geo.neighbor_ids, geo.proj_rel, geo.depth_values)
    with torch.no_grad():
        want = ops.plane_sweep_variance(full, geo.neighbor_ids, geo.proj_rel, geo.depth_values)
        want_out = hp.forward_scene(full, meta)
        with torch.autocast("cuda", dtype=torch.float16):
            got = ops.plane_sweep_variance(half, geo.neighbor_ids, geo.proj_rel, geo.depth_values)
            warped = F_.homo_warping(half[geo.neighbor_ids[:, 0]], torch.eye(4).repeat(N, 1, 1), torch.eye(4).repeat(N, 1, 1),
                                     geo.depth_values.half())
            out = hp.forward_scene(half, meta)
            logits16 = net(want.half())               # a low-precision volume: cast up, autocast off inside the module
        assert got.dtype == torch.float32 and torch.equal(got, want)
        assert warped.dtype == torch.float32 and tuple(warped.shape) == (N, C, D) + hw
        for k in ("variance", "prob_volume", "est_depth", "est_densities", "depth_coding", "volume"):
            assert out[k].dtype == torch.float32 and torch.equal(out[k], want_out[k]), k
        assert torch.equal(out["valid"], want_out["valid"])
        assert logits16.dtype == torch.float32 and torch.equal(logits16, net(want.half().float()))
    # training under autocast: the gradient arrives at the float16 leaf through the cast.  The loss is scaled as AMP's GradScaler
    # scales it (gradients of ~1e-6 sit in float16's subnormal range, 6e-8 apart)
    scale = 65536.0
    leaf = half.clone().requires_grad_(True)
    with torch.autocast("cuda", dtype=torch.float16):
        v = ops.plane_sweep_variance(leaf, geo.neighbor_ids, geo.proj_rel, geo.depth_values)
    (v.square().mean() * scale).backward()
    ref = full.clone().requires_grad_(True)
    (ops.plane_sweep_variance(ref, geo.neighbor_ids, geo.proj_rel, geo.depth_values).square().mean() * scale).backward()
    assert leaf.grad.dtype == torch.float16 and torch.isfinite(leaf.grad).all()
    np.testing.assert_allclose(leaf.grad.float().cpu().numpy(), ref.grad.cpu().numpy(), rtol=2e-3, atol=1e-4 * float(ref.grad.abs().max()))
