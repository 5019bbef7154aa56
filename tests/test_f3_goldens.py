"""Row f-3 (3-D neck, head convolutions) and the training route of row f-1 pinned to the REFERENCE's numbers
(missing #4 / weak #1-#2).  The fixtures G10 / G11 / G12 hold outputs of the reference classes themselves
(tests/golden/make_goldens.py executes `IndoorImVoxelNeck`, `NerfDetHead`, `ImVoxelHead_ARKit` and `CostRegNet_3DGS` from
the files where they lie); weights and inputs come from the committed integer LCG, so only outputs are stored.

CPU: the package's modules on the framework's layers against the fixtures (same ATen operators as the reference ran).
GPU: the WHOLE shipped neck, both heads and a training step of the cost network on the HIP kernels, 1e-4 of each
tensor's scale (north_star tolerance).  Nothing here reads /root/reference.
"""
import sys

import numpy as np
import pytest
import torch

from conftest import GOLDEN, load_golden

sys.path.insert(0, GOLDEN)
from lcg import lcg_fill_state, lcg_uniform, _lcg_uniform_loop  # noqa: E402

TOL = 1e-4


def test_vectorised_lcg_equals_its_definition():
    for n, seed in ((1, 0), (5, 3), (65536, 8), (65537, 88), (140001, 8003)):
        assert np.array_equal(lcg_uniform(n, seed), _lcg_uniform_loop(n, seed)), (n, seed)


# ------------------------------------------------------------------------------------------------------------ inputs
def _neck_and_input(g):
    from mvsdet_amd.neck import IndoorImVoxelNeck
    net = IndoorImVoxelNeck(256, 128, [1, 1, 1]).eval()
    assert sorted(net.state_dict()) == [str(k) for k in g["keys"]]
    with torch.no_grad():
        lcg_fill_state(net, int(g["weight_seed"]))
    shape = tuple(int(v) for v in g["in_shape"])
    x = torch.from_numpy(lcg_uniform(int(np.prod(shape)), int(g["input_seed"]))).reshape(shape)
    keep = torch.from_numpy(lcg_uniform(int(np.prod(shape[2:])), int(g["mask_seed"]))).reshape((1, 1) + shape[2:]) > float(g["mask_threshold"])
    return net, x * keep


def _check_neck(outs, block0, g, tol):
    scales = g["level_scales"]
    got = [outs[0][:, ::2, ::2, ::2, ::2], outs[1][:, ::2], outs[2]]
    for i, (a, key) in enumerate(zip(got, ("level0", "level1", "level2"))):
        err = float(np.abs(a.cpu().numpy() - g[key]).max())
        assert err <= tol * max(1.0, float(scales[i])), f"{key}: max |d| {err:.3e}"
    if block0 is not None:
        b = g["block0"]
        err = float(np.abs(block0[:, ::4, ::2, ::2, ::2].cpu().numpy() - b).max())
        assert err <= tol * max(1.0, float(np.abs(b).max())), f"block0: max |d| {err:.3e}"


def _head_and_inputs(g, tag):
    from mvsdet_amd.head import NerfDetHeadConvs
    ch, n_reg, n_cls = (128, 6, 18) if tag == "scannet" else (128, 7, 17)   # mvsdet_res50_2x_low_res_depth.py:40-43, mvsdet_arkit_base.py:41-44
    head = NerfDetHeadConvs(n_cls, 3, ch, n_reg, arkit_head=(tag == "arkit")).eval()
    with torch.no_grad():
        lcg_fill_state(head, int(g["weight_seed"]))
        for i, s in enumerate(head.scales):
            s.scale.fill_(0.5 + 0.25 * i)
    xs = [torch.from_numpy(lcg_uniform(ch * (40 >> i) * (40 >> i) * (16 >> i), int(g["input_seed"]) + i)).reshape(1, ch, 40 >> i, 40 >> i, 16 >> i)
          for i in range(3)]
    return head, xs


def _check_heads(res, g, tag, tol):
    centers, regs, clss = res
    for i in range(3):
        for name, t in (("center", centers[i]), ("reg", regs[i]), ("cls", clss[i])):
            ref = g[f"{tag}_{name}{i}"]
            a = t[:, :, ::2, ::2, ::2] if i == 0 else t
            assert tuple(a.shape) == ref.shape
            err = float(np.abs(a.cpu().numpy() - ref).max())
            assert err <= tol * max(1.0, float(np.abs(ref).max())), f"{tag} {name}{i}: max |d| {err:.3e}"


def _costreg_train_step(g, device, precision):
    from mvsdet_amd.costreg import CostRegNet3DGS
    net = CostRegNet3DGS(256, 64).train()
    net.matrix_precision = precision
    with torch.no_grad():
        lcg_fill_state(net, int(g["weight_seed"]))
    net = net.to(device)
    shape = tuple(int(v) for v in g["in_shape"])
    x = torch.from_numpy(lcg_uniform(int(np.prod(shape)), int(g["input_seed"]))).reshape(shape).abs().to(device).requires_grad_(True)
    y = net(x)
    R = torch.from_numpy(lcg_uniform(y.numel(), int(g["r_seed"]))).reshape(y.shape).to(device)
    (y * R).sum().backward()
    return net, x, y


def _check_costreg_grads(net, x, y, g, tol_fwd, elementwise_tol, norm_tol, outlier_share):
    """logits and BatchNorm statistics element-wise; gradients element-wise on all but `outlier_share` of the sampled
    elements (an activation within the forward noise of zero flips its ReLU decision and moves a few gradient entries),
    and every gradient tensor in norm."""
    np.testing.assert_allclose(y.detach().cpu().numpy(), g["logits"], rtol=0, atol=tol_fwd * max(1.0, float(np.abs(g["logits"]).max())))
    for k, b in net.named_buffers():
        if k.endswith("running_mean") or k.endswith("running_var"):     # every training fixture stores them
            np.testing.assert_allclose(b.cpu().numpy(), g["b:" + k], rtol=1e-4, atol=1e-5, err_msg=k)
    items = [("grad_input", x.grad.reshape(-1)[::97].cpu().numpy(), g["grad_input"], None)]
    params = dict(net.named_parameters())
    assert sorted(params) == [str(k) for k in g["param_keys"]]
    for k in sorted(params):
        gr = params[k].grad.reshape(-1)
        items.append((k, gr[::int(g["s:" + k])].cpu().numpy(), g["g:" + k], (float((gr.double() ** 2).sum()), float(g["n:" + k]))))
    for name, a, ref, norms in items:
        scale = max(float(np.abs(ref).max()), 1e-12)
        if elementwise_tol is not None:
            bad = np.abs(a - ref) > elementwise_tol * scale
            assert bad.mean() <= outlier_share, f"{name}: {bad.mean():.2e} of the sampled gradient entries off by more than {elementwise_tol:g} x scale"
        rel = float(np.linalg.norm((a - ref).astype(np.float64)) / max(np.linalg.norm(ref.astype(np.float64)), 1e-30))
        assert rel <= norm_tol, f"{name}: relative error of the sample in norm {rel:.2e}"
        if norms is not None:
            assert abs(norms[0] ** 0.5 - norms[1] ** 0.5) <= norm_tol * max(norms[1] ** 0.5, 1e-30), f"{name}: norm of the whole gradient"


# --------------------------------------------------------------------------------------------------------------- CPU
def test_g10_neck_on_the_framework_layers():
    g = load_golden("g10_neck")
    net, x = _neck_and_input(g)
    with torch.no_grad():
        _check_neck(net(x), net.down_layer_0(x), g, 1e-5)


@pytest.mark.parametrize("tag", ["scannet", "arkit"])
def test_g11_heads_on_the_framework_layers(tag):
    g = load_golden("g11_heads")
    head, xs = _head_and_inputs(g, tag)
    with torch.no_grad():
        _check_heads(head(xs), g, tag, 1e-5)


def test_g12_cost_network_gradients_on_the_framework_layers():
    g = load_golden("g12_cost_regularisation_grads")
    net, x, y = _costreg_train_step(g, "cpu", "fp32")
    _check_costreg_grads(net, x, y, g, 1e-5, 1e-4, 1e-4, 0.0)


def test_g12b_margin_fixture_on_the_framework_layers():
    """G12b = G12 on an input whose pre-ReLU activations all keep a margin from zero (make_goldens.g12b_relu_margin_grads):
    the fixture says so itself, and the package's module on ATen reproduces the reference's gradients on it."""
    g = load_golden("g12b_cost_regularisation_grads_margin")
    assert float(g["layer_margins"].min()) > float(g["margin"]) >= 3e-5 and len(g["layer_margins"]) == 7
    net, x, y = _costreg_train_step(g, "cpu", "fp32")
    _check_costreg_grads(net, x, y, g, 1e-5, 1e-4, 1e-4, 0.0)


def _g12c_masks(g, net, device):
    """{BatchNorm module of `net`: the reference's ReLU decisions for it} from the fixture's packed bits."""
    mods = dict(net.named_modules())
    out = {}
    for name in (str(n) for n in g["mask_names"]):
        shape = tuple(int(v) for v in g["maskshape:" + name])
        bits = np.unpackbits(g["mask:" + name])[: int(np.prod(shape))].astype(bool).reshape(shape)
        out[mods[name]] = torch.from_numpy(bits).to(device)
    assert len(out) == 7
    return out


def test_g12c_masks_fixture_on_the_framework_layers():
    """G12c (the reference's gradients at a shape where every fused training form is active, with its seven ReLU decisions
    stored): the package's module on ATen reproduces logits, running statistics, gradients AND the decisions themselves."""
    g = load_golden("g12c_cost_regularisation_grads_masks")
    from mvsdet_amd.costreg import CostRegNet3DGS
    net = CostRegNet3DGS(256, 64).train()
    with torch.no_grad():
        lcg_fill_state(net, int(g["weight_seed"]))
    seen = {}
    for name, m in net.named_modules():
        if isinstance(m, torch.nn.BatchNorm3d):
            m.register_forward_hook(lambda mod, inp, out, name=name: seen.__setitem__(name, (out.detach() > 0).clone()))
    shape = tuple(int(v) for v in g["in_shape"])
    x = torch.from_numpy(lcg_uniform(int(np.prod(shape)), int(g["input_seed"]))).reshape(shape).abs().requires_grad_(True)
    y = net(x)
    R = torch.from_numpy(lcg_uniform(y.numel(), int(g["r_seed"]))).reshape(y.shape)
    (y * R).sum().backward()
    _check_costreg_grads(net, x, y, g, 1e-5, 1e-4, 1e-4, 0.0)
    want = _g12c_masks(g, net, "cpu")
    mods = dict(net.named_modules())
    for name, m in seen.items():
        differ = int((m != want[mods[name]]).sum())
        assert differ <= 2, f"{name}: {differ} ReLU decisions differ from the reference's"   # same ATen operators: none expected


# --------------------------------------------------------------------------------------------------------------- GPU
@pytest.mark.gpu
def test_g10_whole_shipped_neck_on_the_hip_kernels(gpu):
    """IndoorImVoxelNeck(256, 128, [1,1,1]) on the (1,256,40,40,16) volume, EVERY level, against the reference class's
    outputs: bf16x3 3x3x3 layers (split over the input channels on the small levels), GEMM shortcut / up-sampling."""
    g = load_golden("g10_neck")
    net, x = _neck_and_input(g)
    net = net.to(gpu)
    with torch.no_grad():
        xd = x.to(gpu)
        _check_neck(net(xd), net.down_layer_0(xd), g, TOL)


@pytest.mark.gpu
def test_g10_neck_on_a_batch_of_two(gpu):
    """G10 at batch 2 (mvsdet.py:695-696 stacks the scenes): the fixture's input twice -- both rows must reproduce the reference
    class's outputs -- and the fixture's input beside a shifted-seed one, whose row must equal that input run alone."""
    g = load_golden("g10_neck")
    net, x = _neck_and_input(g)
    net = net.to(gpu)
    shape = tuple(x.shape)
    other = torch.from_numpy(lcg_uniform(int(np.prod(shape)), int(g["input_seed"]) + 1000)).reshape(shape)
    with torch.no_grad():
        xd, od = x.to(gpu), other.to(gpu)
        twice = net(torch.cat((xd, xd), 0))
        for row in (0, 1):
            _check_neck([t[row:row + 1] for t in twice], None, g, TOL)
        mixed = net(torch.cat((od, xd), 0))
        _check_neck([t[1:2] for t in mixed], None, g, TOL)
        alone = net(od)
        for a, b in zip(mixed, alone):
            assert float((a[0:1] - b).abs().max()) <= 1e-5 * max(1.0, float(b.abs().max()))


@pytest.mark.gpu
@pytest.mark.parametrize("tag", ["scannet", "arkit"])
def test_g11_heads_on_a_batch_of_two(gpu, tag):
    g = load_golden("g11_heads")
    head, xs = _head_and_inputs(g, tag)
    head = head.to(gpu)
    with torch.no_grad():
        res = head([torch.cat((x.to(gpu), x.to(gpu) * 0.5), 0) for x in xs])
        _check_heads(tuple([t[0:1] for t in part] for part in res), g, tag, TOL)
        half = head([x.to(gpu) * 0.5 for x in xs])
        for pa, pb in zip(res, half):
            for a, b in zip(pa, pb):
                assert float((a[1:2] - b).abs().max()) <= 1e-5 * max(1.0, float(b.abs().max()))


@pytest.mark.gpu
@pytest.mark.parametrize("tag", ["scannet", "arkit"])
def test_g11_heads_on_the_hip_kernels(gpu, tag):
    g = load_golden("g11_heads")
    head, xs = _head_and_inputs(g, tag)
    head = head.to(gpu)
    with torch.no_grad():
        _check_heads(head([x.to(gpu) for x in xs]), g, tag, TOL)


@pytest.mark.gpu
@pytest.mark.parametrize("precision", ["fp32", "bf16x3"])
def test_g12_cost_network_training_step_on_the_hip_kernels(gpu, precision):
    """A training step of CostRegNet3DGS (forward, input and weight gradients, training-mode BatchNorm on our kernels) against
    the parameter gradients of the REFERENCE module under autograd.  fp32 route: element-wise 1e-4 of each tensor's scale.
    bf16x3 route: logits and BatchNorm statistics element-wise (1e-4), gradients 2e-2 in norm.  Measured with
    tools/study/train_precision_probe2.py: every operator of the bf16x3 route is within ~1e-5 of its fp32 twin, but the forward
    activations carry that 1e-5 and an activation this close to zero takes the other ReLU branch -- ONE such flip among the N
    elements of a layer moves the layer's gradient by sqrt(2 / N) in norm (5.5e-3 at the 65 536 full-resolution activations of
    this input; 1.6e-3 at the probe's 786 432), and training-mode BatchNorm spreads it over every entry.  The fp32 route has
    1e-6 of forward noise and flips none here.  """
    g = load_golden("g12_cost_regularisation_grads")
    net, x, y = _costreg_train_step(g, gpu, precision)
    if precision == "fp32":
        _check_costreg_grads(net, x, y, g, TOL, TOL, TOL, 0.0)
    else:
        _check_costreg_grads(net, x, y, g, TOL, None, 2e-2, 0.0)


@pytest.mark.gpu
@pytest.mark.parametrize("precision", ["fp32", "bf16x3"])
def test_g12b_training_step_far_from_every_relu_kink(gpu, precision):
    """The pin of the bf16x3 TRAINING route (VERDICT r4 task 6).  On G12's input the route can only be held to 2e-2 in norm because
    one activation within the forward noise of zero flips its ReLU; G12b's input keeps every pre-ReLU value of every layer further
    than 3e-5 x rms from zero (a 7-sigma error would be needed), so here the route is held ELEMENT-WISE: 1e-3 of each gradient
    tensor's scale with no outliers allowed, and 1e-4 in norm; the fp32 route 1e-4 element-wise as on G12.  A wrong term in any
    backward kernel shows at these bounds."""
    g = load_golden("g12b_cost_regularisation_grads_margin")
    net, x, y = _costreg_train_step(g, gpu, precision)
    if precision == "fp32":
        _check_costreg_grads(net, x, y, g, TOL, TOL, TOL, 0.0)
    else:
        _check_costreg_grads(net, x, y, g, TOL, 1e-3, 1e-4, 0.0)


@pytest.mark.gpu
@pytest.mark.parametrize("fused_stats", [True, False])
def test_g12c_training_step_with_the_reference_relu_decisions(gpu, fused_stats, record_property):
    """The default training route (bf16x3, BatchNorm statistics from the producing kernels' epilogues) and its separate-pass twin
    against the REFERENCE module's gradients at (2,256,12,64,32) -- the stride-1 AND the transposed statistics forms are active
    (conv9 reads one 3 x 16 x 8 coarse tile per view, conv11 eight) -- with the reference's own seven ReLU decisions imposed
    (costreg.RELU_MASKS): no activation can take the other branch, so every gradient is held ELEMENT-WISE (VERDICT r5 weak #1,
    next #3).  Logits and running statistics 1e-4; gradients element-wise 1e-3 of each tensor's scale with no outliers (the
    bf16x3 bound of G12b) and 1e-4 in norm."""
    from mvsdet_amd import costreg
    from mvsdet_amd.costreg import CostRegNet3DGS
    g = load_golden("g12c_cost_regularisation_grads_masks")
    net = CostRegNet3DGS(256, 64).train()
    assert net.matrix_precision == "bf16x3"
    with torch.no_grad():
        lcg_fill_state(net, int(g["weight_seed"]))
    net = net.to(gpu)
    shape = tuple(int(v) for v in g["in_shape"])
    x = torch.from_numpy(lcg_uniform(int(np.prod(shape)), int(g["input_seed"]))).reshape(shape).abs().to(gpu).requires_grad_(True)
    was = costreg.FUSED_BN_STATS, costreg.RELU_MASKS
    calls = {"s1": 0, "t": 0}
    from mvsdet_amd import ops
    real_s1, real_t = ops.conv3d_k3_bf16x3_stats, ops.convT3d_k3_s2_bf16x3_stats

    def count(key, fn):
        def wrapped(*a, **k):
            r = fn(*a, **k)
            calls[key] += int(r is not None)
            return r
        return wrapped
    try:
        costreg.FUSED_BN_STATS = fused_stats
        costreg.RELU_MASKS = ("apply", _g12c_masks(g, net, gpu))
        ops.conv3d_k3_bf16x3_stats, ops.convT3d_k3_s2_bf16x3_stats = count("s1", real_s1), count("t", real_t)
        y = net(x)
        R = torch.from_numpy(lcg_uniform(y.numel(), int(g["r_seed"]))).reshape(y.shape).to(gpu)
        (y * R).sum().backward()
    finally:
        costreg.FUSED_BN_STATS, costreg.RELU_MASKS = was
        ops.conv3d_k3_bf16x3_stats, ops.convT3d_k3_s2_bf16x3_stats = real_s1, real_t
    record_property("fused_statistics_calls", dict(calls))
    if fused_stats:   # both forms really ran: three stride-1 layers, two transposed ones
        assert calls["s1"] == 3 and calls["t"] == 2, calls
    else:
        assert calls["s1"] == 0 and calls["t"] == 0, calls
    _check_costreg_grads(net, x, y, g, TOL, 1e-3, 1e-4, 0.0)


@pytest.mark.gpu
def test_sgd_trajectories_of_the_two_training_routes_agree(gpu):
    """30 steps of plain SGD (lr 3e-4: the loss falls by 40 %) on G12's input, on three routes: our fp32 kernels, bf16x3 (the
    default under autograd) and the framework's own fp32 layers (ATen / MIOpen; `hip_backward = False`).  Training amplifies ANY
    difference between two routes step by step -- two fp32 implementations part at the same rate --, so the bound has two parts:
    bf16x3 stays within 1e-4 relative of our fp32 route at EVERY step, and it drifts no further from it than twice what the
    framework's fp32 route drifts from it (measured, profiles/r05_sgd_routes_probe.txt: 3.7e-5 against 3.6e-5 at this rate,
    1.04e-4 against 9.5e-5 at lr 1e-3, 4.5e-3 against 3.5e-3 at lr 3e-3): the split-bf16 matrix path trains like an fp32 one
    (tools/train.py over mvs_models/mvsnet.py:73-113)."""
    from mvsdet_amd.costreg import CostRegNet3DGS
    g = load_golden("g12_cost_regularisation_grads")
    shape = tuple(int(v) for v in g["in_shape"])
    x = torch.from_numpy(lcg_uniform(int(np.prod(shape)), int(g["input_seed"]))).reshape(shape).abs().to(gpu)
    R = torch.from_numpy(lcg_uniform(2 * int(np.prod(shape[2:])) * shape[0], int(g["r_seed"]))).reshape(shape[0], 2, *shape[2:]).to(gpu)
    losses = {}
    for route in ("fp32", "bf16x3", "aten"):
        net = CostRegNet3DGS(256, 64).train()
        net.matrix_precision = "fp32" if route == "aten" else route
        net.hip_backward = route != "aten"
        with torch.no_grad():
            lcg_fill_state(net, int(g["weight_seed"]))
        net = net.to(gpu)
        opt = torch.optim.SGD(net.parameters(), lr=3e-4)
        tr = []
        for step in range(30):
            opt.zero_grad(set_to_none=True)
            loss = ((net(x) - R) ** 2).mean()          # a loss with a minimum: the trajectory descends instead of running away
            loss.backward()
            opt.step()
            tr.append(float(loss.detach()))
        losses[route] = np.array(tr)
    a = losses["fp32"]
    assert a[-1] < 0.7 * a[0], f"the fp32 trajectory does not descend: {a[0]:.4f} -> {a[-1]:.4f}"
    rel = np.abs(a - losses["bf16x3"]) / np.abs(a)
    yard = np.abs(a - losses["aten"]) / np.abs(a)
    assert rel.max() <= 1e-4, f"loss trajectories part: max relative difference {rel.max():.2e} at step {int(rel.argmax())}"
    assert rel.max() <= 2.0 * yard.max() + 1e-6, f"bf16x3 drifts {rel.max():.2e} from the fp32 route, the framework's fp32 layers only {yard.max():.2e}"
