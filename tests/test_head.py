"""The detection head's convolutions (mvsdet_amd/head.py; nerfdet_head.py:94-118): structure and parameter names on the CPU,
the fused MFMA route against ATen-CPU on the GPU."""
import numpy as np
import pytest
import torch


def _randomise(m, seed):
    g = torch.Generator().manual_seed(seed)
    with torch.no_grad():
        for p in m.parameters():
            p.copy_(torch.randn(p.shape, generator=g) * (0.05 if p.dim() > 1 else 0.3))
    return m.eval()


def test_head_structure_and_reference_forward():
    from mvsdet_amd.head import NerfDetHeadConvs
    m = NerfDetHeadConvs(18, 3, 128, 6)
    names = set(dict(m.named_parameters()))
    assert names == {"conv_center.weight", "conv_reg.weight", "conv_cls.weight", "conv_cls.bias",
                     "scales.0.scale", "scales.1.scale", "scales.2.scale"}
    assert tuple(m.conv_center.weight.shape) == (1, 128, 3, 3, 3) and tuple(m.conv_reg.weight.shape) == (6, 128, 3, 3, 3)
    assert tuple(m.conv_cls.weight.shape) == (18, 128, 3, 3, 3) and m.conv_center.bias is None and m.conv_reg.bias is None
    m.init_weights()
    np.testing.assert_allclose(m.conv_cls.bias.detach().numpy(), np.full(18, -np.log(99.0), np.float32), rtol=1e-6)
    # nerfdet_head.py:110-118 restated with functional ops
    _randomise(m, 1)
    F = torch.nn.functional
    xs = [torch.randn(1, 128, 8 >> i, 8 >> i, 4 >> i, generator=torch.Generator().manual_seed(i)) for i in range(3)]
    with torch.no_grad():
        centers, regs, clss = m(xs)
        for i, x in enumerate(xs):
            np.testing.assert_allclose(centers[i].numpy(), F.conv3d(x, m.conv_center.weight, padding=1).numpy(), rtol=1e-6, atol=1e-6)
            np.testing.assert_allclose(regs[i].numpy(), torch.exp(F.conv3d(x, m.conv_reg.weight, padding=1) * m.scales[i].scale).numpy(),
                                       rtol=1e-6, atol=1e-6)
            np.testing.assert_allclose(clss[i].numpy(), F.conv3d(x, m.conv_cls.weight, m.conv_cls.bias, padding=1).numpy(), rtol=1e-6, atol=1e-6)
    assert NerfDetHeadConvs.flops([40, 40, 16]) == 54.0 * 128 * 25 * (25600 + 3200 + 400)


@pytest.mark.gpu
@pytest.mark.parametrize("n_classes,n_reg,ch,grid,arkit", [(18, 6, 128, (40, 40, 16), False), (17, 7, 64, (12, 20, 8), True)])
def test_head_fused_mfma_route_vs_aten(gpu, n_classes, n_reg, ch, grid, arkit):
    from mvsdet_amd.head import NerfDetHeadConvs
    m = _randomise(NerfDetHeadConvs(n_classes, 3, ch, n_reg, arkit_head=arkit), 2)
    with torch.no_grad():
        for i, s in enumerate(m.scales):
            s.scale.fill_(0.5 + 0.25 * i)
    xs = [torch.randn((1, ch) + tuple(g >> i for g in grid), generator=torch.Generator().manual_seed(10 + i)) for i in range(3)]
    with torch.no_grad():
        ref = m(xs)                                   # the framework's layers, ATen-CPU
        got = m.to(gpu)([x.to(gpu) for x in xs])      # one fused MFMA convolution per level
    for a_list, b_list in zip(got, ref):
        for a, b in zip(a_list, b_list):
            assert a.shape == b.shape
            np.testing.assert_allclose(a.cpu().numpy(), b.numpy(), rtol=1e-4, atol=1e-4 * max(1.0, float(b.abs().max())))
    # a weight update invalidates the cached fused weight
    with torch.no_grad():
        m.conv_center.weight.mul_(2.0)
        got2 = m([x.to(gpu) for x in xs])[0][0]
    np.testing.assert_allclose(got2.cpu().numpy(), 2.0 * ref[0][0].numpy(), rtol=1e-4, atol=1e-4 * max(1.0, float(ref[0][0].abs().max())))
