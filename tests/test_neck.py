"""the 3-D neck that consumes the hot path's volume (mmdet3d/models/necks/imvoxel_neck.py:70-231).
CPU: structure, parameter names (what a reference checkpoint's `neck_3d.*` entries are called) and the GEMM forms of
the 1x1x1 shortcut and the kernel-2 transposed convolution against ATen's layers.  GPU: the MFMA route against the
framework's layers (ATen-CPU) on the same weights."""
import numpy as np
import pytest
import torch


def _randomise(m, seed=0):
    g = torch.Generator().manual_seed(seed)
    for name, p in m.named_parameters():
        with torch.no_grad():
            p.copy_(torch.randn(p.shape, generator=g) * (0.5 / max(1, p[0].numel()) ** 0.5 if p.dim() > 1 else 0.3) + (1.0 if name.endswith("bn.weight") or name.endswith(".1.weight") or name.endswith(".4.weight") else 0.0))
    for name, b in m.named_buffers():
        if name.endswith("running_mean"):
            b.copy_(torch.randn(b.shape, generator=g) * 0.1)
        elif name.endswith("running_var"):
            b.copy_(torch.rand(b.shape, generator=g) + 0.5)
    return m.eval()


def test_structure_and_parameter_names():
    from mvsdet_amd.neck import IndoorImVoxelNeck
    m = IndoorImVoxelNeck(256, 128, [1, 1, 1])   # configs/mvsdet_res50_2x_low_res.py: neck_3d
    keys = set(m.state_dict())
    for k in ("down_layer_0.0.conv0.conv.weight", "down_layer_0.0.conv1.bn.running_var", "down_layer_1.0.downsample.conv.weight",
              "down_layer_2.0.downsample.bn.bias", "up_block_1.0.weight", "up_block_2.3.weight", "up_block_2.4.running_mean",
              "out_block_0.0.weight", "out_block_2.1.bias"):
        assert k in keys, k
    assert not any(k.startswith("up_block_0") for k in keys)
    sd = m.state_dict()
    assert sd["down_layer_1.0.conv0.conv.weight"].shape == (512, 256, 3, 3, 3)
    assert sd["down_layer_2.0.downsample.conv.weight"].shape == (1024, 512, 1, 1, 1)
    assert sd["up_block_2.0.weight"].shape == (1024, 512, 2, 2, 2) and sd["out_block_2.0.weight"].shape == (128, 1024, 3, 3, 3)
    n_params = sum(p.numel() for p in m.parameters())
    assert 70e6 < n_params < 85e6          # "~77 M params by my count"
    outs = IndoorImVoxelNeck(8, 4, [1, 2, 1]).eval()(torch.randn(2, 8, 8, 8, 4))
    assert [tuple(o.shape) for o in outs] == [(2, 4, 8, 8, 4), (2, 4, 4, 4, 2), (2, 4, 2, 2, 1)]
    assert abs(IndoorImVoxelNeck.flops(1, [40, 40, 16]) / 1e9 - 490) < 5


def test_gemm_forms_match_aten_layers():
    """The fast route's two GEMM rewrites, evaluated on the CPU: 1x1x1 stride-2 conv + BN, ConvTranspose3d(k=2,s=2) + BN + ReLU."""
    from mvsdet_amd import neck as NK
    g = torch.Generator().manual_seed(3)
    ds = _randomise(NK._ConvModule(6, 10, 1, 2, 0, act=False), 1)
    x = torch.randn(2, 6, 8, 6, 4, generator=g)
    xs = x[:, :, ::2, ::2, ::2]
    scale, shift = NK._bn_affine(ds.bn)
    wmat = ds.conv.weight.detach().reshape(10, 6) * scale[:, None]
    got = torch.baddbmm(shift.view(1, -1, 1), wmat.unsqueeze(0).expand(2, -1, -1), xs.reshape(2, 6, -1)).view(2, 10, 4, 3, 2)
    with torch.no_grad():
        np.testing.assert_allclose(got.numpy(), ds(x).numpy(), rtol=1e-5, atol=1e-5)
    up = _randomise(NK._UpBlock(6, 4), 2)
    x = torch.randn(2, 6, 3, 4, 2, generator=g)
    deconv, bn = up[0], up[1]
    scale, shift = NK._bn_affine(bn)
    wmat = (deconv.weight.detach() * scale.view(1, -1, 1, 1, 1)).permute(2, 3, 4, 1, 0).reshape(8 * 4, 6)
    y = torch.matmul(wmat.unsqueeze(0), x.reshape(2, 6, -1)).view(2, 2, 2, 2, 4, 3, 4, 2)
    y = torch.relu(y.permute(0, 4, 5, 1, 6, 2, 7, 3).reshape(2, 4, 6, 8, 4) + shift.view(1, -1, 1, 1, 1))
    with torch.no_grad():
        ref = torch.relu(bn(deconv(x)))
    np.testing.assert_allclose(y.numpy(), ref.numpy(), rtol=1e-5, atol=1e-5)


@pytest.mark.gpu
@pytest.mark.parametrize("cin,cout,grid,n", [(64, 64, (24, 24, 8), 1), (128, 64, (12, 20, 8), 2)])
def test_neck_mfma_route_vs_aten(gpu, cin, cout, grid, n):
    from mvsdet_amd.neck import IndoorImVoxelNeck
    m = _randomise(IndoorImVoxelNeck(cin, cout, [1, 1, 1]), 5)
    x = torch.randn((n, cin) + grid, generator=torch.Generator().manual_seed(6))
    with torch.no_grad():
        ref = m(x)                     # the framework's layers, ATen-CPU
        got = m.to(gpu)(x.to(gpu))     # MFMA kernels + GEMMs
    for a, b in zip(got, ref):
        assert a.shape == b.shape
        scale = float(b.abs().max())
        np.testing.assert_allclose(a.cpu().numpy(), b.numpy(), rtol=0, atol=1e-4 * max(scale, 1.0))
    # under autograd the module takes the framework's layers (training is not on this route): gradients flow
    m.train()
    xg = x.to(gpu).requires_grad_(True)
    sum(o.square().mean() for o in m(xg)).backward()
    assert torch.isfinite(xg.grad).all() and float(xg.grad.abs().sum()) > 0


@pytest.mark.gpu
def test_neck_shipped_configuration(gpu):
    """in_channels=256, out_channels=128, n_blocks=[1,1,1] on the (256,40,40,16) volume: the first residual block against
    ATen-CPU (90 GFLOP of the 490), all outputs finite and of the reference's shapes."""
    from mvsdet_amd.neck import IndoorImVoxelNeck
    m = _randomise(IndoorImVoxelNeck(256, 128, [1, 1, 1]), 7)
    x = torch.randn((1, 256, 40, 40, 16), generator=torch.Generator().manual_seed(8))
    with torch.no_grad():
        ref0 = m.down_layer_0(x)
        mg = m.to(gpu)
        got0 = mg.down_layer_0(x.to(gpu))
        outs = mg(x.to(gpu))
    np.testing.assert_allclose(got0.cpu().numpy(), ref0.numpy(), rtol=0, atol=1e-4 * max(1.0, float(ref0.abs().max())))
    assert [tuple(o.shape) for o in outs] == [(1, 128, 40, 40, 16), (1, 128, 20, 20, 8), (1, 128, 10, 10, 4)]
    assert all(bool(torch.isfinite(o).all()) for o in outs)


@pytest.mark.gpu
@pytest.mark.parametrize("N,Cin,Cout,grid,stride,res", [(1, 512, 128, (10, 10, 4), 1, False), (1, 96, 64, (20, 20, 8), 1, True),
                                                      (1, 256, 128, (20, 20, 8), 2, False)])
def test_conv_split_over_input_channels(gpu, N, Cin, Cout, grid, stride, res):
    """Small volumes run the 3x3x3 convolution split over the input channels (partial sums + epilogue kernel): through the
    C ABI the split form (workspace given) against the unsplit one (workspace NULL) and against ATen-CPU."""
    from mvsdet_amd import _lib, ops
    lib = _lib.load()
    g = torch.Generator().manual_seed(Cin + stride)
    x = torch.randn((N, Cin) + grid, generator=g)
    w = torch.randn(Cout, Cin, 3, 3, 3, generator=g) / (27 * Cin) ** 0.5
    scale, shift = torch.rand(Cout, generator=g) + 0.5, torch.randn(Cout, generator=g)
    ref = torch.nn.functional.conv3d(x, w, stride=stride, padding=1) * scale.view(1, -1, 1, 1, 1) + shift.view(1, -1, 1, 1, 1)
    residual = torch.randn(ref.shape, generator=g) if res else None
    if res:
        ref = ref + residual
    ref = torch.relu(ref)
    xd, wd, sc, sh = x.to(gpu), ops.permute_conv_weight(w.to(gpu)), scale.to(gpu), shift.to(gpu)
    rd = residual.to(gpu) if res else None
    wbytes = lib.mvsdet_conv3d_k3_mfma_workspace_bytes(N, Cin, Cout, *grid, stride)
    assert wbytes > 0, "this shape is meant to be split"
    ws = torch.empty(wbytes // 4, device=gpu)
    outs = []
    for wsp, wsb in ((ws, wbytes), (None, 0)):
        out = torch.full(ref.shape, float("nan"), device=gpu)
        _lib.check(lib.mvsdet_conv3d_k3_mfma_ws_f32(_lib.ptr(xd), _lib.ptr(wd), _lib.ptr(sc), _lib.ptr(sh), _lib.ptr(rd), _lib.ptr(out),
                                                    _lib.ptr(wsp), wsb, N, Cin, Cout, *grid, stride, 1, _lib.current_stream(gpu)), "conv")
        torch.cuda.synchronize()
        outs.append(out.cpu())
    tol = 1e-4 * max(1.0, float(ref.abs().max()))
    np.testing.assert_allclose(outs[0].numpy(), ref.numpy(), rtol=0, atol=tol)
    np.testing.assert_allclose(outs[1].numpy(), ref.numpy(), rtol=0, atol=tol)
    np.testing.assert_allclose(outs[0].numpy(), outs[1].numpy(), rtol=0, atol=tol)
    # the operator takes the split route by itself
    got = ops.conv3d_k3_mfma(xd, wd, sc, sh, True, stride, rd).cpu()
    assert torch.equal(got, outs[0])


@pytest.mark.gpu
@pytest.mark.parametrize("N,Cin,Cout,grid", [(1, 256, 512, (40, 40, 16)), (2, 512, 1024, (20, 20, 8)), (3, 64, 128, (6, 10, 4))])
def test_shortcut_gemm_kernel(gpu, N, Cin, Cout, grid):
    """csrc/neck_gemm.hip MODE 0 -- the 1x1x1 stride-2 shortcut with the BatchNorm folded in (imvoxel_neck.py:196-217): against the
    framework's Conv3d + BatchNorm3d in float64 (1e-5 of the output's scale: three-term split operands), for the shipped levels,
    a batch, and a voxel count that is no multiple of the 64-voxel block."""
    from mvsdet_amd import neck as NK
    from mvsdet_amd import ops
    ds = _randomise(NK._ConvModule(Cin, Cout, 1, 2, 0, act=False), 11).to(gpu)
    x = torch.randn((N, Cin) + grid, generator=torch.Generator().manual_seed(12)).to(gpu)
    with torch.no_grad():
        wq, bias = NK._gemm_weight(ds.conv, ds.bn, split=True)
        got = ops.conv3d_k1_s2_bf16x3(x, wq, bias, Cout)
        ref = ds.double()(x.double())
        # a split weight made for another layer, or not a split weight at all, is refused instead of read out of bounds
        with pytest.raises(ValueError):
            ops.conv3d_k1_s2_bf16x3(x, wq[: wq.numel() // 2].contiguous(), bias, Cout)
        with pytest.raises(TypeError):
            ops.conv3d_k1_s2_bf16x3(x, wq.float(), bias, Cout)
        with pytest.raises(ValueError):
            ops.convT3d_k2_s2_bf16x3(x, wq, bias.repeat(8)[: Cout], Cout, True)     # (8 Cout, Cin) rows asked of a (Cout, Cin) split
    assert got.shape == ref.shape
    err = float((got.double() - ref).abs().max())
    assert err <= 1e-5 * max(1.0, float(ref.abs().max())), err


@pytest.mark.gpu
@pytest.mark.parametrize("N,Cin,Cout,grid", [(1, 512, 256, (20, 20, 8)), (2, 1024, 512, (10, 10, 4)), (3, 64, 32, (3, 5, 2))])
def test_upsampling_gemm_kernel(gpu, N, Cin, Cout, grid):
    """csrc/neck_gemm.hip MODE 1 -- ConvTranspose3d(k=2, s=2) + BatchNorm + ReLU with the 2x2x2 interleave in the epilogue
    (imvoxel_neck.py:166-180): against the framework's layers in float64."""
    from mvsdet_amd import neck as NK
    from mvsdet_amd import ops
    up = _randomise(NK._UpBlock(Cin, Cout), 13).to(gpu)
    x = torch.randn((N, Cin) + grid, generator=torch.Generator().manual_seed(14)).to(gpu)
    with torch.no_grad():
        wq, bias = NK._gemm_weight(up[0], up[1], split=True)
        got = ops.convT3d_k2_s2_bf16x3(x, wq, bias, Cout, True)
        upd = up.double()
        ref = torch.relu(upd[1](upd[0](x.double())))
    assert got.shape == ref.shape
    err = float((got.double() - ref).abs().max())
    assert err <= 1e-5 * max(1.0, float(ref.abs().max())), err


@pytest.mark.gpu
def test_neck_on_a_batch_equals_the_scenes_one_by_one(gpu):
    """mvsdet.py:695-696 stacks the scenes' volumes: the shipped neck on (3,256,40,40,16) against each volume alone -- eval-mode
    BatchNorm is per-sample, the kernels' split decisions depend on the grid size, so equal up to summation order (1e-5 of the scale)."""
    from mvsdet_amd.neck import IndoorImVoxelNeck
    m = _randomise(IndoorImVoxelNeck(256, 128, [1, 1, 1]), 7).to(gpu)
    x = torch.randn((3, 256, 40, 40, 16), generator=torch.Generator().manual_seed(9)).to(gpu)
    with torch.no_grad():
        both = m(x)
        for i in range(3):
            one = m(x[i:i + 1])
            for a, b in zip(both, one):
                s = float(b.abs().max())
                assert float((a[i:i + 1] - b).abs().max()) <= 1e-5 * max(1.0, s)


@pytest.mark.gpu
@pytest.mark.parametrize("seed", range(10))
def test_gemm_kernels_randomised(gpu, seed):
    """csrc/neck_gemm.hip over random shapes: channel counts that are any multiple of 32 (K) / 128 rows, grids whose voxel count is
    no multiple of the 64-voxel block, batches; both modes against the framework's layers in float64, 1e-5 of the scale."""
    import random
    from mvsdet_amd import neck as NK
    from mvsdet_amd import ops
    rng = random.Random(1000 + seed)
    N = rng.randint(1, 3)
    cin = 32 * rng.randint(1, 12)
    grid = (rng.randint(1, 5), rng.randint(1, 7), rng.randint(1, 6))
    g = torch.Generator().manual_seed(seed)
    # mode 1: transposed k2 s2 (rows = 8 Cout: Cout any multiple of 16)
    cout = 16 * rng.randint(1, 10)
    up = _randomise(NK._UpBlock(cin, cout), 20 + seed).to(gpu)
    x = torch.randn((N, cin) + grid, generator=g).to(gpu)
    with torch.no_grad():
        wq, bias = NK._gemm_weight(up[0], up[1], split=True)
        got = ops.convT3d_k2_s2_bf16x3(x, wq, bias, cout, True)
        upd = up.double()
        ref = torch.relu(upd[1](upd[0](x.double())))
    assert float((got.double() - ref).abs().max()) <= 1e-5 * max(1.0, float(ref.abs().max())), ("convT", N, cin, cout, grid)
    # mode 0: 1x1x1 stride 2 (rows = Cout: a multiple of 128), even input grid
    cout0 = 128 * rng.randint(1, 3)
    ds = _randomise(NK._ConvModule(cin, cout0, 1, 2, 0, act=False), 40 + seed).to(gpu)
    x0 = torch.randn((N, cin, 2 * grid[0], 2 * grid[1], 2 * grid[2]), generator=g).to(gpu)
    with torch.no_grad():
        wq, bias = NK._gemm_weight(ds.conv, ds.bn, split=True)
        got = ops.conv3d_k1_s2_bf16x3(x0, wq, bias, cout0)
        ref = ds.double()(x0.double())
    assert float((got.double() - ref).abs().max()) <= 1e-5 * max(1.0, float(ref.abs().max())), ("k1s2", N, cin, cout0, grid)
