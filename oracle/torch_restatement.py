"""PyTorch-CPU restatement of stage 1 of the reference path -- TEST INFRASTRUCTURE / CPU BASELINE ONLY.

The reference computes the cost volume with ATen operators (F.grid_sample, in-place adds, div/sub);
this file restates mvs_models/module.py:105-146 (homo_warping) and mvsdet.py:439-467 (variance block)
with the same operators, so that bench.py's `cpu_baseline` can time "what the reference does on the host
CPUs" (multi-threaded ATen kernels) next to the plain-C oracle.  Checked against the golden vectors in
tests/test_oracle_golden.py::test_torch_restatement.  Never imported by mvsdet_amd/.
"""
import torch
import torch.nn.functional as F


def homo_warp(src_fea, proj, depth_values):
    """src_fea (B,C,H,W); proj (B,4,4) = src_proj @ inverse(ref_proj) (module.py:116); depth_values (B,D)."""
    B, C, H, W = src_fea.shape
    D = depth_values.shape[1]
    with torch.no_grad():
        rot, trans = proj[:, :3, :3], proj[:, :3, 3:4]
        ys, xs = torch.meshgrid(torch.arange(H, dtype=torch.float32), torch.arange(W, dtype=torch.float32), indexing="ij")
        pix = torch.stack((xs.reshape(-1), ys.reshape(-1), torch.ones(H * W)))[None].repeat(B, 1, 1)
        ray = torch.matmul(rot, pix)                                            # module.py:126
        pts = ray.unsqueeze(2) * depth_values.view(B, 1, D, 1) + trans.view(B, 3, 1, 1)   # :128,135
        xy = pts[:, :2] / pts[:, 2:3]                                           # :136
        gx = xy[:, 0] / ((W - 1) / 2) - 1                                       # :137
        gy = xy[:, 1] / ((H - 1) / 2) - 1                                       # :138
        grid = torch.stack((gx, gy), dim=3).view(B, D * H, W, 2)
    out = F.grid_sample(src_fea, grid, mode="bilinear", padding_mode="zeros", align_corners=False)   # :142
    return out.view(B, C, D, H, W)


def plane_sweep_variance(feat, nbr, proj, depth_values, view_chunk=None):
    """feat (N,C,H,W); nbr (N,K) int64; proj (N,K,4,4); depth_values (N,D) -> (N,C,D,H,W).
    `view_chunk` bounds the live (chunk,C,D,H,W) temporaries (batch rows are independent)."""
    N, C, H, W = feat.shape
    K = nbr.shape[1]
    D = depth_values.shape[1]
    out = []
    step = view_chunk or N
    for s in range(0, N, step):
        e = min(N, s + step)
        ref = feat[s:e].unsqueeze(2).repeat(1, 1, D, 1, 1)       # mvsdet.py:439
        vsum, vsq = ref, ref ** 2                                # :441-442
        for j in range(K):
            warped = homo_warp(feat[nbr[s:e, j]], proj[s:e, j], depth_values[s:e])
            vsum += warped                                       # :463 (eval-mode in-place form)
            vsq += warped.pow_(2)                                # :464
            del warped
        out.append(vsq.div_(K + 1).sub_(vsum.div_(K + 1).pow_(2)))   # :467
    return torch.cat(out, 0) if len(out) > 1 else out[0]
