import sys; sys.path.insert(0,'/root/repo')
import torch
from mvsdet_amd import ops, synthetic
from mvsdet_amd.hotpath import MVSDetHotPath
gpu=torch.device('cuda:0')
N, C, D, hw = 40, 256, 12, (60, 80)
hp = MVSDetHotPath([40, 40, 16], [0.16, 0.16, 0.2], [0.2, 5.0], D)
meta = synthetic.make_img_meta(N, hw, seed=0)
feat = synthetic.make_features(N, C, hw, seed=0, device=gpu)
geo = hp.prepare_scene(meta, gpu)
ref = None
for it in range(30):
    v = ops.plane_sweep_variance(feat, geo.neighbor_ids, geo.proj_rel, geo.depth_values)
    if ref is None:
        ref = v.clone(); continue
    diff = (v != ref)
    nb = diff.sum().item()
    if nb:
        idx = diff.nonzero()
        print('iter', it, 'bad', nb, 'first', idx[0].tolist(), 'last', idx[-1].tolist(),
              'n', idx[:,0].unique().tolist()[:8], 'c%32', (idx[:,1]%32).unique().tolist()[:8], 'd', idx[:,2].unique().tolist(),
              'y', idx[:,3].unique().tolist()[:12], 'x', idx[:,4].unique().tolist()[:12])
print('done')
