"""fuse_pools / weight-stationary 1x1 plan against the plain forward-only plan: every shared end point and both
encoder outputs must agree bit for bit (same k order per accumulator, exact max)."""
import os, sys
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), '..'))
import torch
from comic_amd import nets
B = int(os.environ.get('B', '8'))
torch.manual_seed(0)
x = torch.rand(B, 224, 224, 3, device='cuda:0') * 2 - 1
pa = nets.CnnPlan('inception_v3', (224, 224), pool_after_projection=True)
pb = nets.CnnPlan('inception_v3', (224, 224), pool_after_projection=True, fuse_pools=True)
params = pa.init_params(0)
import numpy as np
rng = np.random.default_rng(1)
for k in params:
    if k.endswith('beta'):
        params[k] = rng.normal(0, 0.2, params[k].shape).astype(np.float32)
    if k.endswith('moving_mean'):
        params[k] = rng.normal(0, 0.1, params[k].shape).astype(np.float32)
ea = nets.CnnEncoder(pa, params, B, 'bf16', 'cuda:0')
eb = nets.CnnEncoder(pb, params, B, 'bf16', 'cuda:0')
# the plain plan with every conv on a fixed im2col tile (no weight-stationary kernel anywhere)
for i, o in enumerate(pa.ops):
    if o['kind'] == 0:
        ea._ops[i].tile = 3
ea._build_group_args()
ima, fma = ea.forward(x)
imb, fmb = eb.forward(x)
torch.cuda.synchronize()
bad = 0
for name in pb.end_points:
    if name in pa.end_points:
        a, b = ea.end_point(name), eb.end_point(name)
        same = torch.equal(a.view(torch.int16) if a.dtype == torch.bfloat16 else a, b.view(torch.int16) if b.dtype == torch.bfloat16 else b)
        if not same:
            d = (a.float() - b.float()).abs().max().item()
            print('%-16s DIFFERENT max abs %.3e' % (name, d)); bad += 1
print('end points compared:', len([n for n in pb.end_points if n in pa.end_points]), 'different:', bad)
print('fm equal', torch.equal(fma, fmb), 'im_embed equal', torch.equal(ima, imb), 'finite', bool(torch.isfinite(fmb).all()))
sys.exit(1 if bad or not torch.equal(fma, fmb) else 0)
