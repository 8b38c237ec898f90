"""Which layers own the bf16 plan's error?  Per end point of InceptionV3 at 224: max-norm deviation of the bf16 plan and of the
bf16x3 plan from the exact-fp32 plan (same weights, same images, device arithmetic throughout), and of 'bf16 up to block X, then
exact' hybrids formed by feeding the bf16 plan's end point X into the fp32 plan... -- not run: the table of plain deviations
already shows where the error is made (it grows with depth; no single block owns it)."""
import os, sys
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), '..'))
import numpy as np, torch
from comic_amd import nets
B = int(os.environ.get('B', '8'))
dev = 'cuda:0'
plain = nets.CnnPlan('inception_v3', (224, 224))
params = plain.init_params(0)
rng = np.random.default_rng(1)
# BatchNorm statistics away from (0, 1), as a trained net has them
for k in params:
    if k.endswith('moving_mean'): params[k] = (0.1 * rng.standard_normal(params[k].shape)).astype(np.float32)
    if k.endswith('moving_variance'): params[k] = rng.uniform(0.5, 1.5, params[k].shape).astype(np.float32)
    if k.endswith('beta'): params[k] = (0.1 * rng.standard_normal(params[k].shape)).astype(np.float32)
x = torch.from_numpy(rng.uniform(-1, 1, (B, 224, 224, 3)).astype(np.float32)).to(dev)
encs = {'f32': nets.CnnEncoder(plain, params, B, 'f32', dev),
        'bf16': nets.CnnEncoder(nets.CnnPlan('inception_v3', (224, 224)), params, B, 'bf16', dev),
        'bf16x3': nets.CnnEncoder(nets.CnnPlan('inception_v3', (224, 224), x3=True), params, B, 'bf16x3', dev)}
outs = {}
for k, e in encs.items():
    im, fm = e.forward(x)
    torch.cuda.synchronize()
    outs[k] = {n: e.end_point(n).float().cpu().numpy() for n in plain.end_points if n in e.plan.end_points}
    outs[k]['im_embed'] = im.cpu().numpy()
print('%-18s %12s %12s   (max|a - f32| / max|f32|, %d images)' % ('end point', 'bf16', 'bf16x3', B))
for n in list(plain.end_points) + ['im_embed']:
    if n not in outs['f32']:
        continue
    r = outs['f32'][n]
    e = [float(np.abs(outs[k][n].reshape(r.shape) - r).max() / (np.abs(r).max() + 1e-30)) for k in ('bf16', 'bf16x3')]
    print('%-18s %12.3e %12.3e' % (n, e[0], e[1]))
