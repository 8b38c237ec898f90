import sys; sys.path.insert(0, '.')
import numpy as np, torch
from comic_amd import nets
from oracle import cnn_ref
from tests.gpu_util import DEV, dev, rel_err
from tests.test_gpu_path import _cnn_grads_device
B = 2
params = cnn_ref.randomize_bn(cnn_ref.init_params(0, 224), seed=1)
rng = np.random.default_rng(11)
x = rng.uniform(-1, 1, (B, 224, 224, 3)).astype(np.float32)
d_net = (1 + 0.5 * rng.standard_normal((B, 2048))).astype(np.float32)
d_fm = (1 + 0.5 * rng.standard_normal((B, 25, 2048))).astype(np.float32) / 25
for dtype in ('f32', 'bf16'):
    want, _, _ = cnn_ref.inception_v3_grads(params, x, d_net, d_fm, act_dtype=dtype)
    enc = nets.CnnEncoder(nets.CnnPlan('inception_v3', (224, 224)), params, B, dtype, DEV)
    enc.forward(dev(x))
    t = enc.backward(dev(d_fm), dev(d_net))
    torch.cuda.synchronize()
    got = _cnn_grads_device(enc, t)
    errs = sorted(((rel_err(got[k], want[k]), k) for k in want), reverse=True)
    print(dtype, 'worst:')
    for e, k in errs[:8]: print('   %.3e %s' % (e, k))
    print('   median %.3e' % errs[len(errs)//2][0])
    # l2-relative
    l2 = sorted(((float(np.linalg.norm(got[k]-want[k])/ (np.linalg.norm(want[k])+1e-30)), k) for k in want), reverse=True)
    print('   worst l2-rel: %.3e %s ; median %.3e' % (l2[0][0], l2[0][1], l2[len(l2)//2][0]))
    # repeat to see atomics nondeterminism
    t = enc.backward(dev(d_fm), dev(d_net)); torch.cuda.synchronize()
    got2 = _cnn_grads_device(enc, t)
    print('   run-to-run max rel diff %.3e' % max(rel_err(got2[k], got[k]) for k in got))
print('bf16 per-layer (plan order, last first):')
order = [w[0] for w in enc.plan.weights][::-1]
for pfx in order[:24]:
    print('   %-55s w %.3e  beta %.3e' % (pfx, rel_err(got[pfx + '/weights'], want[pfx + '/weights']), rel_err(got[pfx + '/BatchNorm/beta'], want[pfx + '/BatchNorm/beta'])))
