"""Is the decoder-mode step host-bound?  Time to ISSUE K steps (no sync) vs time until the GPU has finished them."""
import os, sys, time
sys.path.insert(0, '.')
import numpy as np, torch
import bench
from comic_amd import decoder as cdec, nets, trainer
dev = 'cuda:0'
plan = nets.CnnPlan('inception_v3', (224, 224), pool_after_projection=True)
tr = trainer.CaptionTrainer(plan.init_params(0), cdec.DecoderSpec(), None, 64, (224, 224), 'bf16', dev, seed=1, plan=plan)
tr.encoder.autotune()
rng = np.random.default_rng(0)
images = torch.from_numpy(rng.uniform(-1, 1, (64, 224, 224, 3)).astype(np.float32)).to(dev)
caps = [bench.synth_captions(rng, 64) for _ in range(4)]
if os.environ.get('INPLACE', '0') == '1':
    tr.encoder.bufs[plan.input].copy_(images); images = tr.encoder.bufs[plan.input]
G = os.environ.get('G', '0') == '1'
def step(i):
    im, fm = tr.encoder.forward(images, use_graph=True)
    tr.decoder.train_step(fm, im, caps[i % 4], training=True, use_graph=G)
    tr.opt.step(tr.decoder.grads, 1e-3)
for i in range(8): step(i)
torch.cuda.synchronize()
K = 30
t0 = time.perf_counter()
for i in range(K): step(i)
t1 = time.perf_counter()
torch.cuda.synchronize()
t2 = time.perf_counter()
print('graph_dec=%d: host issue %.3f ms/step, total %.3f ms/step (no overlap stream)' % (G, (t1 - t0) / K * 1e3, (t2 - t0) / K * 1e3))
# decoder only, fixed features
im, fm = tr.encoder.forward(images, use_graph=True)
torch.cuda.synchronize()
t0 = time.perf_counter()
for i in range(K):
    tr.decoder.train_step(fm, im, caps[i % 4], training=True, use_graph=G)
    tr.opt.step(tr.decoder.grads, 1e-3)
t1 = time.perf_counter()
torch.cuda.synchronize()
t2 = time.perf_counter()
print('decoder only: host issue %.3f ms/step, total %.3f ms/step' % ((t1 - t0) / K * 1e3, (t2 - t0) / K * 1e3))
