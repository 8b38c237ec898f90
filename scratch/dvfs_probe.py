"""Does the decoder-only step time depend on what the GPU did just before (clock governor)?"""
import os, sys, time
sys.path.insert(0, '.')
import numpy as np, torch
import bench
from comic_amd import decoder as cdec, nets, trainer
dev = 'cuda:0'
plan = nets.CnnPlan('inception_v3', (224, 224), pool_after_projection=True)
tr = trainer.CaptionTrainer(plan.init_params(0), cdec.DecoderSpec(), None, 64, (224, 224), 'bf16', dev, seed=1, plan=plan)
rng = np.random.default_rng(0)
images = torch.from_numpy(rng.uniform(-1, 1, (64, 224, 224, 3)).astype(np.float32)).to(dev)
caps = [bench.synth_captions(rng, 64) for _ in range(4)]
im, fm = tr.encoder.forward(images)
def dec_loop(K):
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for i in range(K):
        tr.decoder.train_step(fm, im, caps[i % 4], training=True)
        tr.opt.step(tr.decoder.grads, 1e-3)
    torch.cuda.synchronize(); return (time.perf_counter() - t0) / K * 1e3
def cnn_loop(K):
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for i in range(K): tr.encoder.forward(images, use_graph=True)
    torch.cuda.synchronize(); return (time.perf_counter() - t0) / K * 1e3
dec_loop(5); cnn_loop(5)
print('decoder 30 steps            : %.3f ms' % dec_loop(30))
print('decoder 30 steps (again)    : %.3f ms' % dec_loop(30))
time.sleep(3)
print('after 3 s idle, decoder 30  : %.3f ms' % dec_loop(30))
print('decoder 300 steps           : %.3f ms' % dec_loop(300))
print('cnn 200 forwards            : %.3f ms' % cnn_loop(200))
print('right after cnn, decoder 30 : %.3f ms' % dec_loop(30))
for r in range(5):
    print('decoder 30 steps #%d         : %.3f ms' % (r, dec_loop(30)))
