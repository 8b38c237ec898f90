"""cnn_finetune step (configs[2]: CNN + decoder trainable, batch 32) alone: timing line + rocprofv3 target."""
import os, sys, time
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), '..'))
import numpy as np, torch
import bench
from comic_amd import decoder as cdec, nets, trainer
dev = 'cuda:0'
Bf = int(os.environ.get('B', '32'))
plan = nets.CnnPlan('inception_v3', (224, 224))
tr = trainer.CaptionTrainer(plan.init_params(0), cdec.DecoderSpec(), None, Bf, (224, 224), 'bf16', dev, seed=5, plan=plan)
tr.enable_cnn_finetune(autotune_backward=os.environ.get('COMIC_AUTOTUNE_BWD', '0') == '1', tune_cache=os.environ.get('COMIC_TUNE_CACHE') or None)
if os.environ.get('COMIC_AUTOTUNE', '1') == '1':
    tr.encoder.autotune(cache=os.environ.get('COMIC_TUNE_CACHE') or None)
rng = np.random.default_rng(0)
imgs = torch.from_numpy(rng.uniform(-1, 1, (Bf, 224, 224, 3)).astype(np.float32)).to(dev)
caps = bench.synth_captions(rng, Bf)
for _ in range(3):
    tr.finetune_step(imgs, caps)
torch.cuda.synchronize()
n = int(os.environ.get('N', '8'))
t0 = time.perf_counter()
for _ in range(n):
    res = tr.finetune_step(imgs, caps)
torch.cuda.synchronize()
dt = (time.perf_counter() - t0) / n
print('cnn_finetune step batch %d: %.3f ms = %.1f images/s (loss %.4f)' % (Bf, dt * 1e3, Bf / dt, float(res['loss'])))
