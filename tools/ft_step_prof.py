"""One cnn_finetune step (InceptionV3 trainable + decoder) repeated: a target for rocprofv3 --kernel-trace."""
import os, sys, time
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), '..'))
import numpy as np, torch
from comic_amd import decoder as cdec, nets, trainer
B = int(os.environ.get('B', '32'))
plan = nets.CnnPlan('inception_v3', (224, 224))
tr = trainer.CaptionTrainer(plan.init_params(0), cdec.DecoderSpec(), None, B, (224, 224), 'bf16', 'cuda:0', seed=5, plan=plan)
tr.enable_cnn_finetune()
if os.environ.get('COMIC_AUTOTUNE', '1') == '1':
    tr.encoder.autotune(cache=os.environ.get('COMIC_TUNE_CACHE') or None)
rng = np.random.default_rng(1)
imgs = torch.from_numpy(rng.uniform(-1, 1, (B, 224, 224, 3)).astype(np.float32)).cuda()
L = 31
caps = np.full((B, L), -1, np.int64)
for b in range(B):
    n = int(rng.integers(16, 29)) if b else 28
    caps[b, 0] = 256; caps[b, 1:1 + n] = rng.integers(0, 256, n); caps[b, 1 + n] = 257
for _ in range(3):
    tr.finetune_step(imgs, caps)
torch.cuda.synchronize()
N = int(os.environ.get('N', '10'))
t0 = time.perf_counter()
for _ in range(N):
    tr.finetune_step(imgs, caps)
torch.cuda.synchronize()
print('finetune step ms', (time.perf_counter() - t0) / N * 1e3)
