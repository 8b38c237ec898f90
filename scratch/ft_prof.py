"""cnn_finetune step alone (batch 32), for rocprofv3 --kernel-trace --stats."""
import sys
sys.path.insert(0, '.')
import numpy as np, torch
import bench
from comic_amd import decoder as cdec, nets, trainer
rng = np.random.default_rng(7)
Bf = 32
plan_ft = nets.CnnPlan('inception_v3', (224, 224))
tr = trainer.CaptionTrainer(plan_ft.init_params(seed=0), cdec.DecoderSpec(), None, Bf, (224, 224), 'bf16', 'cuda:0', seed=5, plan=plan_ft)
tr.enable_cnn_finetune()
imgs = torch.from_numpy(rng.uniform(-1, 1, (Bf, 224, 224, 3)).astype(np.float32)).to('cuda:0')
caps = bench.synth_captions(rng, Bf)
for _ in range(3):
    tr.finetune_step(imgs, caps)
torch.cuda.synchronize()
import time
n, t0 = 10, time.perf_counter()
for _ in range(n):
    tr.finetune_step(imgs, caps)
torch.cuda.synchronize()
print('finetune images/s', Bf * n / (time.perf_counter() - t0))
