import sys, time; sys.path.insert(0, '.')
import numpy as np, torch
from comic_amd import decoder as cdec, nets, trainer
import bench
B = int(sys.argv[1]) if len(sys.argv) > 1 else 64
plan = nets.CnnPlan('inception_v3', (224, 224))
tr = trainer.CaptionTrainer(plan.init_params(0), cdec.DecoderSpec(), None, B, (224, 224), 'bf16', 'cuda:0', plan=plan, seed=1)
tr.enable_cnn_finetune()
rng = np.random.default_rng(0)
x = torch.rand(B, 224, 224, 3, device='cuda:0') * 2 - 1
caps = bench.synth_captions(rng, B)
for _ in range(3): res = tr.finetune_step(x, caps)
torch.cuda.synchronize()
t0 = time.perf_counter(); n = 10
for _ in range(n): res = tr.finetune_step(x, caps)
torch.cuda.synchronize(); dt = (time.perf_counter() - t0) / n
print('finetune B=%d: %.2f ms/step, %.0f img/s, loss %.4f' % (B, dt * 1e3, B / dt, float(res['loss'])))
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
e0.record()
for _ in range(5): tr.encoder.backward(res['dfm'], res['dim_embed'])
e1.record(); torch.cuda.synchronize()
print('cnn backward alone: %.2f ms' % (e0.elapsed_time(e1) / 5))
