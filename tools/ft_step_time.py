"""cnn_finetune step (configs[2]: CNN + decoder trainable, batch 32) alone: timing line + rocprofv3 target."""
import os, sys, time
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), '..'))
import numpy as np, torch
import bench
from comic_amd import decoder as cdec, nets, trainer
dev = 'cuda:0'
Bf = int(os.environ.get('B', '32'))
X3 = os.environ.get('X3', '0') == '1'            # the bf16x3 plan (cnn_finetune at the fp32 parity bar)
plan = nets.CnnPlan('inception_v3', (224, 224), x3=X3)
DTYPE = 'bf16x3' if X3 else os.environ.get('DTYPE', 'bf16')       # DTYPE=f32: the exact-fp32 plan
tr = trainer.CaptionTrainer(plan.init_params(0), cdec.DecoderSpec(), None, Bf, (224, 224), DTYPE, dev, seed=5, plan=plan)
tr.enable_cnn_finetune(autotune_backward=os.environ.get('COMIC_AUTOTUNE_BWD', '0') == '1', tune_cache=os.environ.get('COMIC_TUNE_CACHE') or None)
if os.environ.get('COMIC_AUTOTUNE', '1') == '1' and DTYPE != 'f32':
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
t_issue = (time.perf_counter() - t0) / n          # host time to issue a step (the device runs behind)
torch.cuda.synchronize()
dt = (time.perf_counter() - t0) / n
print('host issue time per step: %.3f ms' % (t_issue * 1e3))
if os.environ.get('PHASES') == '1':               # host time of the phases of one step (each drained: not a step time)
    import time as _t
    def ph(name, fn):
        torch.cuda.synchronize(); a = _t.perf_counter(); r = fn(); b = _t.perf_counter(); torch.cuda.synchronize()
        print('  %-28s issue %.3f ms, drained %.3f ms' % (name, (b - a) * 1e3, (_t.perf_counter() - a) * 1e3)); return r
    im_fm = ph('encoder.forward', lambda: tr.encoder.forward(imgs, use_graph=tr.use_graph))
    r = ph('decoder.train_step', lambda: tr.decoder.train_step(im_fm[1], im_fm[0], np.asarray(caps), training=True, dp=tr.dp, use_graph=False, want_input_grads=True))
    t = ph('encoder.backward', lambda: tr.encoder.backward(r['dfm'], r['dim_embed']))
    ph('optimisers', lambda: (tr.opt.step(tr.decoder.grads, tr.lr()), tr.opt_cnn[0].step(t.dw, tr.lr()), tr.opt_cnn[1].step(t.dbeta, tr.lr())))
    ph('refresh + clear', lambda: (tr.encoder.refresh_weights(), tr.encoder.clear_grads_async()))
print('cnn_finetune step batch %d: %.3f ms = %.1f images/s (loss %.4f)' % (Bf, dt * 1e3, Bf / dt, float(res['loss'])))
