"""cnn_finetune backward alone (drained) under the lane modes of CnnEncoder: three lanes (two chain lanes + weight-gradient
lane), one chain lane + weight-gradient lane, everything on one stream.   B=32 python tools/ft_lane_modes.py"""
import os, sys, time
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), '..'))
import numpy as np, torch
import bench
from comic_amd import decoder as cdec, nets, trainer
dev = 'cuda:0'
Bf = int(os.environ.get('B', '32'))
rng = np.random.default_rng(0)
imgs = torch.from_numpy(rng.uniform(-1, 1, (Bf, 224, 224, 3)).astype(np.float32)).to(dev)
caps = bench.synth_captions(rng, Bf)
for lanes, branch in ((True, True), (True, False), (False, False)):
    plan = nets.CnnPlan('inception_v3', (224, 224))
    tr = trainer.CaptionTrainer(plan.init_params(0), cdec.DecoderSpec(), None, Bf, (224, 224), 'bf16', dev, seed=5, plan=plan)
    tr.encoder.backward_lanes, tr.encoder.backward_branch_lanes = lanes, branch
    tr.enable_cnn_finetune()
    tr.encoder.autotune(cache=os.environ.get('COMIC_TUNE_CACHE') or None)
    for _ in range(3):
        tr.finetune_step(imgs, caps)
    torch.cuda.synchronize()
    n = 10
    t0 = time.perf_counter()
    for _ in range(n):
        tr.finetune_step(imgs, caps)
    torch.cuda.synchronize()
    step = (time.perf_counter() - t0) / n * 1e3
    im_fm = tr.encoder.forward(imgs, use_graph=tr.use_graph)
    r = tr.decoder.train_step(im_fm[1], im_fm[0], np.asarray(caps), training=True, dp=tr.dp, use_graph=False, want_input_grads=True)
    ts = []
    for _ in range(6):
        torch.cuda.synchronize(); a = time.perf_counter()
        tr.encoder.backward(r['dfm'], r['dim_embed'])
        b = time.perf_counter(); torch.cuda.synchronize(); c = time.perf_counter()
        ts.append(((b - a) * 1e3, (c - a) * 1e3))
    print('wgrad lane %d, branch lanes %d: step %.3f ms; backward issue %.3f ms, drained %.3f ms' % (
        lanes, branch, step, min(t[0] for t in ts), min(t[1] for t in ts)), flush=True)
    del tr
