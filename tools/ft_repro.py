"""Run-to-run reproducibility of the full-size bf16 cnn_finetune step: REPS pairs of trainers from one state, 3 steps each;
prints per-step losses and the largest parameter difference.  LANES=0: no branch lanes (one chain + weight-gradient lane)."""
import os, sys
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), '..'))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), '..', 'tests'))
import numpy as np, torch
from comic_amd import nets, trainer, decoder as cdec
import test_gpu_path as T
from oracle import cnn_ref
B = 32
cnn_p = cnn_ref.randomize_bn(cnn_ref.init_params(0, 224), seed=1)
spec, cfg = T._spec_and_cfg(D=512, E=256, C=2048, Cg=2048)
p = T._rand_params(cfg, 4)
rng = np.random.default_rng(8)
x = torch.from_numpy(rng.uniform(-1, 1, (B, 224, 224, 3)).astype(np.float32)).cuda()
_, _, caps = T._batch(spec, B, 24, 9)
ref = None
POISON = os.environ.get('POISON')          # fill the allocator's free blocks with a pattern before every trainer
for rep in range(int(os.environ.get('REPS', '8'))):
    if POISON:
        junk = [torch.full((256 << 20,), int(POISON, 16), dtype=torch.int32, device='cuda:0') for _ in range(12)]   # 12 GiB
        torch.cuda.synchronize()
        del junk
    tr = trainer.CaptionTrainer(cnn_p, spec, p, B, (224, 224), 'bf16', 'cuda:0', lr_start=1e-3, lr_end=1e-3, max_step=10)
    tr.use_graph = False
    if os.environ.get('LANES', '1') == '0':
        tr.encoder.backward_branch_lanes = False
    if os.environ.get('WLANE', '1') == '0':
        tr.encoder.backward_lanes = False
    tr.enable_cnn_finetune()
    if os.environ.get('ACTF', '1') == '0':          # scheduled backward without the fused activation gradients
        import functools
        tr.encoder.backward = functools.partial(tr.encoder.backward, act_fusion=False)
    out = []
    NOISE = int(os.environ.get('NOISE', '0'))      # background launches on another stream around every step (timing perturbation)
    if NOISE and 'bg' not in globals():
        globals()['bg'] = torch.cuda.Stream()
        globals()['bga'] = torch.randn(2048, 2048, device='cuda:0')
    for s in range(3):
        if NOISE:
            with torch.cuda.stream(bg):
                for _ in range(NOISE):
                    bgb = bga @ bga
        r = tr.finetune_step(x, caps, training=False)
        torch.cuda.synchronize()
        out.append((float(r['loss']), tr.encoder.w_master.data.clone(), tr.encoder.beta.data.clone(), tr.decoder.params.data.clone()))
    if ref is None:
        ref = out
    print('rep %d (voided %d):' % (rep, tr.decoder.voided_steps()), ' | '.join('loss %.6f dW %.2e dbeta %.2e ddec %.2e' % (
        o[0], float((o[1] - q[1]).abs().max()), float((o[2] - q[2]).abs().max()), float((o[3] - q[3]).abs().max())) for o, q in zip(out, ref)))
    del tr
