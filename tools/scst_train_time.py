"""Decoder training step alone at SCST sizes (B = 224 hypotheses of 32 images x (1 greedy + 6 beams ...)) vs batch 64."""
import os, sys, time
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), '..'))
import numpy as np, torch
from comic_amd import decoder as cdec
import bench
dev = 'cuda:0'
rng = np.random.default_rng(0)
for B in (64, 224):
    spec = cdec.DecoderSpec()
    dec = cdec.Decoder(spec, None, dev, seed=1)
    fm = torch.randn(B, spec.M, spec.C, device=dev)
    im = torch.randn(B, spec.Cg, device=dev)
    caps = bench.synth_captions(rng, B)
    rewards = rng.standard_normal(B).astype(np.float32)
    for _ in range(3):
        r = dec.train_step(fm, im, caps, rewards=rewards, training=True)
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(10):
        r = dec.train_step(fm, im, caps, rewards=rewards, training=True)
    e1.record(); e1.synchronize()
    print('B %d  Tp %d  train step %.3f ms  path %d' % (B, r['Tp'], e0.elapsed_time(e1) / 10, dec.lib.comic_decoder_train_path()))
