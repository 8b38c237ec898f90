"""Decoder training step alone on random features, any geometry (rocprofv3 --kernel-trace --stats target).
   M=196 C=832 CG=1024 B=64 python tools/dec_step_time.py"""
import os, sys
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), '..'))
import numpy as np, torch
import bench
from comic_amd import decoder as cdec, optim
M, C, CG, B = (int(os.environ.get(k, d)) for k, d in (('M', '196'), ('C', '832'), ('CG', '1024'), ('B', '64')))
dev = 'cuda:0'

spec = cdec.DecoderSpec(M=M, C=C, Cg=CG)
dec = cdec.Decoder(spec, None, dev, seed=1)
opt = optim.AdamTF(dec.params)
rng = np.random.default_rng(0)
fm = torch.randn(B, M, C, device=dev)
im = torch.randn(B, CG, device=dev)
caps = [bench.synth_captions(rng, B) for _ in range(4)]
g = os.environ.get('GRAPH', '0') == '1'
for i in range(4):
    dec.train_step(fm, im, caps[i % 4], training=True, use_graph=g); opt.step(dec.grads, 1e-3)
torch.cuda.synchronize()
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
n = int(os.environ.get('N', '12'))
e0.record()
for i in range(n):
    r = dec.train_step(fm, im, caps[i % 4], training=True, use_graph=g); opt.step(dec.grads, 1e-3)
e1.record(); e1.synchronize()
print('decoder step M=%d C=%d B=%d: %.3f ms  (T\' %d, path %d, loss %.4f)' % (M, C, B, e0.elapsed_time(e1) / n, int(r['Tp']),
      dec.lib.comic_decoder_train_path(), float(r['loss'])))
