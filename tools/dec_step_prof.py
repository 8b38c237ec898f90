"""Decoder training step alone at a chosen geometry (default: Inception-V1 Mixed_4f, M = 196): a target for rocprofv3 --kernel-trace."""
import os, sys, time
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), '..'))
import numpy as np, torch
from comic_amd import decoder as cdec
M, C, Cg, B = int(os.environ.get('M', '196')), int(os.environ.get('C', '832')), int(os.environ.get('CG', '1024')), int(os.environ.get('B', '64'))
spec = cdec.DecoderSpec(M=M, C=C, Cg=Cg)
dec = cdec.Decoder(spec, None, 'cuda:0', seed=3)
rng = np.random.default_rng(1)
fm = torch.from_numpy(rng.standard_normal((B, M, C)).astype(np.float32)).cuda()
im = torch.from_numpy(rng.standard_normal((B, Cg)).astype(np.float32)).cuda()
L = 31
caps = np.full((B, L), -1, np.int64)
for b in range(B):
    n = int(rng.integers(16, 29)) if b else 28
    caps[b, 0] = spec.start_id; caps[b, 1:1 + n] = rng.integers(0, 256, n); caps[b, 1 + n] = spec.end_id
for _ in range(3):
    dec.train_step(fm, im, caps, training=True)
torch.cuda.synchronize()
t0 = time.perf_counter()
N = int(os.environ.get('N', '10'))
for _ in range(N):
    dec.train_step(fm, im, caps, training=True)
torch.cuda.synchronize()
print('decoder step ms', (time.perf_counter() - t0) / N * 1e3, 'path', dec.lib.comic_decoder_train_path())
