import os, sys
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), '..', '..'))
import numpy as np, torch
from comic_amd import decoder as cdec
for M, C, Cg in ((196, 832, 1024), (64, 2048, 2048)):
    spec = cdec.DecoderSpec(M=M, C=C, Cg=Cg)
    a, b = cdec.Decoder(spec, None, 'cuda:0', seed=3), cdec.Decoder(spec, None, 'cuda:0', seed=3)
    rng = np.random.default_rng(1)
    B = 64
    fm = torch.from_numpy(rng.standard_normal((B, M, C)).astype(np.float32)).cuda()
    im = torch.from_numpy(rng.standard_normal((B, Cg)).astype(np.float32)).cuda()
    caps = np.full((B, 20), -1, np.int64)
    for r in range(B):
        n = int(rng.integers(6, 18)); caps[r, 0] = 256; caps[r, 1:1 + n] = rng.integers(0, 256, n); caps[r, 1 + n] = 257
    for it in range(4):
        ra = a.train_step(fm, im, caps, training=True, seed=50 + it, use_graph=False)
        rb = b.train_step(fm, im, caps, training=True, seed=50 + it, use_graph=True)
        torch.cuda.synchronize()
        assert a.lib.comic_decoder_train_path() == 3
        assert float(ra['loss']) == float(rb['loss']), (M, it, float(ra['loss']), float(rb['loss']))
        assert torch.equal(a.grads.data, b.grads.data), (M, it)
    print('M', M, 'graph replay == eager, path 3, loss', float(ra['loss']))
