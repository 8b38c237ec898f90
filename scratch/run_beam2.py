import sys, time; sys.path.insert(0, '.')
import numpy as np, torch
from comic_amd import decoder as cdec
B = 50
rng = np.random.default_rng(7)
fm = torch.from_numpy(rng.standard_normal((B, 25, 2048)).astype(np.float32)).cuda()
im = torch.from_numpy(rng.standard_normal((B, 2048)).astype(np.float32)).cuda()
dec2 = cdec.Decoder(cdec.DecoderSpec(), None, 'cuda:0', seed=3)
for _ in range(3): r = dec2.beam_search(fm, im, 3, 60, want_attention=False)
torch.cuda.synchronize(); t0 = time.perf_counter()
for _ in range(5): r = dec2.beam_search(fm, im, 3, 60, want_attention=False)
torch.cuda.synchronize(); print('beam3 COMIC-256: %.2f ms' % ((time.perf_counter() - t0) / 5 * 1e3))
