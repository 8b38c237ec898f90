import os, sys, ctypes as C
os.environ['COMIC_HIP_LIB'] = os.path.join(os.path.dirname(os.path.abspath(__file__)), '..', '..', 'comic-compact-image-captioning-with-attention_amd', 'lib', 'libcomic_hip_ASTAMPS.so')
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), '..', '..'))
import numpy as np, torch
from comic_amd import decoder as cdec
dev = 'cuda:0'
B, V, W = 50, 25599, 3
spec = cdec.DecoderSpec(V=V, H=1, fm_projection=None, token_type='word', start_id=V - 2, end_id=V - 1)
dec = cdec.Decoder(spec, None, dev, seed=3)
fm = torch.randn(B, spec.M, spec.C, device=dev); im = torch.randn(B, spec.Cg, device=dev)
for _ in range(3):
    dec.beam_search(fm, im, W, 30, want_attention=False, use_graph=False)
torch.cuda.synchronize()
buf = np.zeros(256 * 16 * 8, np.int64)
dec.lib.comic_debug_at_stamps(C.c_void_p(buf.ctypes.data))
st = buf.reshape(256, 16, 8)[:168].astype(np.float64)
live = st[:, 0, 5] > 0
st = st[live]
t0 = st[:, :, 0].min()
us = (st - t0) / 100.0
names = ['start', 'q', 'scored', 'sync1', 'softmax', 'end']
for w in (0, 1, 8, 15):
    print('attn wave', w, ' '.join('%s %.2f/%.2f' % (names[i], np.median(us[:, w, i]), us[:, w, i].max()) for i in range(6)))
print('WGs', live.sum())
