import os, sys, ctypes as C
os.environ['COMIC_HIP_LIB'] = os.path.join(os.path.dirname(os.path.abspath(__file__)), '..', '..', 'comic-compact-image-captioning-with-attention_amd', 'lib', 'libcomic_hip_STAMPS.so')
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
buf = np.zeros(256 * 8 * 8, np.int64)
rc = dec.lib.comic_debug_bl_stamps(C.c_void_p(buf.ctypes.data))
st = buf.reshape(256, 8, 8)[:200].astype(np.float64)
t0 = st[:, :, 0].min()
us = (st - t0) / 100.0          # 100 MHz
names = ['start', 'bar q0', 'bar q1', 'bar q2', 'bar q3', 'loop end', 'epi end']
for w in (0, 1, 2, 4, 7):
    print('wave', w, ' '.join('%s %.2f/%.2f' % (names[i], np.median(us[:, w, i]), us[:, w, i].max()) for i in range(7) if not (w >= 2 and False)))
print('last end over WGs: %.2f us; start spread %.2f' % (us[:, :, 5:7].max(), us[:, :, 0].max()))

buf2 = np.zeros(256 * 8 * 8, np.int64)
dec.lib.comic_debug_bm_stamps(C.c_void_p(buf2.ctypes.data))
st2 = buf2.reshape(256, 8, 8)[:50, :4].astype(np.float64)
t0 = st2[:, :, 0].min()
us2 = (st2 - t0) / 100.0
names2 = ['start', 'consts', 'fill', 'per-beam', 'final', 'end']
for w in (0, 1, 3):
    print('merge wave', w, ' '.join('%s %.2f/%.2f' % (names2[i], np.median(us2[:, w, i]), us2[:, w, i].max()) for i in range(6)))
