import sys, time; sys.path.insert(0, '.')
import numpy as np, torch
from comic_amd import decoder as cdec, nets
B, IMG, V = 50, 224, 25599
plan = nets.CnnPlan('inception_v3', (IMG, IMG), pool_after_projection=True)
enc = nets.CnnEncoder(plan, plan.init_params(0), B, 'bf16', 'cuda:0')
rng = np.random.default_rng(7)
imgs = torch.from_numpy(rng.uniform(-1, 1, (B, IMG, IMG, 3)).astype(np.float32)).to('cuda:0')
spec = cdec.DecoderSpec(V=V, H=1, fm_projection=None, token_type='word', start_id=V - 2, end_id=V - 1)
dec = cdec.Decoder(spec, None, 'cuda:0', seed=3)
im, fm = enc.forward(imgs)
for _ in range(2): r = dec.beam_search(fm, im, 3, 30, want_attention=False, use_graph=False)
torch.cuda.synchronize()
n, t0 = 5, time.perf_counter()
for _ in range(n): r = dec.beam_search(fm, im, 3, 30, want_attention=False, use_graph=False)
torch.cuda.synchronize(); dt = (time.perf_counter() - t0) / n
print('beam3 word decoder only (eager): %.2f ms per batch of %d (%d steps)' % (dt * 1e3, B, r['predicted_ids'].shape[0]))
