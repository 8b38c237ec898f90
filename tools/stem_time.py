"""Time the streaming stem op (kind 8) alone at batch B (COMIC_STEM_DBG selects what is skipped)."""
import os, sys, ctypes as C
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), '..'))
import torch
from comic_amd import nets, _lib as L
B = int(os.environ.get('B', '640'))
plan = nets.CnnPlan('inception_v3', (224, 224), pool_after_projection=True, fuse_pools=True)
enc = nets.CnnEncoder(plan, plan.init_params(0), B, 'bf16', 'cuda:0')
x = torch.rand(B, 224, 224, 3, device='cuda:0') * 2 - 1
enc.forward(x); torch.cuda.synchronize()
i = [k for k, o in enumerate(plan.ops) if o['kind'] == 8][0]
first = C.byref(enc._ops, i * C.sizeof(L.CnnOp))
st = L.stream_ptr()
def run():
    L.check(enc.lib.comic_cnn_forward(first, 1, enc._bufptr, enc._bufch, enc._wt, B, 1, st), 's')
run(); run()
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
best = 1e9
for _ in range(3):
    e0.record()
    for _ in range(10):
        run()
    e1.record(); e1.synchronize()
    best = min(best, e0.elapsed_time(e1) / 10 * 1e3)
print('COMIC_STEM_DBG=%s  %.1f us' % (os.environ.get('COMIC_STEM_DBG', '0'), best))
