"""One grouped conv launch of the InceptionV3 plan at B images, on a forced tile, repeated: a target for rocprofv3 --pmc.
OP = index of the group's first op, TILES = comma list of tile ids (each runs N times), B, N."""
import os, sys, ctypes as C
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), '..'))
import numpy as np, torch
from comic_amd import nets, _lib as L
B = int(os.environ.get('B', '1280')); OP = int(os.environ.get('OP', '43')); N = int(os.environ.get('N', '6'))
tiles = [int(t) for t in os.environ.get('TILES', '44,58,59').split(',')]
plan = nets.CnnPlan('inception_v3', (224, 224), pool_after_projection=True, fuse_pools=True)
enc = nets.CnnEncoder(plan, plan.init_params(0), B, 'bf16', 'cuda:0')
plan = enc.plan            # (small batches: the sibling plan without fused chains, CnnPlan.small_batch_plan)
x = torch.rand(B, 224, 224, 3, device='cuda:0') * 2 - 1
enc.forward(x)
torch.cuda.synchronize()
st = L.stream_ptr()
n_ops = len(plan.ops)
n = 1
while OP + n < n_ops and enc._ops[OP + n].group == enc._ops[OP].group and enc._ops[OP].group > 0:
    n += 1
n_rec = sum(1 for j in range(n_ops) if enc._ops[j].group > 0)
rec_bytes = enc.lib.comic_cnn_group_args_bytes(enc._ops, n_ops) // max(n_rec, 1)
rec = sum(1 for j in range(OP) if enc._ops[j].group > 0)
first = C.byref(enc._ops, OP * C.sizeof(L.CnnOp))
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
for tile in tiles:
    enc._ops[OP].tile = tile
    enc._build_group_args()
    def run():
        L.check(enc.lib.comic_cnn_forward_grouped(first, n, enc._bufptr, enc._bufch, enc._wt, B, 1,
                                                  enc._group_args.data_ptr() + rec * rec_bytes, st), 'g')
    run(); run()
    e0.record()
    for _ in range(N):
        run()
    e1.record(); e1.synchronize()
    print('op %d x%d tile %d: %.1f us' % (OP, n, tile, e0.elapsed_time(e1) / N * 1e3))
