"""Per-launch timing of the InceptionV3 plan (after autotune): every op or group on its own, HIP events."""
import os, sys, ctypes as C
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), '..'))
import numpy as np, torch
from comic_amd import nets, _lib as L
B = int(os.environ.get('B', '64'))
X3 = os.environ.get('X3', '0') == '1'           # the bf16x3 plan (no rewrites)
plan = nets.CnnPlan(os.environ.get('NET', 'inception_v3'), (int(os.environ.get('IMG', '224')),) * 2,
                    pool_after_projection=os.environ.get('COMIC_POOL_REWRITE', '1') == '1',
                    fuse_pools=os.environ.get('FUSE', '1') == '1' and not X3, x3=X3)
enc = nets.CnnEncoder(plan, plan.init_params(0), B, 'bf16', 'cuda:0')
plan = enc.plan            # (small batches: the sibling plan without fused chains, CnnPlan.small_batch_plan)
if os.environ.get('COMIC_AUTOTUNE', '1') == '1':
    enc.autotune(cache=os.environ.get('COMIC_TUNE_CACHE'))
IMG = int(os.environ.get('IMG', '224'))
x = torch.rand(B, IMG, IMG, 3, device='cuda:0') * 2 - 1
for _ in range(3):
    enc.forward(x)
torch.cuda.synchronize()
st = L.stream_ptr()
n_ops = len(plan.ops)
grouped = enc._group_args is not None
n_rec = sum(1 for j in range(n_ops) if enc._ops[j].group > 0)
rec_bytes = enc.lib.comic_cnn_group_args_bytes(enc._ops, n_ops) // max(n_rec, 1)
rec = 0
i = 0
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
tot = 0.0
by_kind = {}
while i < n_ops:
    o = plan.ops[i]
    op = enc._ops[i]
    n = 1
    first = C.byref(enc._ops, i * C.sizeof(L.CnnOp))
    if grouped and op.group > 0:
        while i + n < n_ops and enc._ops[i + n].group == op.group:
            n += 1
        off = rec * rec_bytes
        def run():
            L.check(enc.lib.comic_cnn_forward_grouped(first, n, enc._bufptr, enc._bufch, enc._wt, B, 1,
                                                      enc._group_args.data_ptr() + off, st), 'g')
        rec += n
    else:
        def run():
            L.check(enc.lib.comic_cnn_forward(first, 1, enc._bufptr, enc._bufch, enc._wt, B, 1, st), 's')
    run(); run()
    R = 20
    e0.record()
    for _ in range(R):
        run()
    e1.record(); e1.synchronize()
    us = e0.elapsed_time(e1) / R * 1e3
    tot += us
    fl = 0
    desc = []
    for j in range(i, i + n):
        q = plan.ops[j]
        M = B * q['Ho'] * q['Wo']
        if q['kind'] < 2:
            fl += 2 * M * q['KH'] * q['KW'] * q['Cin'] * q['Cout']
        desc.append('%dx%d/%d %d->%d' % (q['KH'], q['KW'], q['SH'], q['Cin'], q['Cout']))
    kind = {0: 'conv', 1: 'stem', 2: 'max', 3: 'avg', 4: 'gap', 7: 'pbr'}.get(o['kind'], str(o['kind']))
    by_kind[kind] = by_kind.get(kind, 0) + us
    print('%3d %-4s x%d %3dx%-3d tile %2d %7.1f us %7.1f TF/s  %s' % (i, kind, n, o['Ho'], o['Wo'], op.tile, us,
                                                                    fl / us / 1e6, ' | '.join(desc)))
    i += n
print('sum of launches: %.1f us' % tot, by_kind)
