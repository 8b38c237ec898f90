"""Fused branch chains (CnnPlan(fuse_chains=True): one launch per Mixed_6b-e block for its six 1x7 / 7x1 convs) against one
launch per conv depth, on ONE box: whole forward of B images from a hipGraph, and the launches of the four blocks alone.
   B=1280 python tools/chain_time.py"""
import os, sys, ctypes as C
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), '..'))
import torch
from comic_amd import nets, _lib as L
B = int(os.environ.get('B', '1280'))
res = {}
x = torch.rand(B, 224, 224, 3, device='cuda:0') * 2 - 1
base = None
for fuse in (False, True, False, True):
    plan = nets.CnnPlan('inception_v3', (224, 224), pool_after_projection=True, fuse_pools=True, fuse_chains=fuse)
    enc = nets.CnnEncoder(plan, plan.init_params(0), B, 'bf16', 'cuda:0', weights_from=base)
    base = base or enc
    enc.autotune(cache=os.environ.get('COMIC_TUNE_CACHE') or None)
    for _ in range(3):
        enc.forward(x, use_graph=True)
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    best = 1e9
    for _ in range(3):
        e0.record()
        for _ in range(10):
            enc.forward(x, use_graph=True)
        e1.record(); torch.cuda.synchronize()
        best = min(best, e0.elapsed_time(e1) / 10)
    # the 7-tap launches alone (eager): every group / op that holds a 7-tap conv
    st = L.stream_ptr()
    n_ops = len(plan.ops)
    rec_bytes = enc.lib.comic_cnn_group_args_bytes(enc._ops, n_ops) // max(1, sum(1 for j in range(n_ops) if enc._ops[j].group > 0))
    runs, rec, i = [], 0, 0
    while i < n_ops:
        o = plan.ops[i]
        n = 1
        if enc._ops[i].group > 0:
            while i + n < n_ops and enc._ops[i + n].group == enc._ops[i].group:
                n += 1
            if any(plan.ops[i + k]['kind'] == 0 and plan.ops[i + k]['KH'] * plan.ops[i + k]['KW'] == 7 and plan.ops[i + k]['H'] == 12
                   for k in range(n)):
                runs.append((i, n, rec * rec_bytes))
            rec += n
        i += n
    def seven():
        for (i, n, off) in runs:
            L.check(enc.lib.comic_cnn_forward_grouped(C.byref(enc._ops, i * C.sizeof(L.CnnOp)), n, enc._bufptr, enc._bufch, enc._wt,
                                                      B, 1, enc._group_args.data_ptr() + off, st), 'grouped')
    seven(); seven()
    b7 = 1e9
    for _ in range(3):
        e0.record()
        for _ in range(10):
            seven()
        e1.record(); torch.cuda.synchronize()
        b7 = min(b7, e0.elapsed_time(e1) / 10)
    print('fuse_chains=%d  B=%d  forward %.3f ms  (%.1f TFLOP/s, %.3f of 2.5 PF)   7-tap launches (%d) %.3f ms' % (
        fuse, B, best, B * enc.flops_per_image / best / 1e9, B * enc.flops_per_image / best / 1e9 / 2500, len(runs), b7), flush=True)
