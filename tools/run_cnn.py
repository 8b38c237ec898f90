"""The InceptionV3 forward alone (bf16, forward-only plan, autotuned): timing line + a target for rocprofv3 --pmc."""
import os, sys
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), '..'))
import numpy as np, torch
from comic_amd import nets
B = int(os.environ.get('B', '64'))
plan = nets.CnnPlan(os.environ.get('NET', 'inception_v3'), (224, 224), group_branches=os.environ.get('COMIC_CNN_GROUP', '1') == '1',
                    pool_after_projection=os.environ.get('COMIC_POOL_REWRITE', '1') == '1',
                    fuse_pools=os.environ.get('COMIC_POOL_REWRITE', '1') == '1' and os.environ.get('COMIC_FUSE_POOLS', '1') == '1',
                    x3=os.environ.get('X3', '0') == '1', fuse_stem_1a=os.environ.get('COMIC_FUSE_1A', '1') == '1')
enc = nets.CnnEncoder(plan, plan.init_params(0), B, 'bf16', 'cuda:0')
plan = enc.plan            # (small batches: the sibling plan without fused chains, CnnPlan.small_batch_plan)
if os.environ.get('COMIC_AUTOTUNE', '1') == '1':
    enc.autotune(cache=os.environ.get('COMIC_TUNE_CACHE') or None)
x = torch.rand(B, 224, 224, 3, device='cuda:0') * 2 - 1
for _ in range(3):
    enc.forward(x)
torch.cuda.synchronize()
g = os.environ.get('GRAPH', '0') == '1'
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
for _ in range(3): enc.forward(x, use_graph=g)
e0.record()
for _ in range(20): enc.forward(x, use_graph=g)
e1.record(); torch.cuda.synchronize()
print('cnn forward ms', e0.elapsed_time(e1) / 20, 'graph', g)
print('tiles', [enc._ops[i].tile for i in range(len(plan.ops)) if plan.ops[i]['kind'] == 0])
