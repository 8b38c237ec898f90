"""Experiment: beam-3 captions/s of bench.extras with N decode loops in flight (COMIC_INFER_IN_FLIGHT)."""
import os, sys
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), '..'))
os.environ['COMIC_EXTRAS_ONLY'] = 'beam'
import numpy as np, torch
import bench
from comic_amd import nets
plan = nets.CnnPlan('inception_v3', (224, 224), pool_after_projection=True, fuse_pools=True)
params = plan.init_params(0)
enc = nets.CnnEncoder(plan, params, 64, 'bf16', 'cuda:0')
out = bench.extras('cuda:0', enc, params, plan)
print('lanes', os.environ.get('COMIC_INFER_IN_FLIGHT', '3'), 'beam3', out['beam3_captions_per_sec'])
