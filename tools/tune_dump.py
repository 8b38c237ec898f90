"""Every tile candidate of the forward autotune at B images (COMIC_TUNE_DUMP): which kernels lose, by how much."""
import os, sys
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), '..'))
os.environ['COMIC_TUNE_DUMP'] = '1'
import torch
from comic_amd import nets
B = int(os.environ.get('B', '64'))
plan = nets.CnnPlan('inception_v3', (224, 224), pool_after_projection=True, fuse_pools=True)
enc = nets.CnnEncoder(plan, plan.init_params(0), B, 'bf16', 'cuda:0')
enc.autotune(verbose=True)
