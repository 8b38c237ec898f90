import os, sys
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), '..', '..'))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), '..', '..', 'tests'))
import numpy as np, torch
import test_gpu_path as T
from comic_amd import decoder as cdec
B, W, V, D = 50, 3, 8962, 128
spec, cfg = T._spec_and_cfg(fm_projection=None, H=1, token_type='word', V=V, D=D, init_method='project_hidden', start_id=V - 2, end_id=V - 1)
p = T._rand_params(cfg, 9)
fm, im, _ = T._batch(spec, B, 6, 23)
pe = dict(p); pe['b_o'] = p['b_o'].copy(); pe['b_o'][spec.end_id] = 9.0
for env in ('1', '0'):
    os.environ['COMIC_BEAM_LOGITS'] = env
    dec = cdec.Decoder(spec, pe, 'cuda:0')
    for i in range(4):
        res = dec.beam_search(T.dev(fm), T.dev(im), W, 10, use_graph=(i > 0))
        print('  call', i, 'T', res['step_ids'].shape[0])
    ctx = list(dec._infer_cache.values())[0] if hasattr(dec, '_infer_cache') else None
    print(env, 'path', dec.lib.comic_decoder_beam_path(), 'T', res['step_ids'].shape[0], 'lengths', res['lengths'][:4].tolist())
    print(' words step0', res['step_ids'][0, :3].tolist(), 'step1', res['step_ids'][1, :3].tolist())
    fin = [(res['step_ids'][t] == spec.end_id).all() for t in range(res['step_ids'].shape[0])]
    print(' all-eos per step', fin)
