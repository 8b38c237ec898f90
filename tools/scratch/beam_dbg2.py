import os, sys
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), '..', '..'))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), '..', '..', 'tests'))
import numpy as np, torch
import test_gpu_path as T
from comic_amd import decoder as cdec
B, W, V, D = [int(x) for x in os.environ.get('CFG', '7,5,25599,512').split(',')]
spec, cfg = T._spec_and_cfg(fm_projection=None, H=1, token_type='word', V=V, D=D, init_method='project_hidden', start_id=V - 2, end_id=V - 1)
p = T._rand_params(cfg, 9)
fm, im, _ = T._batch(spec, B, 6, 23)
for eos in (1.5, 9.0, 9.0):
    pe = dict(p); pe['b_o'] = p['b_o'].copy(); pe['b_o'][spec.end_id] = eos
    os.environ['COMIC_BEAM_LOGITS'] = '0'; os.environ['COMIC_LSTM_STREAM'] = '0'
    ref = cdec.Decoder(spec, pe, 'cuda:0').beam_search(T.dev(fm), T.dev(im), W, 10, use_graph=False)
    del os.environ['COMIC_BEAM_LOGITS']; del os.environ['COMIC_LSTM_STREAM']
    dec = cdec.Decoder(spec, pe, 'cuda:0')
    for i in range(6):
        res = dec.beam_search(T.dev(fm), T.dev(im), W, 10, use_graph=(i % 2 == 1) or i > 3)
        ok = res['step_ids'].shape == ref['step_ids'].shape and (res['step_ids'] == ref['step_ids']).all()
        print('eos', eos, 'call', i, 'T', res['step_ids'].shape[0], 'ref T', ref['step_ids'].shape[0], 'ok', bool(ok), res['step_ids'][0, 0].tolist())
