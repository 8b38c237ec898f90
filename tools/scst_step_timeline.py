"""Host timeline of one SCST step as bench.py's `scst_images_per_sec` runs it (rollouts of 29 steps, hypotheses cut at MS-COCO
lengths): where the host waits for the device and where the device waits for the host."""
import os, sys, time, types
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), '..'))
import numpy as np, torch
from comic_amd import decoder as cdec, nets, optim
from comic_amd.ops import id_to_caption, radix_ids_to_captions_and_ids, build_radix_wtoi
from comic_amd.scst.scorers import captionScorer
from comic_amd.scst import prepro_ngrams
device = 'cuda:0'
rng = np.random.default_rng(0)
Bs, W, IMG = 32, 7, 224
words = ['w%d' % i for i in range(10000)]
wtoi = {'<PAD>': -1}
for i, w in enumerate(words):
    wtoi[w] = i
for tok in ('<UNK>', '<GO>', '<EOS>'):
    wtoi[tok] = len(wtoi) - 1
cfg = types.SimpleNamespace(token_type='radix', radix_base=256, wtoi=wtoi, itow={str(v): k for k, v in wtoi.items()})
table = build_radix_wtoi(wtoi, 256)
refs = [[' '.join(rng.choice(words[:200], int(rng.integers(8, 15)))) for _ in range(5)] for _ in range(Bs)]
df = prepro_ngrams.build(['i%d,<GO> %s <EOS>' % (i, r) for i, rl in enumerate(refs) for r in rl])
scorer = captionScorer(df, dict(ciderD=1.0, bleu=[0, 0, 0, 2]))
plan = nets.CnnPlan('inception_v3', (IMG, IMG), pool_after_projection=True, fuse_pools=True)
enc = nets.CnnEncoder(plan, plan.init_params(0), Bs, 'bf16', device)
dec = cdec.Decoder(cdec.DecoderSpec(), None, device, seed=4)
dec.params.view('b_o')[257] = -30.0
opt = optim.AdamTF(dec.params)
imgs = torch.from_numpy(rng.uniform(-1, 1, (Bs, IMG, IMG, 3)).astype(np.float32)).to(device)
iters = 29
len_rng = np.random.default_rng(11)
ahead = {}
GRAPH = os.environ.get('GRAPH', '1') == '1'       # GRAPH=0: eager launches (visible to rocprofv3's kernel trace)

def cut(ids2d):
    ids2d = np.array(ids2d, copy=True)
    for r in range(ids2d.shape[0]):
        ids2d[r, 2 * int(len_rng.integers(8, 15)):] = 257
    return ids2d

def step(T):
    t = [time.perf_counter()]
    mark = lambda: t.append(time.perf_counter())
    if 'f' in ahead:
        im, fm = ahead.pop('f')
    else:
        im, fm = enc.forward(imgs, use_graph=GRAPH); im, fm = im.clone(), fm.clone()
    fb = dec.beam_search_ids(fm, im, W, iters, use_graph=GRAPH); fg = dec.greedy(fm, im, iters, defer=True, use_graph=GRAPH); mark()      # 1 enqueue rollouts
    beam = fb().transpose(2, 1, 0); mark()                                                                  # 2 wait beam
    caps, ids = radix_ids_to_captions_and_ids(cut(beam.reshape(-1, beam.shape[-1])), cfg, table); mark()     # 3 beam text + ids
    cap_beam = [[c] for c in caps]; mark()                                                                  # 4 (lists)
    imt, fmt = im.repeat(W, 1), fm.repeat(W, 1, 1)
    dec.train_step(fmt, imt, ids, training=True, use_graph=GRAPH, phase='fwd'); mark()                       # 5 enqueue fwd
    g = fg()[0]; mark()                                                                                     # 6 wait greedy
    cap_greedy = [[c] for c in id_to_caption(cut(g), cfg)]; mark()                                          # 7 greedy text
    a, b = enc.forward(imgs, use_graph=GRAPH); ahead['f'] = (a.clone(), b.clone()); mark()                   # 8 enqueue encoder
    hyp, ss, sg = scorer.get_hypo_scores(refs, cap_beam, cap_greedy); mark()                                # 9 score
    res = dec.train_step(None, None, ids, rewards=(ss - sg).astype(np.float32), training=True, use_graph=GRAPH, phase='bwd')
    opt.step(dec.grads, 1e-3); mark()                                                                       # 10 enqueue bwd
    torch.cuda.synchronize(); mark()                                                                        # 11 drain
    T.append(np.diff(t) * 1e3)

T = []
for _ in range(4):
    step([])
torch.cuda.synchronize()
for _ in range(8):
    step(T)
names = ['enqueue rollouts', 'wait beam', 'beam text + ids', 'lists', 'enqueue fwd', 'wait greedy', 'greedy text', 'enqueue encoder',
         'score', 'enqueue bwd+opt', 'drain']
m = np.mean(T, axis=0)
print('  '.join('%s %.2f' % (n, v) for n, v in zip(names, m)), '| step %.2f ms' % m.sum())
