"""Time split of one SCST step (bench extras configuration) with synchronising timers."""
import sys, time, types
sys.path.insert(0, '.')
import numpy as np, torch
import bench
from comic_amd import decoder as cdec, nets, optim
from comic_amd.ops import id_to_caption, captions_to_batched_ids, build_radix_wtoi
from comic_amd.scst.scorers import captionScorer
from comic_amd.scst import prepro_ngrams
dev = 'cuda:0'; IMG = 224
plan = nets.CnnPlan('inception_v3', (IMG, IMG), pool_after_projection=True)
rng = np.random.default_rng(7)
Bs, W = 32, 7
words = ['w%d' % i for i in range(10000)]
wtoi = {'<PAD>': -1}
for i, w in enumerate(words): wtoi[w] = i
for tok in ('<UNK>', '<GO>', '<EOS>'): wtoi[tok] = len(wtoi) - 1
cfg = types.SimpleNamespace(token_type='radix', radix_base=256, wtoi=wtoi, itow={str(v): k for k, v in wtoi.items()})
table = build_radix_wtoi(wtoi, 256)
refs = [[' '.join(rng.choice(words[:200], int(rng.integers(8, 15)))) for _ in range(5)] for _ in range(Bs)]
df = prepro_ngrams.build(['i%d,<GO> %s <EOS>' % (i, r) for i, rl in enumerate(refs) for r in rl])
scorer = captionScorer(df, dict(ciderD=1.0, bleu=[0, 0, 0, 2]))
dec = cdec.Decoder(cdec.DecoderSpec(), None, dev, seed=4)
dec.params.view('b_o')[257] = 2.0
opt = optim.AdamTF(dec.params)
enc = nets.CnnEncoder(plan, plan.init_params(0), Bs, 'bf16', dev); enc.autotune()
imgs = torch.from_numpy(rng.uniform(-1, 1, (Bs, IMG, IMG, 3)).astype(np.float32)).to(dev)
iters = 40
T = {}
def tick(name, t0):
    torch.cuda.synchronize(); T[name] = T.get(name, 0) + time.perf_counter() - t0; return time.perf_counter()
def step(measure):
    t = time.perf_counter()
    im, fm = enc.forward(imgs, use_graph=True)
    if measure: t = tick('cnn', t)
    greedy, _, _ = dec.greedy(fm, im, iters)
    if measure: t = tick('greedy', t)
    beam = dec.beam_search(fm, im, W, iters, want_attention=False)['predicted_ids'].transpose(2, 1, 0)
    if measure: t = tick('beam7', t)
    cap_beam = [[c] for c in id_to_caption(beam.reshape(-1, beam.shape[-1]), cfg)]
    cap_greedy = [[c] for c in id_to_caption(greedy, cfg)]
    if measure: t = tick('id_to_caption', t)
    hypos, sc_s, sc_g = scorer.get_hypo_scores(refs, cap_beam, cap_greedy)
    if measure: t = tick('scorer', t)
    ids = captions_to_batched_ids(hypos, cfg, table)
    if measure: t = tick('to_ids', t)
    im2, fm2 = im.repeat(W, 1), fm.repeat(W, 1, 1)
    res = dec.train_step(fm2, im2, ids, rewards=(sc_s - sc_g).astype(np.float32), training=True, use_graph=True)
    opt.step(dec.grads, 1e-3)
    if measure: t = tick('train_step(224 rows, T=%d)' % ids.shape[1], t)
for _ in range(3): step(False)
n = 5
for _ in range(n): step(True)
tot = sum(T.values())
for k, v in T.items(): print('%-32s %7.2f ms  %5.1f%%' % (k, v / n * 1e3, 100 * v / tot))
print('total %.2f ms -> %.0f images/s' % (tot / n * 1e3, Bs * n / tot))
