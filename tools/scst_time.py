"""Segments of the SCST step of bench.py's extras (batch 32, greedy + beam-7, 40 steps max): encoder, greedy, beam, host
text + reward, reward-weighted training step, Adam -- each synchronised (so the sum exceeds the pipelined step)."""
import os, sys, time, types
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), '..'))
import numpy as np, torch
import bench
from comic_amd import decoder as cdec, nets, optim
from comic_amd.ops import id_to_caption, captions_to_batched_ids, build_radix_wtoi
from comic_amd.scst.scorers import captionScorer
from comic_amd.scst import prepro_ngrams
from oracle import cnn_ref
device = 'cuda:0'
rng = np.random.default_rng(7)
IMG = 224
plan = nets.CnnPlan('inception_v3', (IMG, IMG), pool_after_projection=True, fuse_pools=True)
cnn_params = plan.init_params(0)
Bs, W = 32, 7
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
spec = cdec.DecoderSpec()
dec = cdec.Decoder(spec, None, device, seed=4)
dec.params.view('b_o')[257] = float(os.environ.get('EOS_BIAS', '2.0'))
opt = optim.AdamTF(dec.params)
enc_s = nets.CnnEncoder(plan, cnn_params, Bs, 'bf16', device)
enc_s.autotune()
imgs = torch.from_numpy(rng.uniform(-1, 1, (Bs, IMG, IMG, 3)).astype(np.float32)).to(device)
iters = 40
seg = {}
def tick(name, t0):
    torch.cuda.synchronize()
    seg[name] = seg.get(name, 0.0) + time.perf_counter() - t0
    return time.perf_counter()
def step(timed):
    t = time.perf_counter()
    im, fm = enc_s.forward(imgs, use_graph=True)
    if timed: t = tick('encoder', t)
    greedy, _, _ = dec.greedy(fm, im, iters)
    if timed: t = tick('greedy', t)
    beam = dec.beam_search(fm, im, W, iters, want_attention=False)['predicted_ids'].transpose(2, 1, 0)
    if timed: t = tick('beam7', t)
    cap_beam = [[c] for c in id_to_caption(beam.reshape(-1, beam.shape[-1]), cfg)]
    cap_greedy = [[c] for c in id_to_caption(greedy, cfg)]
    if timed: t = tick('id_to_caption', t)
    hypos, sc_s, sc_g = scorer.get_hypo_scores(refs, cap_beam, cap_greedy)
    if timed: t = tick('reward', t)
    ids = captions_to_batched_ids(hypos, cfg, table)
    if timed: t = tick('to_ids', t)
    im2, fm2 = im.repeat(W, 1), fm.repeat(W, 1, 1)
    res = dec.train_step(fm2, im2, ids, rewards=(sc_s - sc_g).astype(np.float32), training=True, use_graph=True)
    if timed: t = tick('train_step(Tp=%d,path=%d)' % (res['Tp'], dec.lib.comic_decoder_train_path()), t)
    opt.step(dec.grads, 1e-3)
    if timed: t = tick('adam', t)
for _ in range(3): step(False)
torch.cuda.synchronize()
n = 10
t0 = time.perf_counter()
for _ in range(n): step(False)
torch.cuda.synchronize()
print('step %.3f ms (unsegmented)' % ((time.perf_counter() - t0) / n * 1e3))
for _ in range(n): step(True)
print({k: round(v / n * 1e3, 3) for k, v in seg.items()}, 'sum %.3f' % (sum(seg.values()) / n * 1e3))
