#!/usr/bin/env python3
"""bench.py -- images/sec of decoder-mode XE training (BASELINE.json configs[1]):
COMIC-256 (radix-256 tokens, 8 heads, tied projection) on InceptionV3 (frozen, bf16 conv
MFMA path), batch 64 per GPU, synthetic 224x224x3 inputs resident in HBM.

One "step" = InceptionV3 forward + keys projection + teacher-forced decoder forward and
backward + (N>1: RCCL all-reduce of the 22.8 MB flat gradient) + fused TF-Adam update.
Prints ONE JSON line on rank 0 (contract in the task statement).
"""
import argparse
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)
os.environ.setdefault('GPU_MAX_HW_QUEUES', '8')      # before the HIP runtime initialises: see comic_amd/__init__.py

BATCH = 64
IMG = 224
FLOP_PER_IMAGE_CNN = 2 * 2835873120          # 94 convs @224 (SURVEY Appendix B / BASELINE.md §2)
PEAK_BF16_MFMA = 2.5e15                       # dense bf16, MI355X_MICROARCH.md chip table
DEFAULT_ENC_GROUP = 30                        # steps per encoder forward in the pipelined frozen-CNN step (measured 35.7k / 35.9k / 36.2k images/s at 10 / 15 / 30)
GRAPH_CNN = os.environ.get('COMIC_GRAPH_CNN', '1') == '1'   # hipGraph replay of the CNN plan
EVENTS = os.environ.get('COMIC_NO_EVENTS', '0') != '1'
STEP_TIMES = [] if os.environ.get('COMIC_STEP_TIMES', '0') == '1' else None      # diagnostic: per-step event / host stamps
GRAPH_DEC = os.environ.get('COMIC_GRAPH_DEC', '1') == '1'   # hipGraph replay of the decoder step (round 2, persistent loops: 1.78 vs 1.80 ms eager; round 1's per-step launches were faster eager)


def synth_captions(rng, B):
    """N ~ U{8..14} words, ids ~ U{0..9999}, radix-256 -> [256, d1 d0 ..., 257], PAD -1."""
    rows = []
    for b in range(B):
        n = 14 if b == 0 else int(rng.integers(8, 15))   # row 0 pins the bucket length (one graph shape)
        ids = rng.integers(0, 10000, n)
        r = [256]
        for w in ids:
            r += [int(w) // 256, int(w) % 256]
        r.append(257)
        rows.append(r)
    L = max(len(r) for r in rows)
    out = np.full((B, L), -1, np.int64)
    for i, r in enumerate(rows):
        out[i, :len(r)] = r
    return out


def extras(device, enc, cnn_params, plan):
    """Secondary figures of BASELINE.json's metric line (not `value`): beam-3 captions/sec on the
    InstaPIC-style word baseline (configs[4]: word tokens, V=25 599, 1 head, no projection, batch 50)
    and SCST images/sec (configs[3]: greedy + beam-7 rollouts, C++ CIDEr-D/BLEU reward, reward-weighted
    step, batch 32).  Synthetic inputs; single GPU."""
    import torch
    from comic_amd import decoder as cdec, nets, optim, streams
    from comic_amd.ops import id_to_caption, radix_ids_to_captions_and_ids, build_radix_wtoi
    from comic_amd.scst.scorers import captionScorer
    from comic_amd.scst import prepro_ngrams
    import types
    out = {}
    rng = np.random.default_rng(7)
    # ---- beam-3 inference, word baseline --------------------------------------------------
    B = 50
    enc50 = nets.CnnEncoder(plan, cnn_params, B, 'bf16', device)
    tune = os.environ.get('COMIC_AUTOTUNE', '1') == '1'
    if tune:
        enc50.autotune()                   # setup, untimed (as for the training encoder)
    imgs = torch.from_numpy(rng.uniform(-1, 1, (B, IMG, IMG, 3)).astype(np.float32)).to(device)
    V = 25599
    spec = cdec.DecoderSpec(V=V, H=1, fm_projection=None, token_type='word', start_id=V - 2, end_id=V - 1)
    dec = cdec.Decoder(spec, None, device, seed=3)
    max_steps = 30
    for _ in range(2):
        im, fm = enc50.forward(imgs, use_graph=True)
        r = {'predicted_ids': dec.beam_search_ids(fm, im, 3, max_steps)()}
    torch.cuda.synchronize()
    # as CaptionModel.infer runs it: ONE encoder forward covers the next G = 4 batches (200 images: 1.6x the MFMA rate of
    # a 50-image forward) and runs on a second stream under the decode steps of the current group
    # (trainer.EncoderPipeline); every timed batch pays a quarter of a forward and one decode
    from comic_amd import trainer as _tr
    G = 4
    encG = nets.CnnEncoder(plan, cnn_params, B * G, 'bf16', device, weights_from=enc50)
    if tune:
        encG.autotune()
    imgsG = imgs.repeat(G, 1, 1, 1).contiguous()
    pipe = _tr.EncoderPipeline(encG, B, G, device)
    pipe.submit(imgsG)
    # as `infer.py` runs it (CaptionModel.infer_pipelined): the decode loops of THREE batches in flight on three streams (a beam
    # step is five dependent launches of 50-230 workgroups; the kernels of the other, independent batches fill the holes)
    NL = max(1, min(5, int(os.environ.get('COMIC_INFER_IN_FLIGHT', '3'))))     # (as CaptionModel.infer_pipelined clamps it)
    lanes = [streams.lane(torch, device, 'infer%d' % k) for k in range(NL)]
    pend = [None] * NL

    def decode_batches(nb):
        r = None
        for i in range(nb):
            im_s, fm_s, rel = pipe.take()
            im, fm = im_s.clone(), fm_s.clone()
            if rel():
                pipe.submit(imgsG)
            k = i % NL
            if pend[k] is not None:
                r = pend[k]()
            lanes[k].wait_stream(torch.cuda.current_stream())
            with torch.cuda.stream(lanes[k]):
                pend[k] = dec.beam_search_ids(fm, im, 3, max_steps, slot=k)
            im.record_stream(lanes[k]); fm.record_stream(lanes[k])
        for k in range(NL):
            if pend[k] is not None:
                r = pend[k]()
                pend[k] = None
        return r
    decode_batches(3 * G)                  # untimed: captures the group encoder's graph and both decode graphs
    torch.cuda.synchronize()
    n, t0 = 4 * G, time.perf_counter()
    r = {'predicted_ids': decode_batches(n)}
    torch.cuda.synchronize()
    dt = (time.perf_counter() - t0) / n
    out['beam3_captions_per_sec'] = round(B / dt, 1)
    out['beam3_config'] = ('word tokens V=25599, 1 head, fm_projection none, batch 50, max 30 steps, %d steps executed; one encoder '
                           'forward per 4 batches on a side stream and the decode loops of three batches in flight '
                           '(CaptionModel.infer_pipelined)' % r['predicted_ids'].shape[0])
    del pipe, encG
    if os.environ.get('COMIC_EXTRAS_ONLY') == 'beam':
        return out
    t0 = time.perf_counter()
    for _ in range(3):
        im, fm = enc50.forward(imgs, use_graph=True)
        r = {'predicted_ids': dec.beam_search_ids(fm, im, 3, max_steps)()}
    torch.cuda.synchronize()
    out['beam3_serial_captions_per_sec'] = round(B * 3 / (time.perf_counter() - t0), 1)
    # decode loop alone against the HBM roofline of the vocabulary projection (per step D*V*s bytes of
    # W_o + rows*V*4 bytes of logits; s = 4: the decoder is fp32)
    t0 = time.perf_counter()
    for _ in range(n):
        r = {'predicted_ids': dec.beam_search_ids(fm, im, 3, max_steps)()}
    torch.cuda.synchronize()
    steps_ex = int(r['predicted_ids'].shape[0])
    us_step = (time.perf_counter() - t0) / n / steps_ex * 1e6
    # bytes that move per step: W_o and the LSTM kernel once each (packed hi / lo bf16 = 4 bytes per weight); the streaming
    # projection never writes its logits, so none are counted
    bytes_step = spec.D * V * 4 + (spec.E + spec.A + spec.D) * 4 * spec.D * 4
    out['beam3_roofline'] = {'bound': 'hbm', 'bytes_per_step': bytes_step, 'us_per_step': round(us_step, 1),
                             'achieved': round(bytes_step / us_step / 1e3, 1), 'peak': 8000.0, 'unit': 'GB/s',
                             'frac': round(bytes_step / us_step / 1e3 / 8000.0, 4),
                             'note': 'whole decode step (streaming LSTM, attention, streaming logits + top-k, merge) over the bytes of W_o '
                                     'and of the LSTM kernel, each streamed once per step for all 150 rows'}
    del dec, enc50
    # ---- SCST step, COMIC-256 -------------------------------------------------------------
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
    dec.params.view('b_o')[257] = 2.0
    opt = optim.AdamTF(dec.params)
    enc_s = nets.CnnEncoder(plan, cnn_params, Bs, 'bf16', device)
    if tune:
        enc_s.autotune()
    imgs = torch.from_numpy(rng.uniform(-1, 1, (Bs, IMG, IMG, 3)).astype(np.float32)).to(device)
    iters = 40                                   # infer_max_length 20 x 2 radix digits

    ahead = {}

    def encode_now():
        # as train_fn's SCST loop runs it: the encoder forward of the NEXT step is enqueued as soon as this step's
        # rollouts are back, so it runs on the device while the host scores them; one forward per step either way
        if 'f' in ahead:
            return ahead.pop('f')
        im, fm = enc_s.forward(imgs, use_graph=True)
        return im.clone(), fm.clone()

    def encode_ahead():
        im, fm = enc_s.forward(imgs, use_graph=True)
        ahead['f'] = (im.clone(), fm.clone())

    def scst_step():
        im, fm = encode_now()
        # as train_fn's SCST loop: the greedy rollout runs on the device while the host turns the beam rollouts into text
        # and ids; the update's forward pass (no reward enters it) and the next step's encoder forward while it scores them
        fetch_beam = dec.beam_search_ids(fm, im, W, iters)
        fetch_greedy = dec.greedy(fm, im, iters, defer=True)
        beam = fetch_beam().transpose(2, 1, 0)             # (W,B,T)
        caps, ids = radix_ids_to_captions_and_ids(beam.reshape(-1, beam.shape[-1]), cfg, table)     # (as train_fn's SCST loop)
        cap_beam = [[c] for c in caps]
        im, fm = im.repeat(W, 1), fm.repeat(W, 1, 1)     # = encoder(imgs tiled W times): frozen CNN, run once
        dec.train_step(fm, im, ids, training=True, use_graph=True, phase='fwd')
        cap_greedy = [[c] for c in id_to_caption(fetch_greedy()[0], cfg)]
        encode_ahead()
        hypos, sc_s, sc_g = scorer.get_hypo_scores(refs, cap_beam, cap_greedy)
        res = dec.train_step(None, None, ids, rewards=(sc_s - sc_g).astype(np.float32), training=True, use_graph=True, phase='bwd')
        opt.step(dec.grads, 1e-3)
        return res
    for _ in range(2):
        res = scst_step()
    torch.cuda.synchronize()
    n, t0 = 5, time.perf_counter()
    for _ in range(n):
        res = scst_step()
    torch.cuda.synchronize()
    out['scst_one_step'] = {'images_per_sec': round(Bs * n / (time.perf_counter() - t0), 1), 'time_steps': int(res['Tp']),
                            'config': 'random weights with an EOS bias of +2: every rollout ends after its first step '
                                      '(device-side early exit) -- the LOWER bracket of the SCST step'}
    # the upper bound: EOS never emitted, every rollout and the training step run all 40 time steps
    dec.params.view('b_o')[257] = -30.0
    for _ in range(2):
        res = scst_step()
    torch.cuda.synchronize()
    n, t0 = 3, time.perf_counter()
    for _ in range(n):
        res = scst_step()
    torch.cuda.synchronize()
    out['scst_full_length'] = {'images_per_sec': round(Bs * n / (time.perf_counter() - t0), 1), 'time_steps': int(res['Tp']),
                               'config': 'the same step with EOS suppressed: greedy + beam-7 run all 40 steps, the 224-hypothesis '
                                         'training step has T\' = %d (persistent loops in four launches: path %d)'
                                         % (int(res['Tp']), int(dec.lib.comic_decoder_train_path()))}
    # the figure reported as scst_images_per_sec: caption lengths as on MS-COCO (N ~ U{8..14} words =
    # 16..28 radix digits + EOS).  Random weights cannot be made to stop there by themselves, so EOS stays suppressed, the
    # rollouts run 29 steps (the length of the longest caption of a batch: dynamic_decode stops when EVERY row has ended)
    # and each sampled hypothesis is cut at its own drawn length before it is scored and trained on.
    real_iters = 29
    len_rng = np.random.default_rng(11)

    def cut(ids2d):
        ids2d = np.array(ids2d, copy=True)
        for r in range(ids2d.shape[0]):
            L = 2 * int(len_rng.integers(8, 15))
            ids2d[r, L:] = 257
        return ids2d

    # as train_fn's SCST loop with --encoder_group 8 (its auto value at batch 32: the frozen CNN's forward for the images of
    # the next eight steps is ONE launch chain of 256 images, enqueued while the host scores the step that used up the
    # previous group)
    G_S = 8
    enc_g = nets.CnnEncoder(plan, cnn_params, Bs * G_S, 'bf16', device, weights_from=enc_s)
    if tune:
        enc_g.autotune()
    imgs_g = torch.from_numpy(rng.uniform(-1, 1, (Bs * G_S, IMG, IMG, 3)).astype(np.float32)).to(device)
    feats = []

    def encode_group():
        im_g, fm_g = enc_g.forward(imgs_g, use_graph=True)
        im_g, fm_g = im_g.clone(), fm_g.clone()
        feats.extend((im_g[k * Bs:(k + 1) * Bs], fm_g[k * Bs:(k + 1) * Bs]) for k in range(G_S))

    def scst_step_realistic():
        if not feats:
            encode_group()
        im, fm = feats.pop(0)
        fetch_beam = dec.beam_search_ids(fm, im, W, real_iters)
        fetch_greedy = dec.greedy(fm, im, real_iters, defer=True)
        beam = fetch_beam().transpose(2, 1, 0)             # (W,B,T)
        caps, ids = radix_ids_to_captions_and_ids(cut(beam.reshape(-1, beam.shape[-1])), cfg, table)
        cap_beam = [[c] for c in caps]
        im, fm = im.repeat(W, 1), fm.repeat(W, 1, 1)
        dec.train_step(fm, im, ids, training=True, use_graph=True, phase='fwd')
        cap_greedy = [[c] for c in id_to_caption(cut(fetch_greedy()[0]), cfg)]
        if not feats:
            encode_group()
        hypos, sc_s, sc_g = scorer.get_hypo_scores(refs, cap_beam, cap_greedy)
        res = dec.train_step(None, None, ids, rewards=(sc_s - sc_g).astype(np.float32), training=True, use_graph=True, phase='bwd')
        opt.step(dec.grads, 1e-3)
        return res
    for _ in range(G_S):
        res = scst_step_realistic()
    torch.cuda.synchronize()
    n, t0 = 2 * G_S, time.perf_counter()       # whole groups: two encoder forwards of 256 images inside the timed region
    for _ in range(n):
        res = scst_step_realistic()
    torch.cuda.synchronize()
    out['scst_images_per_sec'] = round(Bs * n / (time.perf_counter() - t0), 1)
    out['scst_conv_mfma_frac'] = round(out['scst_images_per_sec'] * FLOP_PER_IMAGE_CNN / PEAK_BF16_MFMA, 5)
    out['scst_config'] = ('COMIC-256, batch 32, greedy + beam-7 rollouts of %d steps (the longest caption of a batch), every '
                          'hypothesis cut at its own length N ~ U{8..14} words (16..28 radix digits + EOS), C++ CIDEr-D+BLEU-4 '
                          'reward, 224-hypothesis training step with T\' = %d (encoder once per 8 steps: 256 images per forward, '
                          'features tiled); brackets: '
                          'scst_one_step (rollouts end at once) and scst_full_length (40 steps)' % (real_iters, int(res['Tp'])))
    del enc_s, dec, opt
    torch.cuda.empty_cache()
    # ---- cnn_finetune step (configs[2]: CNN + decoder trainable, batch 32) ---------------------
    from comic_amd import trainer
    Bf = 32
    plan_ft = nets.CnnPlan('inception_v3', (IMG, IMG))           # trainable CNN: the plain plan (has a backward)
    tr = trainer.CaptionTrainer(cnn_params, cdec.DecoderSpec(), None, Bf, (IMG, IMG), 'bf16', device, seed=5, plan=plan_ft)
    tr.enable_cnn_finetune()            # (CnnEncoder.autotune_backward exists; measured equal to the heuristic tiles: 6.3-6.5 ms either way)
    if tune:
        tr.encoder.autotune()              # forward variants of the plain plan (setup, untimed)
    imgs = torch.from_numpy(rng.uniform(-1, 1, (Bf, IMG, IMG, 3)).astype(np.float32)).to(device)
    caps = synth_captions(rng, Bf)
    for _ in range(3):
        tr.finetune_step(imgs, caps)
    torch.cuda.synchronize()
    n, t0 = 10, time.perf_counter()
    for _ in range(n):
        res = tr.finetune_step(imgs, caps)
    torch.cuda.synchronize()
    out['cnn_finetune_images_per_sec'] = round(Bf * n / (time.perf_counter() - t0), 1)
    out['cnn_finetune_config'] = ('COMIC-256 + InceptionV3 trainable (94 conv weights + BN betas, bf16 activations / '
                                  'fp32 masters), batch 32, 224x224; loss %.4f' % float(res['loss']))
    # forward + backward-data + backward-weight = 3x the forward conv FLOPs per image
    out['cnn_finetune_conv_mfma_frac'] = round(out['cnn_finetune_images_per_sec'] * 3 * FLOP_PER_IMAGE_CNN / PEAK_BF16_MFMA, 5)
    del tr
    torch.cuda.empty_cache()
    # the same step on the bf16x3 plan: the trainable CNN at the fp32 parity bar (gradients 1.3e-5 from the fp32 oracle where
    # the bf16 plan is at 4e-2; tests/test_gpu_path.py test_inception_v3_backward_224_bf16x3_meets_the_fp32_bar)
    try:
        plan_x3 = nets.CnnPlan('inception_v3', (IMG, IMG), x3=True)
        tr = trainer.CaptionTrainer(cnn_params, cdec.DecoderSpec(), None, Bf, (IMG, IMG), 'bf16x3', device, seed=5, plan=plan_x3)
        tr.enable_cnn_finetune()
        if tune:
            tr.encoder.autotune()
        for _ in range(3):
            tr.finetune_step(imgs, caps)
        torch.cuda.synchronize()
        n, t0 = 10, time.perf_counter()
        for _ in range(n):
            res = tr.finetune_step(imgs, caps)
        torch.cuda.synchronize()
        out['cnn_finetune_x3'] = {'images_per_sec': round(Bf * n / (time.perf_counter() - t0), 1), 'cnn_dtype': 'bf16x3',
                                  'loss': round(float(res['loss']), 4),
                                  'note': 'cnn_finetune at the fp32 parity bar on the bf16 matrix cores: activations / d conv as '
                                          'hi-lo regions, 3 bf16 products per conv each way, fp32 gradient buffers'}
        del tr
    except Exception as e:
        out['cnn_finetune_x3'] = {'error': repr(e)}
    torch.cuda.empty_cache()
    # ---- decoder-mode XE at 299 x 299: the north star's 8x8x2048 map (M = 64), batch 64, serial steps ----------------
    plan299 = nets.CnnPlan('inception_v3', (299, 299), pool_after_projection=True, fuse_pools=True)
    # frozen CNN: ONE forward covers the batches of the next G_299 steps, as in the headline (serial here: forward, then its steps)
    G_299 = 10
    tr = trainer.CaptionTrainer(cnn_params, cdec.DecoderSpec(M=64), None, BATCH, (299, 299), 'bf16', device, seed=6, plan=plan299,
                                encoder_group=G_299)
    if tune:
        tr.encoder.autotune()
    imgs = torch.from_numpy(rng.uniform(-1, 1, (BATCH * G_299, 299, 299, 3)).astype(np.float32)).to(device)
    caps = synth_captions(rng, BATCH)

    def group_299():
        im_g, fm_g = tr.encoder.forward(imgs, use_graph=True)
        for j in range(G_299):
            r = tr.decoder.train_step(fm_g[j * BATCH:(j + 1) * BATCH], im_g[j * BATCH:(j + 1) * BATCH], np.asarray(caps),
                                      training=True, use_graph=tr.use_graph_decoder)
            tr.opt.step(tr.decoder.grads, tr.lr())
        return r
    for _ in range(2):
        group_299()
    torch.cuda.synchronize()
    n, t0 = 2, time.perf_counter()
    for _ in range(n):
        res = group_299()
    torch.cuda.synchronize()
    dt = (time.perf_counter() - t0) / (n * G_299)
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(5):
        tr.encoder.forward(imgs, use_graph=True)
    e1.record(); e1.synchronize()
    fwd_ms = e0.elapsed_time(e1) / 5 / G_299                     # per step of 64 images
    flop299 = 2 * plan299.macs
    out['xe_299'] = {'images_per_sec': round(BATCH / dt, 1), 'ms_per_step': round(dt * 1e3, 3), 'feature_map': '8x8x2048 (M = 64)',
                     'config': 'COMIC-256, InceptionV3 frozen, batch 64, 299x299x3, one encoder forward per %d steps (not overlapped)' % G_299,
                     'cnn_forward_ms_per_step': round(fwd_ms, 3), 'flop_per_image': flop299,
                     'decoder_time_loops': {0: 'per-step launches', 1: 'persistent forward', 3: 'persistent forward + backward'}.get(
                         int(tr.decoder.lib.comic_decoder_train_path()), '?'),
                     'cnn_mfma_frac': round(flop299 * BATCH / (fwd_ms * 1e-3) / PEAK_BF16_MFMA, 5), 'loss': round(float(res['loss']), 4)}
    del imgs
    del tr
    torch.cuda.empty_cache()
    # ---- the fp32 plan (the one that meets the north star's 1e-3 against the oracle): its throughput, and what the
    # benchmarked bf16 plan deviates from it on ONE XE step of the same batch (same weights, dropout off) -------------------
    plan_f32 = nets.CnnPlan('inception_v3', (IMG, IMG))
    spec0 = cdec.DecoderSpec()
    tr32 = trainer.CaptionTrainer(cnn_params, spec0, None, BATCH, (IMG, IMG), 'f32', device, seed=8, plan=plan_f32)
    p_same = tr32.decoder.params.to_numpy()
    tr16 = trainer.CaptionTrainer(cnn_params, spec0, p_same, BATCH, (IMG, IMG), 'bf16', device, seed=8, plan=plan)
    if tune:
        tr16.encoder.autotune()
    imgs = torch.from_numpy(rng.uniform(-1, 1, (BATCH, IMG, IMG, 3)).astype(np.float32)).to(device)
    caps = synth_captions(rng, BATCH)
    dev_rel = {}
    got = {}
    for name, t in (('f32', tr32), ('bf16', tr16)):
        im_e, fm_e = t.encoder.forward(imgs, use_graph=False)
        r = t.decoder.train_step(fm_e, im_e, caps, training=False)
        torch.cuda.synchronize()
        got[name] = dict(fm=fm_e.float().cpu().numpy(), logits=r['logits'].cpu().numpy(), loss=float(r['loss']),
                         grads=t.decoder.grads.to_numpy())

    def rel(a, b):
        return float(np.abs(a - b).max() / (np.abs(b).max() + 1e-30))
    dev_rel['feature_map'] = rel(got['bf16']['fm'], got['f32']['fm'])
    dev_rel['logits'] = rel(got['bf16']['logits'], got['f32']['logits'])
    dev_rel['loss'] = abs(got['bf16']['loss'] - got['f32']['loss']) / abs(got['f32']['loss'])
    gk = {k: rel(got['bf16']['grads'][k], got['f32']['grads'][k]) for k in got['f32']['grads']}
    dev_rel['grad_max'] = max(gk.values())
    dev_rel['grad_worst'] = max(gk, key=gk.get)
    for _ in range(3):
        tr32.xe_step(imgs, caps)
    torch.cuda.synchronize()
    n, t0 = 8, time.perf_counter()
    for _ in range(n):
        res = tr32.xe_step(imgs, caps)
    torch.cuda.synchronize()
    dt = (time.perf_counter() - t0) / n
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(3):
        tr32.encoder.forward(imgs, use_graph=True)
    e1.record(); e1.synchronize()
    f32_ms = e0.elapsed_time(e1) / 3
    out['xe_f32'] = {'images_per_sec': round(BATCH / dt, 1), 'ms_per_step': round(dt * 1e3, 3),
                     'config': 'the same step with the exact-fp32 CNN plan (v_mfma_f32_16x16x4_f32, plain op order: the plan the '
                               '1e-3 parity tests run), batch 64, one forward per step, no overlap',
                     'cnn_forward_ms': round(f32_ms, 3),
                     'cnn_f32_mfma_frac': round(FLOP_PER_IMAGE_CNN * BATCH / (f32_ms * 1e-3) / 157.3e12, 5),
                     'bf16_plan_vs_f32_plan': {k: (round(v, 6) if isinstance(v, float) else v) for k, v in dev_rel.items()},
                     'note': 'bf16_plan_vs_f32_plan: max|a-b| / max|b| per tensor between the benchmarked bf16 plan and the fp32 plan '
                             'on one XE step of the same batch and weights, dropout off -- the price of the benchmarked precision'}
    # ---- the fast plan AT the parity bar: bf16x3 (hi / lo split activations and filters on the bf16 matrix cores, nets.CnnPlan(x3=
    # True), COMIC_OP_X3): its deviation from the fp32 plan on the same step, and its throughput with one forward per G_X3 steps --
    G_X3 = 5                                    # 320 images per forward (a bf16x3 buffer of 109x109x192 stays below 2^31 bytes)
    plan_x3 = nets.CnnPlan('inception_v3', (IMG, IMG), x3=True, pool_after_projection=True)     # the frozen-CNN form, as CaptionModel builds it
    trx1 = trainer.CaptionTrainer(cnn_params, spec0, p_same, BATCH, (IMG, IMG), 'bf16x3', device, seed=8, plan=plan_x3)
    im_e, fm_e = trx1.encoder.forward(imgs, use_graph=False)
    r = trx1.decoder.train_step(fm_e, im_e, caps, training=False)
    torch.cuda.synchronize()
    gx = dict(fm=fm_e.float().cpu().numpy(), logits=r['logits'].cpu().numpy(), loss=float(r['loss']), grads=trx1.decoder.grads.to_numpy())
    dev_x3 = {'feature_map': rel(gx['fm'], got['f32']['fm']), 'logits': rel(gx['logits'], got['f32']['logits']),
              'loss': abs(gx['loss'] - got['f32']['loss']) / abs(got['f32']['loss'])}
    gk = {k: rel(gx['grads'][k], got['f32']['grads'][k]) for k in got['f32']['grads']}
    dev_x3['grad_max'] = max(gk.values())
    dev_x3['grad_worst'] = max(gk, key=gk.get)
    del trx1, gx
    trx = trainer.CaptionTrainer(cnn_params, spec0, None, BATCH, (IMG, IMG), 'bf16x3', device, seed=8, plan=plan_x3,
                                 encoder_group=G_X3)
    if tune:
        trx.encoder.autotune()
    imgs_x = torch.from_numpy(rng.uniform(-1, 1, (BATCH * G_X3, IMG, IMG, 3)).astype(np.float32)).to(device)

    def x3_group():
        im_g, fm_g = trx.encoder.forward(imgs_x, use_graph=True)
        for j in range(G_X3):
            rr = trx.decoder.train_step(fm_g[j * BATCH:(j + 1) * BATCH], im_g[j * BATCH:(j + 1) * BATCH], np.asarray(caps),
                                        training=True, use_graph=trx.use_graph_decoder)
            trx.opt.step(trx.decoder.grads, trx.lr())
        return rr
    for _ in range(2):
        x3_group()
    torch.cuda.synchronize()
    n, t0 = 4, time.perf_counter()
    for _ in range(n):
        res = x3_group()
    torch.cuda.synchronize()
    dt = (time.perf_counter() - t0) / (n * G_X3)
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(5):
        trx.encoder.forward(imgs_x, use_graph=True)
    e1.record(); e1.synchronize()
    x3_ms = e0.elapsed_time(e1) / 5 / G_X3                      # per step of 64 images
    out['xe_x3'] = {'images_per_sec': round(BATCH / dt, 1), 'ms_per_step': round(dt * 1e3, 3),
                    'config': 'the same step with the bf16x3 CNN plan: activations stored as [hi | lo | hi] bf16 channel regions, filters '
                              '[W_hi | W_hi | W_lo], v_mfma_f32_16x16x32_bf16 over 3x the input channels (hi*W_hi + lo*W_hi + hi*W_lo, fp32 '
                              'accumulation), the pool branches behind their projections (the frozen-CNN rewrite); batch 64, one forward per %d steps, not overlapped' % G_X3,
                    'cnn_forward_ms_per_step': round(x3_ms, 3),
                    'cnn_mfma_frac_useful': round(FLOP_PER_IMAGE_CNN * BATCH / (x3_ms * 1e-3) / PEAK_BF16_MFMA, 5),
                    'cnn_mfma_frac_issued': round(3 * FLOP_PER_IMAGE_CNN * BATCH / (x3_ms * 1e-3) / PEAK_BF16_MFMA, 5),
                    'bf16x3_plan_vs_f32_plan': {k: (round(v, 7) if isinstance(v, float) else v) for k, v in dev_x3.items()},
                    'loss': round(float(res['loss']), 4),
                    'note': 'bf16x3_plan_vs_f32_plan as bf16_plan_vs_f32_plan above; against the ORACLE the plan passes the 1e-3 end-point test '
                            'of the fp32 plan (tests/test_gpu_path.py::test_inception_v3_forward_224_bf16x3_meets_the_fp32_bar)'}
    del trx, imgs_x
    del tr32, tr16, got
    torch.cuda.empty_cache()
    # ---- the reference CLI's DEFAULT backbone: Inception-V1, attention over Mixed_4f = 14x14x832 (M = 196), batch 64 ------
    plan_v1 = nets.CnnPlan('inception_v1', (IMG, IMG))
    v1_params = plan_v1.init_params(seed=0)
    Hf, Wf, Cf = plan_v1.fm_dims()
    spec_v1 = cdec.DecoderSpec(M=Hf * Wf, C=Cf, Cg=1024)
    # frozen CNN: ONE forward covers the batches of the next G_V1 steps, as in the headline (serial here: forward, then its steps)
    G_V1 = 10
    trv = trainer.CaptionTrainer(v1_params, spec_v1, None, BATCH, (IMG, IMG), 'bf16', device, seed=9, plan=plan_v1,
                                 encoder_group=G_V1)
    if tune:
        trv.encoder.autotune()
    imgs_g = torch.from_numpy(rng.uniform(-1, 1, (BATCH * G_V1, IMG, IMG, 3)).astype(np.float32)).to(device)

    def v1_group():
        im_g, fm_g = trv.encoder.forward(imgs_g, use_graph=True)
        for j in range(G_V1):
            r = trv.decoder.train_step(fm_g[j * BATCH:(j + 1) * BATCH], im_g[j * BATCH:(j + 1) * BATCH], np.asarray(caps),
                                       training=True, use_graph=trv.use_graph_decoder)
            trv.opt.step(trv.decoder.grads, trv.lr())
        return r
    for _ in range(2):
        v1_group()
    torch.cuda.synchronize()
    n, t0 = 2, time.perf_counter()
    for _ in range(n):
        res = v1_group()
    torch.cuda.synchronize()
    dt = (time.perf_counter() - t0) / (n * G_V1)
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(5):
        trv.encoder.forward(imgs_g, use_graph=True)
    e1.record(); e1.synchronize()
    v1_ms = e0.elapsed_time(e1) / 5 / G_V1                      # per step of 64 images
    flop_v1 = 2 * plan_v1.macs
    v1_path = int(trv.decoder.lib.comic_decoder_train_path())
    # attention bytes per time step (tied): B*M*D keys + B*(2D + H*M) small vectors, fp32
    att_bytes = 4 * (BATCH * spec_v1.M * spec_v1.D + BATCH * (2 * spec_v1.D + spec_v1.H * spec_v1.M))
    out['xe_v1'] = {'images_per_sec': round(BATCH / dt, 1), 'ms_per_step': round(dt * 1e3, 3),
                    'config': 'reference default (train.py:56,65): Inception-V1, Mixed_4f %dx%dx%d (M = %d), COMIC-256, batch 64, '
                              '224x224x3, one encoder forward per %d steps (not overlapped)' % (Hf, Wf, Cf, Hf * Wf, G_V1),
                    'cnn_forward_ms_per_step': round(v1_ms, 3), 'flop_per_image': flop_v1,
                    'cnn_mfma_frac': round(flop_v1 * BATCH / (v1_ms * 1e-3) / PEAK_BF16_MFMA, 5),
                    'decoder_ms': round(dt * 1e3 - v1_ms, 3), 'decoder_time_loops': {0: 'per-step launches', 1: 'persistent forward', 3: 'persistent forward + backward'}.get(v1_path, str(v1_path)),
                    'attention_roofline': {'bound': 'hbm', 'bytes_per_time_step': att_bytes, 'unit': 'GB/s', 'peak': 8000.0,
                                           'note': 'keys of a batch row are 401 KB fp32 at M = 196: more than a CU\'s LDS.  The persistent '
                                                   'forward loop holds a channel quarter of them per workgroup, the backward loop its own '
                                                   'memory rows (round 3); round 2 ran both as per-step launches'},
                    'loss': round(float(res['loss']), 4)}
    del trv, imgs_g
    torch.cuda.empty_cache()
    # ---- the 224 x 224 encoder at 64 images per forward (no grouping): the same roofline definition as `roofline.frac` --
    enc64 = nets.CnnEncoder(plan, cnn_params, BATCH, 'bf16', device)
    if tune:
        enc64.autotune()
    imgs = torch.from_numpy(rng.uniform(-1, 1, (BATCH, IMG, IMG, 3)).astype(np.float32)).to(device)
    for _ in range(3):
        enc64.forward(imgs, use_graph=True)
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(10):
        enc64.forward(imgs, use_graph=True)
    e1.record(); e1.synchronize()
    ms64 = e0.elapsed_time(e1) / 10
    out['cnn_frac_at_batch64'] = {'cnn_forward_ms': round(ms64, 4), 'images_per_forward': BATCH,
                                  'frac': round(FLOP_PER_IMAGE_CNN * BATCH / (ms64 * 1e-3) / PEAK_BF16_MFMA, 5)}
    del enc64, imgs
    torch.cuda.empty_cache()
    try:
        out['input_pipeline'] = input_pipeline_rate(device)
    except Exception as e:                       # (e.g. no sample photographs on the box: the figure is optional)
        out['input_pipeline'] = {'error': repr(e)}
    return out


def input_pipeline_rate(device, n_files=128, threads=16):
    """JPEG files -> network input with the split decoder: Huffman decoding on `threads` C threads
    (libcomic_jpeg.so), inverse DCT / upsampling / colour conversion / resize / crop on the device; images/s of the loader
    alone (tools/loader_bench.py FILES=photo is the longer version, tools/train_files_bench.py the training step fed by it)."""
    import shutil
    import tempfile
    import torch
    from PIL import Image
    from comic_amd import inputs
    import sklearn
    sd = os.path.join(os.path.dirname(sklearn.__file__), 'datasets', 'images')
    photos = [Image.open(os.path.join(sd, f)).convert('RGB') for f in ('china.jpg', 'flower.jpg')]
    d = tempfile.mkdtemp()
    try:
        paths = []
        for i in range(n_files):
            im = photos[i % 2].crop((i % 40, i % 27, 600 + i % 40, 400 + i % 27)).resize((640, 480), Image.BICUBIC)
            p = os.path.join(d, '%d.jpg' % i)
            im.save(p, quality=90, subsampling=2)
            paths.append(p)
        kb = sum(os.path.getsize(p) for p in paths) / len(paths) / 1024
        jpool = inputs.JpegSplitPool(threads, max_batch=BATCH)
        pre = inputs.DevicePreprocessor(device, IMG, IMG)
        pre.enable_split(jpool, 6)
        params = [(False, 16, 16)] * BATCH
        ref = pre(list(map(inputs.decode_image, paths[:BATCH])), params)
        got = pre.finish(pre.pack_paths_split(paths[:BATCH], params))
        same = bool(torch.equal(got, ref))            # against PIL decode + the same device preprocessing
        n, inflight = 0, []
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for rep in range(24):
            for b in range(0, n_files, BATCH):
                inflight.append(pre.pack_paths_split(paths[b:b + BATCH], params))
                if len(inflight) > 3:
                    pre.finish(inflight.pop(0))
                n += BATCH
        while inflight:
            pre.finish(inflight.pop(0))
        torch.cuda.synchronize()
        dt = time.perf_counter() - t0
        jpool.close()
        return {'images_per_sec': round(n / dt, 1), 'host_threads': threads, 'bit_identical_to_pil_path': same,
                'files': '%d x 640x480 JPEG, quality 90, 4:2:0 (camera photographs re-encoded), %.0f KB on average' % (n_files, kb),
                'path': 'libcomic_jpeg.so (Huffman decoding on C threads into packed non-zero coefficients) -> '
                        'comic_jpeg_preprocess_packed (blocks expanded in LDS, inverse DCT, then resize / crop / scale with the '
                        'taps upsampled and colour-converted from the component planes)'}
    finally:
        shutil.rmtree(d, ignore_errors=True)


def heaviest_conv_launch(enc, plan, reps=20):
    """The single conv launch with the most FLOPs (InceptionV3 @224: Conv2d_4a_3x3, 52x52 3x3 80->192), alone on the
    GPU with its autotuned kernel variant: HIP events on the launch stream around `reps` back-to-back launches."""
    import ctypes as C
    import torch
    from comic_amd import _lib as L
    best, bi = 0, None
    plan = enc.plan           # (the encoder's own op table: small batches run the sibling plan without fused chains)
    for i, o in enumerate(plan.ops):
        if o['kind'] == 0 and not o.get('group'):
            fl = 2 * enc.batch * o['Ho'] * o['Wo'] * o['KH'] * o['KW'] * o['Cin'] * o['Cout']
            if fl > best:
                best, bi = fl, i
    if bi is None:
        return None
    o, st = plan.ops[bi], L.stream_ptr()
    first = C.byref(enc._ops, bi * C.sizeof(L.CnnOp))

    def run():
        L.check(enc.lib.comic_cnn_forward(first, 1, enc._bufptr, enc._bufch, enc._wt, enc.batch, 1, st), 'conv launch')
    run(); run()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps):
        run()
    e1.record(); e1.synchronize()
    us = e0.elapsed_time(e1) / reps * 1e3
    tile = int(enc._ops[bi].tile)
    return {'layer': '%dx%d %dx%d/%d %d->%d, batch %d' % (o['Ho'], o['Wo'], o['KH'], o['KW'], o['SH'], o['Cin'], o['Cout'], enc.batch),
            'kernel_variant': tile, 'kernel': 'conv_patch_kernel' if tile > 12 or tile == 0 else 'conv_igemm_dma_kernel',
            'flop': best, 'us': round(us, 2), 'achieved': round(best / us / 1e6, 1), 'unit': 'TFLOP/s',
            'frac': round(best / us / 1e6 / (PEAK_BF16_MFMA / 1e12), 4)}


def cpu_baseline(seconds_budget=20.0):
    """The oracle (numpy restatement of the reference graph: 'port') timed on this host's
    cores for the same step at the reference's CPU-runnable size (configs[0]: batch 2)."""
    from oracle import cnn_ref, decoder_ref as dr
    rng = np.random.default_rng(0)
    B = 2
    params = cnn_ref.init_params(0, IMG)
    cfg = dr.DecoderConfig()
    p = dr.init_params(cfg, 0)
    x = rng.uniform(-1, 1, (B, IMG, IMG, 3)).astype(np.float32)
    caps = synth_captions(rng, B)
    n, t0 = 0, time.time()
    while True:
        im, fm = cnn_ref.encoder(params, x)
        masks = dr.make_dropout_masks(cfg, B, caps.shape[1] - 1, fm.shape[1], n)
        out = dr.train_forward(p, cfg, fm, im, caps, masks)
        grads, _, _ = dr.train_backward(p, cfg, out)
        for k in p:
            p[k] = p[k] - np.float32(1e-3) * grads[k]
        n += 1
        if time.time() - t0 > seconds_budget or n >= 8:
            break
    dt = time.time() - t0
    out = dict(value=round(n * B / dt, 3), unit='images/sec', cores=os.cpu_count(), kind='port',
               sample='%d decoder-mode XE steps at batch %d (BASELINE configs[0], a plumbing-size batch: InceptionV3 fwd + decoder '
                      'fwd/bwd + SGD update), numpy/OpenBLAS oracle on all host cores' % (n, B))
    # the "framework CPU path" stand-in of BASELINE.md section 3.2(b): the same step on torch-CPU (oneDNN convolutions,
    # autograd backward of the decoder, TF-Adam), fp32 -- the literal TF-1 binary is not installable here
    try:
        import torch
        from oracle import torch_ref
        p32 = {k: v.astype(np.float32) for k, v in dr.init_params(cfg, 0).items()}
        state = None
        torch_ref.torch_train_step(params, p32, cfg, x, caps, state=None)          # warm-up (oneDNN primitive caches)
        m, t1 = 0, time.time()
        while True:
            _, state = torch_ref.torch_train_step(params, p32, cfg, x, caps, state=state)
            m += 1
            if time.time() - t1 > 10.0 or m >= 8:
                break
        out['torch_cpu'] = dict(value=round(m * B / (time.time() - t1), 3), unit='images/sec', threads=torch.get_num_threads(),
                                sample='%d steps at batch %d (BASELINE configs[0]: a plumbing-size batch, not a tuned CPU run): torch %s '
                                       'CPU, oneDNN conv forward + autograd decoder backward + TF-Adam, fp32' % (m, B, torch.__version__))
    except Exception as e:
        out['torch_cpu'] = {'error': repr(e)}
    return out


def pick_encoder_group(steps):
    """Steps per encoder forward: the largest divisor of the timed step count up to DEFAULT_ENC_GROUP (K timed steps
    then issue exactly K*BATCH images of encoder work); if none divides (a prime K above the default), the default with
    a last partly used group (more encoder work than consumed, never less)."""
    for g in range(min(DEFAULT_ENC_GROUP, steps), 1, -1):
        if steps % g == 0:
            return g
    return 1 if steps <= 1 else DEFAULT_ENC_GROUP


def launch_ranks(n):
    """`python -m torch.distributed.run --nnodes=1 --nproc-per-node n --master-addr 127.0.0.1 --master-port P bench.py
    <the same arguments>` as a child process; returns its exit code.  The port is one the kernel has just handed out."""
    import socket
    import subprocess
    with socket.socket() as s:
        s.bind(('127.0.0.1', 0))
        port = s.getsockname()[1]
    env = dict(os.environ)
    env.setdefault('HSA_ENABLE_IPC_MODE_LEGACY', '0')      # dmabuf IPC: RCCL between processes needs it on this stack
    env.setdefault('OMP_NUM_THREADS', '4')
    cmd = [sys.executable, '-m', 'torch.distributed.run', '--nnodes=1', '--nproc-per-node', str(n), '--master-addr',
           '127.0.0.1', '--master-port', str(port), os.path.abspath(__file__)] + sys.argv[1:]
    return subprocess.run(cmd, env=env).returncode


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument('--gpus', type=int, default=1)
    ap.add_argument('--steps', type=int, default=30)
    ap.add_argument('--warmup', type=int, default=5)
    ap.add_argument('--no-cpu-baseline', action='store_true')
    ap.add_argument('--no-extras', action='store_true')
    args = ap.parse_args()

    if args.gpus > 1 and 'WORLD_SIZE' not in os.environ:
        # plain `python bench.py --gpus N`: this process becomes the launcher.  It has not touched the GPU (no torch import,
        # no HIP call), starts one rank per GPU as a CHILD process group (never an exec) and leaves with the child's code;
        # rank 0's JSON line goes straight to the inherited stdout.
        raise SystemExit(launch_ranks(args.gpus))
    import torch
    import torch.distributed as dist
    world = int(os.environ.get('WORLD_SIZE', '1'))
    rank = int(os.environ.get('RANK', '0'))
    local_rank = int(os.environ.get('LOCAL_RANK', '0'))
    if args.gpus != world:
        raise SystemExit('bench.py: --gpus %d but the launcher started WORLD_SIZE=%d ranks' % (args.gpus, world))
    # COMIC_DIST_BACKEND=gloo: rehearsal of the N > 1 path on fewer GPUs than ranks (gloo moves CUDA tensors through the
    # host; ranks share the visible devices round-robin).  The driver's scaling runs use the default: nccl = RCCL.
    backend = os.environ.get('COMIC_DIST_BACKEND', 'nccl')
    local_rank %= max(1, torch.cuda.device_count()) if backend != 'nccl' else (local_rank + 1)
    if backend != 'nccl' and int(os.environ.get('LOCAL_WORLD_SIZE', str(world))) > torch.cuda.device_count():
        # ranks SHARING a GPU: the persistent time loops need all their workgroups resident at once (one per CU), which two
        # processes on one device cannot both have -- the bounded waits expire and the end-of-step gate voids the step
        # (NaN loss; observed).  The rehearsal runs the loops as per-step launches.
        os.environ.setdefault('COMIC_PERSIST', '0')
    torch.cuda.set_device(local_rank)
    device = 'cuda:%d' % local_rank
    if world > 1:
        if backend == 'nccl':
            dist.init_process_group('nccl', device_id=torch.device(device))
        else:
            dist.init_process_group(backend)

    from comic_amd import decoder as cdec, nets, trainer
    dp = trainer.DataParallel(dist if world > 1 else None)
    # decoder mode: the CNN is frozen, so the plan may use the forward-only pool-branch rewrite
    plan = nets.CnnPlan('inception_v3', (IMG, IMG), branch_streams=os.environ.get('COMIC_CNN_LANES', '0') == '1',
                        group_branches=os.environ.get('COMIC_CNN_GROUP', '1') == '1',
                        pool_after_projection=os.environ.get('COMIC_POOL_REWRITE', '1') == '1',
                        fuse_pools=os.environ.get('COMIC_POOL_REWRITE', '1') == '1' and os.environ.get('COMIC_FUSE_POOLS', '1') == '1')
    cnn_params = plan.init_params(seed=0)                       # random-init weights (no checkpoints offline)
    spec = cdec.DecoderSpec()                                   # COMIC-256 on a 5x5x2048 map
    overlap = os.environ.get('COMIC_OVERLAP', '1') == '1'
    # frozen CNN: one encoder forward covers the image batches of GROUP consecutive steps (trainer.CaptionTrainer)
    GROUP = int(os.environ.get('COMIC_ENC_GROUP', '0')) or pick_encoder_group(args.steps)
    ENC_BATCH = BATCH * GROUP
    tr = trainer.CaptionTrainer(cnn_params, spec, None, BATCH, (IMG, IMG), 'bf16', device, dp=dp, seed=1, plan=plan,
                                encoder_group=GROUP)
    if os.environ.get('COMIC_TUNE_POLITE', '0') == '1' and os.environ.get('COMIC_OVERLAP', '1') == '1':
        tr.enable_overlap(int(os.environ.get('COMIC_POLITE_LDS_KB', '84')))
    if os.environ.get('COMIC_AUTOTUNE', '1') == '1':
        tr.encoder.autotune(verbose=os.environ.get('COMIC_VERBOSE', '0') == '1',    # setup, untimed
                            cache=os.environ.get('COMIC_TUNE_CACHE') or None)
    # identical initial parameters on every rank (C2: broadcast)
    if world > 1:
        dist.broadcast(tr.decoder.params.data, 0)
    rng = np.random.default_rng(48964896 + rank)                # train.py:203 seed
    # N_IMG_SETS distinct image groups (resident in HBM) are served in rotation, so consecutive encoder forwards do not
    # re-read the same 385 MB; each forward starts with a device-to-device copy into the encoder's input buffer
    N_IMG_SETS = max(1, int(os.environ.get('COMIC_IMG_SETS', '2')))
    image_sets = [torch.from_numpy(rng.uniform(-1, 1, (ENC_BATCH, IMG, IMG, 3)).astype(np.float32)).to(device)
                  for _ in range(N_IMG_SETS)]
    images = image_sets[0]
    cap_sets = [synth_captions(rng, BATCH) for _ in range(4)]

    def barrier():
        if world > 1:
            dist.barrier()
        torch.cuda.synchronize()

    # XE is normalised by the GLOBAL token count: every step all-reduces its token count as a device
    # scalar on the stream (Decoder.train_step(dp=...)); the host never waits for it
    tr.use_graph = GRAPH_CNN and GRAPH_DEC
    if overlap:
        tr.enable_overlap(int(os.environ.get('COMIC_POLITE_LDS_KB', '84')))
    # the image groups live in HBM; a forward copies its group into the encoder's input buffer (device to device)
    tr.encoder.bufs[plan.input].copy_(images)
    # COMIC_BENCH_H2D=1 (not the headline line): every group's images come from pinned host memory over PCIe, copied
    # on the encoder's stream in front of its forward -- the PCIe-inclusive rate noted in DESIGN.md §5
    h2d = os.environ.get('COMIC_BENCH_H2D', '0') == '1' and overlap and GROUP > 1
    images_host = images.cpu().pin_memory() if h2d else None
    inbuf = tr.encoder.bufs[plan.input]
    n_sub = [0]

    def next_images():
        n_sub[0] += 1
        return image_sets[n_sub[0] % N_IMG_SETS]

    def submit(evp=None):
        if h2d:
            side = tr._pipe.side
            side.wait_stream(torch.cuda.current_stream())
            with torch.cuda.stream(side):
                inbuf.copy_(images_host, non_blocking=True)
            tr.submit_images(inbuf, evp)
        else:
            tr.submit_images(next_images(), evp)
    # setup (untimed, like the autotune): the encoder's hipGraph is captured on its second call and the decoder
    # allocates its buffers per caption shape on first use -- neither belongs to a timed step, whatever --warmup is.
    # No optimiser step here: the parameters the W warmup / K timed steps train are untouched.
    for j in range(2):
        im_embed, fm = tr.encoder.forward(images, use_graph=GRAPH_CNN)
    for cs in cap_sets:
        tr.decoder.train_step(fm[:BATCH], im_embed[:BATCH], cs, training=True, use_graph=GRAPH_DEC)
    torch.cuda.synchronize()
    for i in range(args.warmup):
        im_embed, fm = tr.encoder.forward(images, use_graph=GRAPH_CNN)
        tr.decoder.train_step(fm[:BATCH], im_embed[:BATCH], cap_sets[i % 4], training=True, use_graph=GRAPH_DEC)
        tr.opt.step(tr.decoder.grads, tr.lr())
    if overlap:
        submit()                          # batch(es) of the first timed step(s); the siblings are issued in the loop
    ev = [(torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)) for _ in range(args.steps)]
    n_fwd = [0]                           # encoder forwards issued inside the timed region
    # the timed region issues ~500 launches per step from Python: a generation-2 garbage collection in the middle
    # of it stalls the host for tens of milliseconds (measured: one 45 ms stall = +1.3 ms per step at 30 steps)
    import gc
    main_prio = int(os.environ.get('COMIC_MAIN_PRIORITY', '0'))
    main_stream = torch.cuda.Stream(device=device, priority=main_prio) if (overlap and main_prio != 0) else None
    gc.collect()
    gc.disable()
    barrier()
    t0 = time.perf_counter()
    def run_steps():
        res = None
        for i in range(args.steps):
            cap = cap_sets[i % 4]
            if overlap and GROUP > 1:
                # steps are served from groups of GROUP batches: the forward of the NEXT group (GROUP*BATCH images) is
                # issued on the side stream once the first step of this group holds its rows; K timed steps issue
                # ceil(K/GROUP) forwards = at least K*BATCH images
                im_embed, fm, release = tr.take_features()

                def consumed():
                    if release():
                        submit(ev[n_fwd[0]] if EVENTS else None)
                        n_fwd[0] += 1
                res = tr.decoder.train_step(fm, im_embed, cap, training=True, dp=dp, use_graph=GRAPH_DEC,
                                            on_inputs_consumed=consumed, copy_inputs=False)
            elif overlap:
                n_fwd[0] += 1
                # step i: decoder(batch i) on the main stream; the encoder forward of batch i+1 is issued on the
                # side stream as soon as the decoder holds its copy of batch i's features.  K timed steps issue
                # K encoder forwards and K decoder steps; HIP events on the side stream bracket the encoder.
                def consumed(i=i):
                    tr._ev_used.record(torch.cuda.current_stream())
                    tr._side.wait_event(tr._ev_used)
                    with torch.cuda.stream(tr._side):
                        if EVENTS: ev[i][0].record(tr._side)
                        tr._pending = tr.encoder.forward(next_images(), use_graph=GRAPH_CNN)
                        if EVENTS: ev[i][1].record(tr._side)
                        tr._ev_cnn.record(tr._side)
                torch.cuda.current_stream().wait_event(tr._ev_cnn)
                im_embed, fm = tr._pending
                res = tr.decoder.train_step(fm, im_embed, cap, training=True, dp=dp, use_graph=GRAPH_DEC,
                                            on_inputs_consumed=consumed)
            else:
                j = i % GROUP
                if j == 0:                      # serial: the forward of this group of steps, then its decoder steps
                    if EVENTS: ev[n_fwd[0]][0].record()
                    feats = tr.encoder.forward(next_images(), use_graph=GRAPH_CNN)
                    if EVENTS: ev[n_fwd[0]][1].record()
                    n_fwd[0] += 1
                im_embed, fm = feats[0][j * BATCH:(j + 1) * BATCH], feats[1][j * BATCH:(j + 1) * BATCH]
                res = tr.decoder.train_step(fm, im_embed, cap, training=True, dp=dp, use_graph=GRAPH_DEC)
            dp.exchange_and_step(tr.opt, tr.decoder.grads, tr.lr())
            if STEP_TIMES is not None:
                e = torch.cuda.Event(enable_timing=True); e.record(); STEP_TIMES.append((e, time.perf_counter()))
        return res
    # COMIC_MAIN_PRIORITY=-1 puts the decoder's launches on a high-priority stream above the encoder's side stream
    # (measured SLOWER, 2.04 vs 1.89 ms per step: the workgroups of the persistent loops then grab CUs one by one
    # and spin until the encoder's kernel has drained from all of them); default 0 = the same priority
    if main_stream is not None:
        main_stream.wait_stream(torch.cuda.current_stream())
        with torch.cuda.stream(main_stream):
            res = run_steps()
    else:
        res = run_steps()
    barrier()
    dt = time.perf_counter() - t0
    gc.enable()
    if STEP_TIMES:
        print('per-step GPU ms :', ' '.join('%.2f' % STEP_TIMES[i][0].elapsed_time(STEP_TIMES[i + 1][0]) for i in range(len(STEP_TIMES) - 1)))
        print('per-step host ms:', ' '.join('%.2f' % ((STEP_TIMES[i + 1][1] - STEP_TIMES[i][1]) * 1e3) for i in range(len(STEP_TIMES) - 1)))
        print('first step host issue done at %.2f ms after t0; last sync took %.2f ms' % ((STEP_TIMES[0][1] - t0) * 1e3, (time.perf_counter() - STEP_TIMES[-1][1]) * 1e3))
    t = torch.tensor([dt], dtype=torch.float64, device=device)
    if world > 1:
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
    dt = float(t.item())
    assert n_fwd[0] * ENC_BATCH >= args.steps * BATCH, 'fewer images encoded than consumed inside the timed region'
    cnn_ms = float(np.mean([a.elapsed_time(b) for a, b in ev[:n_fwd[0]]])) if EVENTS else float('nan')
    # the same forward alone on the GPU (not overlapped with the decoder), for reference
    tr.enable_overlap(0)                 # full occupancy again (the overlapped forward ran 1 workgroup per CU)
    for _ in range(3):
        tr.encoder.forward(images, use_graph=GRAPH_CNN)
    iso = [(torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)) for _ in range(10)]
    for a, b in iso:      # the input buffer already holds a group: the forward itself, no copy
        a.record(); tr.encoder.forward(inbuf, use_graph=GRAPH_CNN); b.record()
    torch.cuda.synchronize()
    cnn_iso_ms = float(np.mean([a.elapsed_time(b) for a, b in iso]))
    # the decoder step alone on the GPU (forward + backward + Adam on resident features, nothing beside it), HIP events
    dfm, dim = fm[:BATCH].clone(), im_embed[:BATCH].clone()
    for i in range(3):
        tr.decoder.train_step(dfm, dim, cap_sets[i % 4], training=True, use_graph=GRAPH_DEC)
        tr.opt.step(tr.decoder.grads, tr.lr())
    dec_ev = [(torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)) for _ in range(12)]
    for i, (a, b) in enumerate(dec_ev):
        a.record()
        rd = tr.decoder.train_step(dfm, dim, cap_sets[i % 4], training=True, use_graph=GRAPH_DEC)
        tr.opt.step(tr.decoder.grads, tr.lr())
        b.record()
    torch.cuda.synchronize()
    dec_ms = float(np.mean([a.elapsed_time(b) for a, b in dec_ev]))
    dec_Tp = int(rd['Tp'])
    dec_path = int(tr.decoder.lib.comic_decoder_train_path())
    loss = float(res['loss'])
    voided = tr.decoder.voided_steps()          # steps the device voided (persistent-loop timeout): must be none
    if world > 1:
        voided = int(dp.max_scalar(voided))
    assert voided == 0, 'the device voided %d training step(s) inside the run ("voided_steps": %d): no valid bench line' % (voided, voided)
    assert np.isfinite(loss), 'non-finite loss'
    top = heaviest_conv_launch(tr.encoder, plan) if rank == 0 else None

    if rank == 0:
        plan = tr.encoder.plan          # (the encoder's own op table)
        n_conv = sum(1 for o in plan.ops if o['kind'] in (0, 1, 8, 9)) + sum(1 for o in plan.ops if o['kind'] == 8) + 2 * sum(1 for o in plan.ops if o['kind'] == 9)
        n_launch = sum(1 for o in plan.ops if o['kind'] in (0, 1, 8, 9) and not o.get('group')) + len({o['group'] for o in plan.ops if o.get('group')})
        # Kernel quality is judged on the forward ALONE on the GPU (HIP events in this process, same graph /
        # launches, right after the timed loop); inside the timed region the same forward is deliberately
        # run at one workgroup per CU underneath the decoder step, so its wall time there says how well
        # the two overlap, not how good the kernel is.  Both are reported.
        achieved = FLOP_PER_IMAGE_CNN * ENC_BATCH / (cnn_iso_ms * 1e-3)
        achieved_in = FLOP_PER_IMAGE_CNN * ENC_BATCH / (cnn_ms * 1e-3)
        out = {
            'metric': 'images/sec (decoder-mode XE training, COMIC-256, InceptionV3 frozen)',
            'value': round(BATCH * world * args.steps / dt, 2), 'unit': 'images/sec', 'n_gpus': world,
            'steps': args.steps, 'warmup': args.warmup, 'ms_per_step': round(dt / args.steps * 1e3, 3),
            'higher_is_better': True, 'scaling': 'weak', 'vs_baseline': None, 'dtype': 'bf16', 'data': 'synthetic',
            'config': {'workload': 'MS-COCO decoder-mode XE, COMIC-256 (radix-256, 8 heads, tied), InceptionV3 '
                                   'frozen, batch 64/GPU, 224x224x3 (BASELINE configs[1])',
                       'per_gpu_batch': BATCH, 'global_batch': BATCH * world, 'image_size': IMG,
                       'feature_map': '5x5x2048', 'decoder_dtype': 'f32', 'parallelism': 'dp%d' % world,
                       'encoder_group': GROUP, 'encoder_forwards_in_timed_region': n_fwd[0],
                       'inputs': 'pinned host memory, H2D copy per group inside the timed region' if h2d else
                                 'resident in HBM, %d distinct image groups in rotation' % N_IMG_SETS},
            'roofline': {'bound': 'mfma', 'kernel': 'conv_igemm_dma / conv_patch (+ _grouped) / conv_img_chain kernels <bf16> (%d convs in %d '
                                                    'launches per forward of %d images, whole InceptionV3 forward timed with HIP events)' % (n_conv, n_launch, ENC_BATCH),
                         'achieved': round(achieved / 1e12, 3), 'peak': PEAK_BF16_MFMA / 1e12, 'unit': 'TFLOP/s',
                         'frac': round(achieved / PEAK_BF16_MFMA, 5), 'traffic': None,
                         'cnn_forward_ms': round(cnn_iso_ms, 4), 'images_per_forward': ENC_BATCH, 'heaviest_launch': top,
                         'in_timed_region': {'cnn_forward_ms': round(cnn_ms, 4), 'achieved': round(achieved_in / 1e12, 3),
                                             'frac': round(achieved_in / PEAK_BF16_MFMA, 5), 'overlapped': bool(overlap)},
                         'note': ('achieved/frac: the forward alone on the GPU, HIP events on its stream, measured in this '
                                  'process after the timed loop (agrees with profiles/*kernel_stats.csv).  in_timed_region: '
                                  'the same forward while it runs on a second stream under the decoder step of the '
                                  'previous batch%s (frozen CNN) at one conv workgroup per CU' %
                                  (' group: one forward per %d steps' % GROUP if GROUP > 1 else '')) if overlap else ''},
            'final_loss': round(loss, 5), 'voided_steps': voided,
        }
        # Second roofline entry: the decoder chain.  Algorithmic HBM bytes of one decoder step = every tensor of the step
        # moved once in each direction it is needed: parameters read in forward and backward, gradients written, Adam
        # (p, g, m, v read; p, m, v written), the feature map read by the key projection and by d W_m, and the per-step
        # activations the backward needs written once and read once.
        sp = tr.decoder.spec
        n_par = int(tr.decoder.params.numel)
        Tq, Bq, Dq, Wdq = dec_Tp, BATCH, sp.D, sp.E + sp.A + sp.D
        act = Tq * Bq * (Wdq + 4 * Dq + 3 * Dq + 2 * Dq + 2 * sp.H * sp.M + 2 * Dq + sp.V * 2 + 2 * Dq + 4 * Dq + sp.E)
        dec_bytes = 4 * (n_par * 10 + 2 * Bq * sp.M * sp.C + 2 * Bq * sp.M * Dq + 2 * act)
        out['decoder_roofline'] = {
            'bound': 'hbm', 'kernel': 'decoder training step alone (forward + backward time loops%s, time-batched GEMMs, '
                                      'TF-Adam), HIP events' % (' as persistent launches' if dec_path == 3 else ''),
            'bytes_per_step': dec_bytes, 'ms_per_step': round(dec_ms, 4), 'time_steps': Tq,
            'achieved': round(dec_bytes / (dec_ms * 1e-3) / 1e9, 1), 'peak': 8000.0, 'unit': 'GB/s',
            'frac': round(dec_bytes / (dec_ms * 1e-3) / 8e12, 5), 'loops_persistent': dec_path,
            'note': 'the chain is latency-bound, not bandwidth-bound: 2 x T\' dependent phases of three to four '
                    'cross-workgroup hand-offs each (DESIGN.md section 4, Persistent time loops)'}
        for tname in ('r06_cnn_hbm_traffic.json', 'r05_cnn_hbm_traffic.json', 'r04_cnn_hbm_traffic.json'):   # committed PMC passes (FETCH_SIZE / WRITE_SIZE,
            tfile = os.path.join(ROOT, 'profiles', tname)                         # corrected per the microarch guide), newest first
            tj = (json.load(open(tfile)).get('by_images_per_forward', {}).get(str(ENC_BATCH)) if os.path.isfile(tfile) else None)
            if tj:
                out['roofline']['traffic'] = tj['per_forward']['conv_only_bytes_corrected']
                out['roofline']['traffic_source'] = ('profiles/%s (bytes per InceptionV3 forward of %d images, conv kernels)'
                                                     % (tname, ENC_BATCH))
                break
        if not args.no_extras and world == 1:
            try:
                out['extras'] = extras(device, tr.encoder, cnn_params, plan)
            except Exception as e:          # secondary figures must never break the headline line
                out['extras'] = {'error': repr(e)}
            if isinstance(out['extras'].get('cnn_frac_at_batch64'), dict):   # the same definition at 64 images per forward
                out['roofline']['frac_at_batch64'] = out['extras']['cnn_frac_at_batch64']['frac']
        if not args.no_cpu_baseline and world == 1:
            out['cpu_baseline'] = cpu_baseline()
        else:
            out['cpu_baseline'] = None
        # compact digest of the secondary configurations as the LAST key, so that a reader who only keeps the tail of the
        # line (the driver's record does) still sees them
        ex = out.get('extras') or {}
        def _g(d, *ks):
            for k in ks:
                d = d.get(k) if isinstance(d, dict) else None
            return d
        out['summary'] = {'images_per_sec': out['value'], 'ms_per_step': out['ms_per_step'], 'cnn_frac': out['roofline']['frac'],
                          'frac_at_batch64': out['roofline'].get('frac_at_batch64'), 'decoder_ms': out['decoder_roofline']['ms_per_step'],
                          'beam3': ex.get('beam3_captions_per_sec'), 'beam3_frac': _g(ex, 'beam3_roofline', 'frac'),
                          'scst': ex.get('scst_images_per_sec'), 'cnn_finetune': ex.get('cnn_finetune_images_per_sec'),
                          'cnn_finetune_frac': ex.get('cnn_finetune_conv_mfma_frac'),
                          'cnn_finetune_x3': _g(ex, 'cnn_finetune_x3', 'images_per_sec'),
                          'xe_x3': _g(ex, 'xe_x3', 'images_per_sec'), 'xe_f32': _g(ex, 'xe_f32', 'images_per_sec'),
                          'xe_299': _g(ex, 'xe_299', 'images_per_sec'), 'xe_v1': _g(ex, 'xe_v1', 'images_per_sec'),
                          'input_pipeline': _g(ex, 'input_pipeline', 'images_per_sec'),
                          'cpu_images_per_sec': _g(out, 'cpu_baseline', 'value'), 'voided_steps': voided}
        print(json.dumps(out))
    if world > 1:
        dist.destroy_process_group()


if __name__ == '__main__':
    main()
