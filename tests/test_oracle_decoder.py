"""Cross-check of the numpy decoder oracle against an INDEPENDENT torch-CPU formulation
(torch ops + autograd) -- SURVEY §8c.  The TF graph arithmetic itself cannot be run
here (parity unpinned), so this pins the oracle's internal consistency: forward values,
analytic gradients, known-answer parameter count, beam/greedy invariants.  CPU only."""
import math

import numpy as np
import pytest
import torch
import torch.nn.functional as F

from oracle import beam_ref, decoder_ref as dr


def small_cfg(**kw):
    base = dict(rnn_size=32, rnn_word_size=16, attn_num_heads=4, softmax_size=18, fm_channels=24,
                im_embed_size=24, radix_base=16, start_id=16, end_id=17)
    base.update(kw)
    return dr.DecoderConfig(**base)


def make_batch(cfg, B=5, M=7, L=9, seed=0, dtype=np.float64):
    rng = np.random.default_rng(seed)
    fm = rng.standard_normal((B, M, cfg.fm_channels)).astype(dtype)
    im = rng.standard_normal((B, cfg.im_embed_size)).astype(dtype)
    caps = np.full((B, L), -1, np.int64)
    for b in range(B):
        n = rng.integers(2, L - 1) if b else L - 2          # row 0 has the max length
        caps[b, 0] = cfg.start_id
        caps[b, 1:1 + n] = rng.integers(0, cfg.softmax_size - 2, n)
        caps[b, 1 + n] = cfg.end_id
    if L > 6:
        caps[1, 4:] = -1            # a short row: exercises impute_finished / frozen state
    return fm, im, caps


from oracle.torch_ref import torch_forward      # the independent torch formulation (also timed by bench.py)


VARIANTS = [
    dict(),
    dict(cnn_fm_projection='independent', attn_probability_fn='sigmoid'),
    dict(cnn_fm_projection=None, attn_num_heads=1, token_type='word', rnn_init_method='project_hidden'),
    dict(cnn_fm_projection=None, attn_context_layer=True, attn_alignment_method='dot'),
    dict(attn_alignment_method='dot', attn_num_heads=2),
    dict(rnn_name='LN_LSTM'),
    dict(rnn_name='LN_LSTM', rnn_init_method='project_hidden', cnn_fm_projection='independent'),
    dict(rnn_name='GRU'),
    dict(rnn_name='GRU', rnn_init_method='project_hidden', attn_alignment_method='dot', attn_num_heads=2),
]


@pytest.mark.parametrize('kw', VARIANTS)
@pytest.mark.parametrize('use_dropout', [False, True])
def test_forward_and_grads_match_torch_autograd(kw, use_dropout):
    cfg = small_cfg(**kw)
    p = dr.init_params(cfg, seed=3, dtype=np.float64)
    rng = np.random.default_rng(5)
    for k in p:
        if k in ('b', 'b_o', 'ln_b', 'b_c') or (k.startswith('cln_') and k.endswith('b')):
            p[k] = p[k] + 0.1 * rng.standard_normal(p[k].shape)
        if k == 'ln_g' or (k.startswith('cln_') and k.endswith('g')):
            p[k] = 1 + 0.1 * rng.standard_normal(p[k].shape)
    fm, im, caps = make_batch(cfg)
    _, _, _, lens = dr.process_inputs(caps, cfg.token_type)
    masks = dr.make_dropout_masks(cfg, fm.shape[0], int(lens.max()), fm.shape[1], 11, np.float64) \
        if use_dropout else None
    out = dr.train_forward(p, cfg, fm, im, caps, masks)
    grads, dfm, dim = dr.train_backward(p, cfg, out)
    xe, ml, l2, logits, amap, tp, fm_t, im_t = torch_forward(p, cfg, fm, im, caps, masks)
    np.testing.assert_allclose(out['logits'], logits.detach().numpy(), rtol=1e-9, atol=1e-10)
    np.testing.assert_allclose(out['attn_maps'], amap.detach().numpy(), rtol=1e-9, atol=1e-12)
    np.testing.assert_allclose(out['xe'], xe.item(), rtol=1e-10)
    np.testing.assert_allclose(out['map_loss'], ml.item(), rtol=1e-10)
    np.testing.assert_allclose(dr.l2_loss(p, cfg.l2_decay), l2.item(), rtol=1e-10)
    (xe + ml + l2).backward()
    for k in p:
        np.testing.assert_allclose(grads[k], tp[k].grad.numpy(), rtol=1e-7, atol=1e-10, err_msg=k)
    np.testing.assert_allclose(dfm, fm_t.grad.numpy(), rtol=1e-7, atol=1e-10)
    np.testing.assert_allclose(dim, im_t.grad.numpy(), rtol=1e-7, atol=1e-10)


def test_scst_reward_weighted_loss_grads():
    cfg = small_cfg()
    p = dr.init_params(cfg, seed=1, dtype=np.float64)
    fm, im, caps = make_batch(cfg, seed=2)
    rewards = np.random.default_rng(0).standard_normal(fm.shape[0])
    out = dr.train_forward(p, cfg, fm, im, caps, None, rewards)
    grads, dfm, _ = dr.train_backward(p, cfg, out)
    xe, ml, l2, *_rest, tp, fm_t, im_t = torch_forward(p, cfg, fm, im, caps, None, rewards)
    np.testing.assert_allclose(out['xe'], xe.item(), rtol=1e-10)
    (xe + ml + l2).backward()
    for k in p:
        np.testing.assert_allclose(grads[k], tp[k].grad.numpy(), rtol=1e-7, atol=1e-10, err_msg=k)


def test_fp32_oracle_close_to_fp64():
    cfg = small_cfg()
    p64 = dr.init_params(cfg, seed=3, dtype=np.float64)
    fm, im, caps = make_batch(cfg)
    o64 = dr.train_forward(p64, cfg, fm, im, caps)
    g64, _, _ = dr.train_backward(p64, cfg, o64)
    p32 = {k: v.astype(np.float32) for k, v in p64.items()}
    o32 = dr.train_forward(p32, cfg, fm.astype(np.float32), im.astype(np.float32), caps)
    g32, _, _ = dr.train_backward(p32, cfg, o32)
    assert o32['logits'].dtype == np.float32
    np.testing.assert_allclose(o32['logits'], o64['logits'], rtol=2e-4, atol=2e-5)
    for k in g64:
        scale = np.abs(g64[k]).max() + 1e-12
        assert np.abs(g32[k] - g64[k]).max() / scale < 1e-3, k


def test_known_answer_param_count():
    """COMIC-256 + Inception-V1 decoder = 4 297 987 params (README.md:222 '4.3 M';
    SURVEY §8 a-P)."""
    cfg = dr.DecoderConfig(fm_channels=832, im_embed_size=1024)
    assert dr.count_params(dr.init_params(cfg)) == 4297987
    cfg3 = dr.DecoderConfig()                       # InceptionV3 feature map
    assert dr.count_params(dr.init_params(cfg3)) == 5707011


def test_readme_decoder_sizes_are_reproduced():
    """README.md:219-233 lists four decoder sizes for Inception-V1 / Mixed_4f (C = 832, C_g = 1024; parameters of scope
    Model/decoder/rnn_decoder, train_fn.py:83): COMIC-256 4.3 M (default) and 4.0 M (legacy: project_hidden init, the
    LN_tanh + im_embed head sits under Model/encoder and is not counted), word baseline (word tokens, 1 head, no
    feature-map projection: README.md:92-104) 12.7 M (default) and 12.2 M (legacy).  The two baseline figures pin the
    MS-COCO vocabulary to 9958..9970 softmax classes; every one of them reproduces both."""
    def count(**kw):
        return dr.count_params(dr.init_params(dr.DecoderConfig(fm_channels=832, im_embed_size=1024, **kw)))
    assert count() == 4297987 and round(count() / 1e6, 1) == 4.3
    legacy_comic = count(rnn_init_method='project_hidden')
    assert legacy_comic == 4297987 - 1024 * (256 + 512) + 1024 * 512 == 4035843 and round(legacy_comic / 1e6, 1) == 4.0
    base = dict(token_type='word', attn_num_heads=1, cnn_fm_projection=None)
    for V in (9958, 9962, 9970):
        n = count(softmax_size=V, start_id=V - 2, end_id=V - 1, **base)
        assert n == 5082625 + 769 * V and round(n / 1e6, 1) == 12.7
        n_leg = count(softmax_size=V, start_id=V - 2, end_id=V - 1, rnn_init_method='project_hidden', **base)
        assert n_leg == n - 1024 * (256 + 832) + 1024 * 512 and round(n_leg / 1e6, 1) == 12.2


def test_process_inputs_masks():
    caps = np.array([[256, 3, 4, 257, -1, -1], [256, 9, 8, 7, 6, 257]])
    inp, tgt, m, lens = dr.process_inputs(caps, 'radix')
    assert inp.tolist() == [[256, 3, 4, 257, -1], [256, 9, 8, 7, 6]]
    assert tgt.tolist() == [[3, 4, 257, 0, 0], [9, 8, 7, 6, 257]]
    assert m.tolist() == [[1, 1, 1, 0, 0], [1, 1, 1, 1, 1]] and lens.tolist() == [3, 5]
    inp_w, _, _, _ = dr.process_inputs(caps, 'word')
    assert inp_w.min() == 0


def test_adam_tf_formula_and_cosine_lr():
    rng = np.random.default_rng(0)
    w = rng.standard_normal(100).astype(np.float32); g = rng.standard_normal(100).astype(np.float32)
    m = np.zeros_like(w); v = np.zeros_like(w)
    w0 = w.copy()
    dr.adam_tf_update(w, g, m, v, 1, 1e-2, eps=1e-2)
    lr_t = 1e-2 * math.sqrt(1 - 0.999) / (1 - 0.9)
    ref = w0 - lr_t * (0.1 * g) / (np.sqrt(0.001 * g * g) + 1e-2)
    np.testing.assert_allclose(w, ref, rtol=1e-5)
    # differs from torch.optim.Adam (eps inside the bias-corrected sqrt)
    assert np.isclose(dr.cosine_lr(0, 100, 1e-2, 1e-5), 1e-2)
    assert np.isclose(dr.cosine_lr(100, 100, 1e-2, 1e-5), 1e-5)
    assert np.isclose(dr.cosine_lr(250, 100, 1e-2, 1e-5), 1e-5)
    assert np.isclose(dr.cosine_lr(50, 100, 1e-2, 1e-5), (1e-2 - 1e-5) / 2 + 1e-5)


# ------------------------------------------------------------------ decoding --
def test_gather_tree_known_answer():
    # T=3, B=1, W=2.  beam0: t0 tok 5, t1 tok 6 (parent 0), t2 tok 9=EOS (parent 1)
    step = np.array([[[5, 7]], [[6, 8]], [[9, 4]]], np.int32)
    par = np.array([[[0, 0]], [[0, 1]], [[1, 0]]], np.int32)
    out = beam_ref.gather_tree(step, par, np.array([3]), 9)
    assert out[:, 0, 0].tolist() == [7, 8, 9]
    assert out[:, 0, 1].tolist() == [5, 6, 4]
    out2 = beam_ref.gather_tree(step, par, np.array([2]), 9)       # max_len 2: t2 stays EOS-filled
    assert out2[:, 0, 0].tolist() == [5, 6, 9]
    # everything after the first EOS becomes EOS
    step3 = np.array([[[9, 1]], [[2, 3]], [[4, 5]]], np.int32)
    par3 = np.zeros((3, 1, 2), np.int32)
    assert beam_ref.gather_tree(step3, par3, np.array([3]), 9)[:, 0, 0].tolist() == [9, 9, 9]


def test_beam1_equals_greedy_until_eos_and_beam_invariants():
    cfg = small_cfg()
    p = dr.init_params(cfg, seed=4, dtype=np.float32)
    p['b_o'][cfg.end_id] = 1.0          # make EOS reachable
    fm, im, _ = make_batch(cfg, B=4, dtype=np.float32)
    gids, glog, gmap = beam_ref.greedy_decode(p, cfg, fm, im, 12)
    pred, scores, hist, dbg = beam_ref.beam_search_decode(p, cfg, fm, im, 3, 12, return_debug=True)
    T, B, W = pred.shape
    assert scores.shape == (T, B, W) and hist.shape == (T, B * W, cfg.attn_num_heads * fm.shape[1])
    # beams sorted best-first at every step
    assert (np.diff(scores, axis=2) <= 1e-6).all()
    # after the first EOS a beam holds only EOS
    for b in range(B):
        for w in range(W):
            row = pred[:, b, w].tolist()
            if cfg.end_id in row:
                k = row.index(cfg.end_id)
                assert all(x == cfg.end_id for x in row[k:])
    # beam width 1 reproduces greedy ids up to (and including) each row's first EOS
    p1, _, _ = beam_ref.beam_search_decode(p, cfg, fm, im, 1, 12)
    for b in range(B):
        g = gids[b].tolist()
        n = g.index(cfg.end_id) + 1 if cfg.end_id in g else len(g)
        n = min(n, p1.shape[0])
        assert p1[:n, b, 0].tolist() == g[:n]
    ids, sc, amap = beam_ref.post_process_beam(pred, scores, hist, cfg, 3, top_beam=True)
    assert ids.shape == (B, T) and amap.shape == (B, cfg.attn_num_heads, T, fm.shape[1])
    np.testing.assert_allclose(amap.sum(-1), 1.0, rtol=1e-5)


def test_legacy_head_oracle_vs_torch_autograd():
    """Legacy encoder head (model_base.py:80-91): LN (eps 1e-12) + tanh + linear without bias."""
    rng = np.random.default_rng(3)
    B, C_, N = 5, 48, 20
    p = dict(ln_gamma=rng.uniform(0.5, 1.5, C_), ln_beta=0.1 * rng.standard_normal(C_), W=rng.standard_normal((C_, N)) / 7)
    net = rng.standard_normal((B, C_)) * 2 + 0.3
    out, cache = dr.legacy_head_forward(p, net)
    tp = {k: torch.tensor(v, dtype=torch.float64, requires_grad=True) for k, v in p.items()}
    y = torch.tanh(F.layer_norm(torch.tensor(net), (C_,), tp['ln_gamma'], tp['ln_beta'], eps=1e-12)) @ tp['W']
    np.testing.assert_allclose(out, y.detach().numpy(), rtol=1e-6, atol=1e-6)
    d = rng.standard_normal((B, N))
    (y * torch.tensor(d)).sum().backward()
    g = dr.legacy_head_backward(p, cache, d)
    for k in g:
        np.testing.assert_allclose(g[k], tp[k].grad.numpy(), rtol=2e-5, atol=1e-6)


def test_momentum_oracle_formula():
    w, acc = np.ones(3, np.float32), np.zeros(3, np.float32)
    g = np.array([1, -2, 0.5], np.float32)
    dr.momentum_tf_update(w, g, acc, 0.1)
    dr.momentum_tf_update(w, g, acc, 0.1)
    np.testing.assert_allclose(acc, 1.9 * g, rtol=1e-6)                 # 0.9*g + g
    np.testing.assert_allclose(w, 1 - 0.1 * g - 0.1 * 1.9 * g, rtol=1e-6)


def test_sampled_decode_draws_follow_the_softmax():
    """SampleEmbeddingHelper restated as argmax(logits + Gumbel noise) (beam_ref.greedy_decode(gumbel=...)): over many
    noise draws the first token's empirical distribution is softmax(logits) [Gumbel-max], and zero noise is greedy."""
    cfg = dr.DecoderConfig(rnn_size=16, rnn_word_size=8, attn_num_heads=2, radix_base=6, softmax_size=8, fm_channels=16,
                           im_embed_size=16, start_id=6, end_id=7)
    p = dr.init_params(cfg, 3)
    rng = np.random.default_rng(0)
    fm = rng.standard_normal((1, 4, cfg.fm_channels)).astype(np.float32)
    im = rng.standard_normal((1, cfg.im_embed_size)).astype(np.float32)
    V = p['b_o'].shape[0]
    _, lg, _ = beam_ref.greedy_decode(p, cfg, fm, im, 1)
    prob = dr.softmax(lg[0, 0].astype(np.float64))
    n = 4000
    counts = np.zeros(V)
    for i in range(n):
        g = (-np.log(-np.log(rng.uniform(1e-12, 1.0, (1, 1, V))))).astype(np.float32)
        ids, _, _ = beam_ref.greedy_decode(p, cfg, fm, im, 1, gumbel=g)
        counts[ids[0, 0]] += 1
    assert np.abs(counts / n - prob).max() < 4.0 * np.sqrt(0.25 / n)
    z_ids, _, _ = beam_ref.greedy_decode(p, cfg, fm, im, 3, gumbel=np.zeros((3, 1, V), np.float32))
    g_ids, _, _ = beam_ref.greedy_decode(p, cfg, fm, im, 3)
    np.testing.assert_array_equal(z_ids, g_ids)
