"""Path-level parity on the GPU: whole InceptionV3 forward, one decoder training step
(forward + backward), greedy and beam decoding -- HIP path vs the CPU oracle on identical
seeded inputs, through the C-ABI executors."""
import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu

from comic_amd import decoder as cdec, nets
from oracle import beam_ref, cnn_ref, decoder_ref as dr
from tests.gpu_util import DEV, F32_RTOL, assert_close, dev, rel_err, sync


# ----------------------------------------------------------------------------- CNN --------
@pytest.fixture(scope='module')
def cnn_params():
    return cnn_ref.randomize_bn(cnn_ref.init_params(0, 224), seed=1)


@pytest.mark.parametrize('dtype,tol', [('f32', 1e-3), ('bf16', 3e-2)])
def test_inception_v3_forward_224(cnn_params, dtype, tol):
    B = 2
    x = np.random.default_rng(48964896).uniform(-1, 1, (B, 224, 224, 3)).astype(np.float32)
    plan = nets.CnnPlan('inception_v3', (224, 224))
    enc = nets.CnnEncoder(plan, cnn_params, B, dtype, DEV)
    im, fm = enc.forward(dev(x))
    sync()
    net_ref, ep = cnn_ref.inception_v3(cnn_params, x, act_dtype=dtype)
    assert fm.shape == (B, 25, 2048) and im.shape == (B, 2048)      # 5x5x2048 at 224
    for name in ('Conv2d_1a_3x3', 'Conv2d_2b_3x3', 'MaxPool_5a_3x3', 'Mixed_5b', 'Mixed_5d', 'Mixed_6a',
                 'Mixed_6e', 'Mixed_7a', 'Mixed_7b'):
        assert_close(enc.end_point(name).float().cpu().numpy(), ep[name], tol, '%s %s' % (name, dtype))
    assert_close(fm.cpu().numpy().reshape(B, 5, 5, 2048), ep['Mixed_7c'], tol, 'Mixed_7c ' + dtype)
    assert_close(im.cpu().numpy(), net_ref.reshape(B, -1), tol, 'im_embed ' + dtype)


def test_inception_v3_forward_299_f32():
    """The north-star's 8x8x2048 feature map needs 299x299 inputs."""
    params = cnn_ref.randomize_bn(cnn_ref.init_params(3, 299), seed=4)
    x = np.random.default_rng(5).uniform(-1, 1, (1, 299, 299, 3)).astype(np.float32)
    enc = nets.CnnEncoder(nets.CnnPlan('inception_v3', (299, 299)), params, 1, 'f32', DEV)
    im, fm = enc.forward(dev(x))
    net_ref, ep = cnn_ref.inception_v3(params, x)
    assert fm.shape == (1, 64, 2048)
    assert_close(fm.cpu().numpy().reshape(1, 8, 8, 2048), ep['Mixed_7c'], 1e-3, 'Mixed_7c@299')
    assert_close(im.cpu().numpy(), net_ref.reshape(1, -1), 1e-3, 'im_embed@299')


def test_cnn_batch_independence_full_batch():
    """Size-independent property at the bench batch (64, bf16): an image's features do not
    depend on its position in the batch or on its neighbours (BN is frozen)."""
    params = cnn_ref.init_params(0, 224)
    plan = nets.CnnPlan('inception_v3', (224, 224))
    rng = np.random.default_rng(0)
    x = rng.uniform(-1, 1, (64, 224, 224, 3)).astype(np.float32)
    enc = nets.CnnEncoder(plan, params, 64, 'bf16', DEV)
    im1, fm1 = (t.clone() for t in enc.forward(dev(x)))
    perm = rng.permutation(64)
    im2, fm2 = enc.forward(dev(x[perm]))
    assert torch.equal(fm1[perm], fm2) and torch.equal(im1[perm], im2)
    assert torch.isfinite(fm1).all()


def test_grouped_branch_launch_is_bit_identical():
    """comic_cnn_forward_grouped (one launch per depth of an Inception block) against the
    op-by-op executor: every end point bit-identical, at a ragged batch (tile tails) and
    with autotuned tiles."""
    params = cnn_ref.randomize_bn(cnn_ref.init_params(0, 224), seed=2)
    x = np.random.default_rng(7).uniform(-1, 1, (5, 224, 224, 3)).astype(np.float32)
    single = nets.CnnEncoder(nets.CnnPlan('inception_v3', (224, 224), group_branches=False), params, 5, 'bf16', DEV)
    grouped = nets.CnnEncoder(nets.CnnPlan('inception_v3', (224, 224), group_branches=True), params, 5, 'bf16', DEV)
    assert grouped._group_args is not None and single._group_args is None
    im0, fm0 = (t.clone() for t in single.forward(dev(x)))
    for tuned in (False, True):
        if tuned:
            grouped.autotune(reps=2)
        im1, fm1 = grouped.forward(dev(x))
        for name in single.plan.end_points:
            assert torch.equal(single.end_point(name), grouped.end_point(name)), name
        assert torch.equal(fm0, fm1) and torch.equal(im0, im1)
        im2, fm2 = grouped.forward(dev(x), use_graph=True)      # second call captures, third replays
        im2, fm2 = grouped.forward(dev(x), use_graph=True)
        assert torch.equal(fm0, fm2) and torch.equal(im0, im2)


# ----------------------------------------------------------------------------- decoder ----
def _spec_and_cfg(**kw):
    base = dict(D=128, E=64, V=258, C=192, Cg=192, H=8, M=25)
    base.update(kw)
    spec = cdec.DecoderSpec(**base)
    cfg = dr.DecoderConfig(rnn_size=spec.D, rnn_word_size=spec.E, attn_num_heads=spec.H,
                           cnn_fm_projection=spec.fm_projection, attn_alignment_method=spec.method,
                           attn_probability_fn=spec.prob, attn_context_layer=spec.context_layer,
                           rnn_init_method=spec.init_method, token_type=spec.token_type, softmax_size=spec.V,
                           fm_channels=spec.C, im_embed_size=spec.Cg, start_id=spec.start_id, end_id=spec.end_id)
    return spec, cfg


def _batch(spec, B, L, seed):
    rng = np.random.default_rng(seed)
    fm = rng.standard_normal((B, spec.M, spec.C)).astype(np.float32)
    im = rng.standard_normal((B, spec.Cg)).astype(np.float32)
    caps = np.full((B, L), -1, np.int64)
    for b in range(B):
        n = L - 2 if b == 0 else int(rng.integers(1, L - 1))
        caps[b, 0] = spec.start_id
        caps[b, 1:1 + n] = rng.integers(0, min(spec.V - 2, 256), n)
        caps[b, 1 + n] = spec.end_id
    return fm, im, caps


def _rand_params(cfg, seed):
    p = dr.init_params(cfg, seed)
    rng = np.random.default_rng(seed + 100)
    for k in ('b', 'b_o', 'ln_b'):
        if k in p:
            p[k] = (0.1 * rng.standard_normal(p[k].shape)).astype(np.float32)
    if 'ln_g' in p:
        p['ln_g'] = (1 + 0.1 * rng.standard_normal(p['ln_g'].shape)).astype(np.float32)
    return p


TRAIN_VARIANTS = [
    dict(),                                                                    # COMIC: radix, tied, add_LN, 8 heads
    dict(D=512, E=256, C=2048, Cg=2048),                                       # full COMIC-256 geometry
    dict(fm_projection='independent', prob='sigmoid'),
    dict(fm_projection=None, H=1, token_type='word', V=300, init_method='project_hidden',
         start_id=298, end_id=299),                                            # InstaPIC-style baseline (config 5)
    dict(fm_projection=None, context_layer=True, method='dot', H=4),
    dict(method='dot', H=2, M=64),
]


@pytest.mark.parametrize('kw', TRAIN_VARIANTS)
@pytest.mark.parametrize('use_dropout', [False, True])
@pytest.mark.parametrize('scst', [False, True])
def test_decoder_train_step_matches_oracle(kw, use_dropout, scst):
    if scst and use_dropout and kw:
        pytest.skip('covered by the other combinations')
    spec, cfg = _spec_and_cfg(**kw)
    B, Lc = 6, 11
    p = _rand_params(cfg, 3)
    fm, im, caps = _batch(spec, B, Lc, 7)
    _, _, _, lens = dr.process_inputs(caps, cfg.token_type)
    masks = dr.make_dropout_masks(cfg, B, int(lens.max()), spec.M, 11) if use_dropout else None
    rewards = np.random.default_rng(9).standard_normal(B).astype(np.float32) if scst else None
    cfg.l2_decay = 0.0                         # L2 is applied inside the fused Adam kernel
    out = dr.train_forward(p, cfg, fm, im, caps, masks, rewards)
    grads, dfm, dim = dr.train_backward(p, cfg, out)
    dec = cdec.Decoder(spec, p, DEV)
    res = dec.train_step(dev(fm), dev(im), caps, masks=masks, rewards=rewards, training=use_dropout,
                         want_input_grads=True)
    sync()
    assert_close(res['logits'].cpu().numpy(), out['logits'], F32_RTOL, 'logits')
    assert_close(res['attn_maps'].cpu().numpy(), out['attn_maps'], F32_RTOL, 'attn_maps')
    assert abs(float(res['loss']) - float(out['xe'])) <= F32_RTOL * abs(float(out['xe'])) + 1e-6
    assert abs(float(res['map_loss']) - float(out['map_loss'])) <= F32_RTOL * abs(float(out['map_loss'])) + 1e-7
    live = np.arange(out['ids'].shape[1])[None, :] < lens[:, None]
    np.testing.assert_array_equal(res['ids'].cpu().numpy()[live], out['ids'][live])
    g = dec.grads.to_numpy()
    for k in grads:
        assert_close(g[k], grads[k], F32_RTOL, 'grad ' + k)
    assert_close(res['dfm'].cpu().numpy(), dfm, F32_RTOL, 'dfm')
    assert_close(res['dim_embed'].cpu().numpy(), dim, F32_RTOL, 'dim_embed')


def test_known_answer_param_count_on_device():
    """README.md:222 '4.3 M' -> 4 297 987 (SURVEY §8 a-P); InceptionV3 variant 5 707 011."""
    s1 = cdec.DecoderSpec(C=832, Cg=1024, M=196)
    s3 = cdec.DecoderSpec()
    assert sum(int(np.prod(v)) if v else 1 for v in s1.param_shapes().values()) == 4297987
    assert sum(int(np.prod(v)) if v else 1 for v in s3.param_shapes().values()) == 5707011


@pytest.mark.parametrize('kw', [dict(), dict(fm_projection=None, H=1, token_type='word', V=300,
                                            init_method='project_hidden', start_id=298, end_id=299)])
def test_greedy_and_beam_match_oracle(kw):
    spec, cfg = _spec_and_cfg(**kw)
    p = _rand_params(cfg, 5)
    p['b_o'][spec.end_id] = 1.5                 # make EOS reachable with random weights
    B = 5
    fm, im, _ = _batch(spec, B, 6, 21)
    dec = cdec.Decoder(spec, p, DEV)
    max_steps = 14
    g_ids, g_logits, g_map = beam_ref.greedy_decode(p, cfg, fm, im, max_steps)
    ids, amap, logits = dec.greedy(dev(fm), dev(im), max_steps, want_logits=True)
    np.testing.assert_array_equal(ids, g_ids)                                  # bit-exact argmax ids
    assert_close(logits.cpu().numpy(), g_logits, F32_RTOL, 'greedy logits')
    assert_close(amap.cpu().numpy(), g_map, F32_RTOL, 'greedy attention maps')
    for W in (3, 7):
        pred, scores, hist, dbg = beam_ref.beam_search_decode(p, cfg, fm, im, W, max_steps, return_debug=True)
        res = dec.beam_search(dev(fm), dev(im), W, max_steps)
        np.testing.assert_array_equal(res['step_ids'], dbg['step_ids'])
        np.testing.assert_array_equal(res['parent_ids'], dbg['parent_ids'])
        np.testing.assert_array_equal(res['predicted_ids'], pred)              # bit-exact beam ids
        np.testing.assert_array_equal(res['lengths'], dbg['lengths'])
        fin = np.isfinite(scores)
        assert_close(np.where(fin, res['scores'], 0), np.where(fin, scores, 0), 1e-4, 'beam scores')
        assert_close(res['attn_hist'], hist, F32_RTOL, 'beam alignment history')


def test_train_step_full_batch_properties():
    """Full bench geometry (B=64, COMIC-256 on a 5x5x2048 map, T=29): (i) deterministic
    across two runs, (ii) rows are independent -- the per-row losses of a 64-batch equal
    those of its two 32-halves, (iii) all gradients finite."""
    spec, cfg = _spec_and_cfg(D=512, E=256, C=2048, Cg=2048)
    fm, im, caps = _batch(spec, 64, 30, 1)
    dec = cdec.Decoder(spec, _rand_params(cfg, 1), DEV)
    r1 = dec.train_step(dev(fm), dev(im), caps, training=False)
    g1 = dec.grads.data.clone()
    lg1 = r1['logits'].clone()
    r2 = dec.train_step(dev(fm), dev(im), caps, training=False)
    assert torch.equal(g1, dec.grads.data) and torch.equal(lg1, r2['logits'])
    assert torch.isfinite(g1).all()
    ra = dec.train_step(dev(fm[:32]), dev(im[:32]), caps[:32], training=False)
    Ta = ra['logits'].shape[1]
    assert_close(ra['logits'].cpu().numpy(), lg1[:32, :Ta].cpu().numpy(), 1e-5, 'half-batch logits')
