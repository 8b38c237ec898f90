"""InceptionV3 oracle vs the pins the reference's own tests hold (shapes, end-point
names, parameter count) and vs an independent torch-CPU formulation.  CPU only."""
import numpy as np
import torch
import torch.nn.functional as F

from oracle import cnn_ref


def test_param_count_and_conv_count():
    log, macs, params = cnn_ref.describe(224)
    assert len(log) == 94
    # inception_v3_test.py:125-133 -> 21 802 784 model variables in inception_v3_base
    assert sum(v.size for v in params.values()) == 21802784
    # 2 836 MMAC @224, 5 711 MMAC @299
    assert round(macs / 1e6) == 2836
    assert round(cnn_ref.describe(299)[1] / 1e6) == 5711
    # no BN gamma by default (inception_v3_test.py:321-328)
    assert not any('gamma' in k for k in params)


def test_endpoint_shapes_299():
    """inception_v3_test.py:93-123 (batch 1 here)."""
    net = cnn_ref._Net(None, np.random.default_rng(0), run=False)
    pooled, ep = cnn_ref._run(net, np.zeros((1, 299, 299, 3), np.float32))
    expected = {'Conv2d_1a_3x3': (149, 149, 32), 'Conv2d_2a_3x3': (147, 147, 32),
                'Conv2d_2b_3x3': (147, 147, 64), 'MaxPool_3a_3x3': (73, 73, 64),
                'Conv2d_3b_1x1': (73, 73, 80), 'Conv2d_4a_3x3': (71, 71, 192),
                'MaxPool_5a_3x3': (35, 35, 192), 'Mixed_5b': (35, 35, 256),
                'Mixed_5c': (35, 35, 288), 'Mixed_5d': (35, 35, 288), 'Mixed_6a': (17, 17, 768),
                'Mixed_6b': (17, 17, 768), 'Mixed_6c': (17, 17, 768), 'Mixed_6d': (17, 17, 768),
                'Mixed_6e': (17, 17, 768), 'Mixed_7a': (8, 8, 1280), 'Mixed_7b': (8, 8, 2048),
                'Mixed_7c': (8, 8, 2048)}
    for k, s in expected.items():
        assert ep[k].shape[1:] == s, k
    assert list(ep.keys())[:18] == list(expected.keys())          # inception_v3_test.py:75-91
    assert pooled.shape == (1, 1, 1, 2048)                        # inception_v3_test.py:46-56


def test_half_size_images_150():
    """inception_v3_test.py:210-222 (testHalfSizeImages): 150 x 150 inputs give a 3 x 3 x 2048 Mixed_7c; the product's
    plan builder agrees (and with the 224 / 299 maps of SURVEY section 0)."""
    net = cnn_ref._Net(None, np.random.default_rng(0), run=False)
    pooled, ep = cnn_ref._run(net, np.zeros((5, 150, 150, 3), np.float32))
    assert ep['Mixed_7c'].shape == (5, 3, 3, 2048) and pooled.shape == (5, 1, 1, 2048)
    from comic_amd import nets
    for size, fm in ((150, (3, 3, 2048)), (224, (5, 5, 2048)), (299, (8, 8, 2048))):
        assert nets.CnnPlan('inception_v3', (size, size)).fm_dims() == fm
        assert nets.CnnPlan('inception_v3', (size, size), pool_after_projection=True, fuse_pools=True).fm_dims() == fm


def test_primitives_vs_torch():
    rng = np.random.default_rng(0)
    x = rng.standard_normal((2, 13, 11, 8)).astype(np.float32)
    xt = torch.tensor(x).permute(0, 3, 1, 2)
    for (kh, kw), s, pad in [((3, 3), 2, 'VALID'), ((3, 3), 1, 'SAME'), ((1, 7), 1, 'SAME'),
                             ((7, 1), 1, 'SAME'), ((5, 5), 1, 'SAME'), ((1, 1), 1, 'SAME'),
                             ((7, 7), 2, 'SAME')]:
        w = rng.standard_normal((kh, kw, 8, 6)).astype(np.float32)
        y = cnn_ref.conv2d(x, w, s, pad)
        wt = torch.tensor(w).permute(3, 2, 0, 1)
        if pad == 'SAME':
            _, pt, pb = cnn_ref.same_pad(13, kh, s)
            _, pl, pr = cnn_ref.same_pad(11, kw, s)
            xin = F.pad(xt, (pl, pr, pt, pb))
        else:
            xin = xt
        yt = F.conv2d(xin, wt, stride=s).permute(0, 2, 3, 1).numpy()
        np.testing.assert_allclose(y, yt, rtol=1e-4, atol=1e-4)
    mp = cnn_ref.max_pool(x, 3, 2, 'VALID')
    np.testing.assert_array_equal(mp, F.max_pool2d(xt, 3, 2).permute(0, 2, 3, 1).numpy())
    ap = cnn_ref.avg_pool(x, 3, 1, 'SAME')
    apt = F.avg_pool2d(xt, 3, 1, 1, count_include_pad=False).permute(0, 2, 3, 1).numpy()
    np.testing.assert_allclose(ap, apt, rtol=1e-5, atol=1e-6)
    # asymmetric SAME padding of V1's 7x7 s2 conv at 224: 2 before, 3 after
    assert cnn_ref.same_pad(224, 7, 2) == (112, 2, 3)


def test_bf16_round():
    x = np.array([1.0, 1.00390625, 1.005859375, -3.1415927, 0.0, 65504.0], np.float32)
    r = cnn_ref.bf16_round(x)
    t = torch.tensor(x).to(torch.bfloat16).float().numpy()
    np.testing.assert_array_equal(r, t)


def test_forward_small_image_finite_and_deterministic():
    params = cnn_ref.randomize_bn(cnn_ref.init_params(0, 139))
    x = np.random.default_rng(1).uniform(-1, 1, (1, 139, 139, 3)).astype(np.float32)
    im, fm = cnn_ref.encoder(params, x)
    assert im.shape == (1, 2048) and fm.shape == (1, 9, 2048)     # 139 -> 3x3 map
    assert np.isfinite(fm).all()
    im2, fm2 = cnn_ref.encoder(params, x, act_dtype='bf16')
    rel = np.abs(fm2 - fm).max() / np.abs(fm).max()
    assert rel < 0.1


def test_inception_v1_known_answers():
    """inception_v1_test.py: end-point shapes at 224 (:91-122) and 5 607 184 model variables (:124-132);
    MACs per image 1.497 G (SURVEY §8 a1')."""
    p = cnn_ref.init_params_v1(0, 224)
    assert sum(v.size for v in p.values()) == 5607184
    x = np.random.default_rng(0).uniform(-1, 1, (1, 224, 224, 3)).astype(np.float32)
    net, ep = cnn_ref.inception_v1(p, x)
    want = {'Conv2d_1a_7x7': (112, 112, 64), 'MaxPool_2a_3x3': (56, 56, 64), 'Conv2d_2b_1x1': (56, 56, 64),
            'Conv2d_2c_3x3': (56, 56, 192), 'MaxPool_3a_3x3': (28, 28, 192), 'Mixed_3b': (28, 28, 256),
            'Mixed_3c': (28, 28, 480), 'MaxPool_4a_3x3': (14, 14, 480), 'Mixed_4b': (14, 14, 512),
            'Mixed_4c': (14, 14, 512), 'Mixed_4d': (14, 14, 512), 'Mixed_4e': (14, 14, 528),
            'Mixed_4f': (14, 14, 832), 'MaxPool_5a_2x2': (7, 7, 832), 'Mixed_5b': (7, 7, 832),
            'Mixed_5c': (7, 7, 1024)}
    for k, shp in want.items():
        assert ep[k].shape == (1,) + shp, k
    assert net.shape == (1, 1, 1, 1024) and np.isfinite(net).all()
    n = cnn_ref._Net(None, np.random.default_rng(0), run=False)
    cnn_ref._run_v1(n, np.zeros((1, 224, 224, 3), np.float32))
    assert n.macs == 1497352192 and len(n.conv_log) == 57


def test_whole_network_vs_torch_formulation():
    """oracle.torch_ref walks the same layer table with torch NCHW convolutions (oneDNN) and autograd: the whole
    InceptionV3 forward and the conv-weight / BN-beta gradients of the numpy oracle agree with it (small image: every
    layer type, SAME / VALID, stride 2, the asymmetric-free V3 paddings)."""
    import torch
    from oracle import torch_ref
    p = cnn_ref.randomize_bn(cnn_ref.init_params(0, 107), 2)
    x = np.random.default_rng(5).uniform(-1, 1, (2, 107, 107, 3)).astype(np.float32)
    im, fm, net = torch_ref.torch_encoder(p, x)
    im_ref, fm_ref = cnn_ref.encoder(p, x)
    assert np.abs(fm.detach().numpy() - fm_ref).max() <= 2e-5 * np.abs(fm_ref).max()
    assert np.abs(im.detach().numpy() - im_ref).max() <= 2e-5 * np.abs(im_ref).max()
    rng = np.random.default_rng(6)
    d_im = rng.standard_normal(im_ref.shape).astype(np.float32)
    d_fm = rng.standard_normal(fm_ref.shape).astype(np.float32)
    (im * torch.from_numpy(d_im)).sum().add((fm * torch.from_numpy(d_fm)).sum()).backward()
    grads, _, _ = cnn_ref.inception_v3_grads(p, x, d_im, d_fm)
    worst = 0.0
    for wn, (w, scale, mean, beta) in net.tw.items():
        gw = w.grad.numpy().transpose(2, 3, 1, 0)                       # OIHW -> HWIO
        worst = max(worst, np.abs(gw - grads[wn]).max() / (np.abs(grads[wn]).max() + 1e-30))
        bn = wn.replace('weights', 'BatchNorm/beta')
        worst = max(worst, np.abs(beta.grad.numpy() - grads[bn]).max() / (np.abs(grads[bn]).max() + 1e-30))
    assert len(net.tw) == 94 and worst < 2e-3, worst
