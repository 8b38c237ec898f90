"""Path-level parity on the GPU: whole InceptionV3 forward, one decoder training step
(forward + backward), greedy and beam decoding -- HIP path vs the CPU oracle on identical
seeded inputs, through the C-ABI executors."""
import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu

import comic_amd._lib as L
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


def test_inception_v3_forward_224_bf16x3_meets_the_fp32_bar(cnn_params):
    """The bf16x3 plan (COMIC_OP_X3: activations as [hi | lo | hi] channel regions, filters [W_hi | W_hi | W_lo], the
    unchanged bf16 MFMA kernels) against the fp32 oracle at the tolerance of the exact-fp32 plan -- logits / gradients
    within 1e-3 relative is the north star's bar, which the plain bf16 plan misses (3e-2).  Grouped launches, autotuned
    tiles and graph replay give the same bits as eager launches on the default tiles."""
    B = 2
    x = np.random.default_rng(48964896).uniform(-1, 1, (B, 224, 224, 3)).astype(np.float32)
    plan = nets.CnnPlan('inception_v3', (224, 224), x3=True)
    assert all(o['flags'] & L.OP_X3 for o in plan.ops if o['kind'] in (0, 1, 2, 3))
    enc = nets.CnnEncoder(plan, cnn_params, B, 'bf16', DEV)
    # the filter copy (comic_cnn_pack_x3_weights, one launch): per tap [bf16(w) | bf16(w) | bf16(w - bf16(w))], zero padding
    for i in (1, 5, 40, 93):
        prefix, kh, kw, cin, cout, stem = plan.weights[i]
        cin_p, cout_p = plan.wphys[i]
        K = kh * kw * cin_p
        kpad, kpad3 = (K + 63) // 64 * 64, (3 * K + 63) // 64 * 64
        m = enc.w_master.view('w%d' % i).view(cout_p, kpad)[:, :K].reshape(cout_p, kh * kw, cin_p)
        hi = m.to(torch.bfloat16)
        lo = (m - hi.float()).to(torch.bfloat16)
        got = enc.w_plan[enc._x3_off[i]:enc._x3_off[i] + cout_p * kpad3].view(cout_p, kpad3)
        assert torch.equal(got[:, :3 * K], torch.cat([hi, hi, lo], dim=2).reshape(cout_p, 3 * K)), prefix
        assert not got[:, 3 * K:].any()
    im, fm = (t.clone() for t in enc.forward(dev(x)))
    sync()
    net_ref, ep = cnn_ref.inception_v3(cnn_params, x, act_dtype='f32')
    worst = 0.0
    for name in ('Conv2d_1a_3x3', 'Conv2d_2b_3x3', 'MaxPool_5a_3x3', 'Mixed_5b', 'Mixed_5d', 'Mixed_6a',
                 'Mixed_6e', 'Mixed_7a', 'Mixed_7b'):
        got = enc.end_point(name).float().cpu().numpy()
        worst = max(worst, rel_err(got, ep[name]))
        assert_close(got, ep[name], F32_RTOL, '%s bf16x3' % name)
    assert_close(fm.cpu().numpy().reshape(B, 5, 5, 2048), ep['Mixed_7c'], F32_RTOL, 'Mixed_7c bf16x3')
    assert_close(im.cpu().numpy(), net_ref.reshape(B, -1), F32_RTOL, 'im_embed bf16x3')
    print('bf16x3 worst end-point error %.2e, feature map %.2e' % (worst, rel_err(fm.cpu().numpy().reshape(B, 5, 5, 2048), ep['Mixed_7c'])))
    enc.autotune(reps=2)
    for _ in range(2):
        im2, fm2 = enc.forward(dev(x), use_graph=True)
    assert torch.equal(fm2, fm) and torch.equal(im2, im)
    # the frozen-CNN form of the plan: the pool branches as 1x1 projection (fp32, raw) -> 3x3 average + BN + ReLU (kind 7,
    # storing the three regions), their projections inside the blocks' 1x1 group launches -- same bar
    plan_p = nets.CnnPlan('inception_v3', (224, 224), x3=True, pool_after_projection=True)
    assert any(o['kind'] == 7 and o['flags'] & L.OP_X3 for o in plan_p.ops)
    enc_p = nets.CnnEncoder(plan_p, cnn_params, B, 'bf16x3', DEV)
    im_p, fm_p = enc_p.forward(dev(x))
    for name in ('Mixed_5b', 'Mixed_6e', 'Mixed_7b'):
        assert_close(enc_p.end_point(name).float().cpu().numpy(), ep[name], F32_RTOL, '%s bf16x3, pool after projection' % name)
    assert_close(fm_p.cpu().numpy().reshape(B, 5, 5, 2048), ep['Mixed_7c'], F32_RTOL, 'Mixed_7c bf16x3, pool after projection')
    assert_close(im_p.cpu().numpy(), net_ref.reshape(B, -1), F32_RTOL, 'im_embed bf16x3, pool after projection')


@pytest.mark.parametrize('dtype,tol', [('f32', 1e-3), ('bf16', 3e-2)])
def test_inception_v3_pool_after_projection(cnn_params, dtype, tol):
    """Forward-only rewrite of the pool branches (1x1 projection first, then the 3x3 average with the
    BN + ReLU epilogue, kind 7): same end points as the reference order, against the oracle and against
    the plain plan; it shares the variables of a plain-plan encoder and refuses cnn_finetune."""
    B = 3
    x = np.random.default_rng(11).uniform(-1, 1, (B, 224, 224, 3)).astype(np.float32)
    plain = nets.CnnEncoder(nets.CnnPlan('inception_v3', (224, 224)), cnn_params, B, dtype, DEV)
    # bf16: also with the pool ops riding in the grouped conv launches (comic_cnn_forward_grouped members of kind 7)
    plan = nets.CnnPlan('inception_v3', (224, 224), pool_after_projection=True, ride_pools=dtype == 'bf16')
    assert sum(1 for o in plan.ops if o['kind'] == 7) == 9 and not any(o['kind'] == 3 for o in plan.ops)
    assert all((o.get('group', 0) > 0) == (dtype == 'bf16') for o in plan.ops if o['kind'] == 7)
    enc = nets.CnnEncoder(plan, cnn_params, B, dtype, DEV, weights_from=plain)
    if dtype == 'bf16':
        enc.autotune(reps=2)
    im0, fm0 = (t.clone() for t in plain.forward(dev(x)))
    im, fm = enc.forward(dev(x))
    net_ref, ep = cnn_ref.inception_v3(cnn_params, x, act_dtype=dtype)
    for name in ('Mixed_5b', 'Mixed_5d', 'Mixed_6b', 'Mixed_6e', 'Mixed_7b'):
        assert_close(enc.end_point(name).float().cpu().numpy(), ep[name], tol, '%s %s' % (name, dtype))
    assert_close(fm.cpu().numpy().reshape(B, 5, 5, 2048), ep['Mixed_7c'], tol, 'Mixed_7c ' + dtype)
    assert_close(im.cpu().numpy(), net_ref.reshape(B, -1), tol, 'im_embed ' + dtype)
    # against the reference op order on the same device arithmetic: only the rounding order differs
    assert_close(fm.cpu().numpy(), fm0.cpu().numpy(), 1e-5 if dtype == 'f32' else 2e-2, 'fm vs plain plan')
    im2, fm2 = enc.forward(dev(x), use_graph=True)
    im2, fm2 = enc.forward(dev(x), use_graph=True)
    assert torch.equal(fm2, fm) and torch.equal(im2, im)
    with pytest.raises(ValueError):
        enc.enable_training()


@pytest.mark.parametrize('B', [3, 70])
def test_stem_stream_with_conv2d_1a_is_bit_identical(cnn_params, B):
    """Op kind 9 (Conv2d_1a_3x3 inside the streaming stem pass, csrc/conv_stem.hip FUSE1A; inception_v3.py:100-111) against
    the separate stem launch + kind 8: the pooled stem output and the whole forward bit for bit (the same operand split,
    MFMA order and epilogue), eagerly and replayed from a graph; more tasks than CUs at B = 70 (a workgroup walks several
    half-images, the image-row ring is re-primed per task)."""
    x = np.random.default_rng(5 + B).uniform(-1, 1, (B, 224, 224, 3)).astype(np.float32)
    sep = nets.CnnPlan('inception_v3', (224, 224), pool_after_projection=True, fuse_pools=True, fuse_stem_1a=False)
    fus = nets.CnnPlan('inception_v3', (224, 224), pool_after_projection=True, fuse_pools=True)
    assert [o['kind'] for o in sep.ops[:2]] == [1, 8] and fus.ops[0]['kind'] == 9 and len(fus.ops) == len(sep.ops) - 1
    assert fus.macs == sep.macs and [w[0] for w in fus.weights] == [w[0] for w in sep.weights]
    e0 = nets.CnnEncoder(sep, cnn_params, B, 'bf16', DEV)
    e1 = nets.CnnEncoder(fus, cnn_params, B, 'bf16', DEV, weights_from=e0)
    im0, fm0 = (t.clone() for t in e0.forward(dev(x)))
    p0 = e0.end_point('MaxPool_3a_3x3').clone()
    im1, fm1 = (t.clone() for t in e1.forward(dev(x)))
    sync()
    assert torch.equal(e1.end_point('MaxPool_3a_3x3'), p0), 'pooled stem output differs'
    assert torch.equal(fm1, fm0) and torch.equal(im1, im0)
    for _ in range(2):
        im2, fm2 = e1.forward(dev(x), use_graph=True)
    assert torch.equal(fm2, fm0) and torch.equal(im2, im0)
    net_ref, ep = cnn_ref.inception_v3(cnn_params, x[:2], act_dtype='bf16')
    assert_close(e1.end_point('MaxPool_3a_3x3')[:2].float().cpu().numpy(), ep['MaxPool_3a_3x3'], 3e-2, 'MaxPool_3a (kind 9)')


@pytest.mark.parametrize('B', [3, 40])
def test_walk_tiles_give_the_bits_of_the_one_tile_launch(cnn_params, B):
    """Tile ids 56..61 (conv_igemm_dma_walk_body: one workgroup per pixel tile walks over the out-channel tiles of all members of a
    shared-input group) against the same groups on tile 44: the whole forward bit for bit, with the 1x1 groups at the head
    of every Inception block on each walk form; a launch whose members do not share their input refuses the ids."""
    x = np.random.default_rng(7 + B).uniform(-1, 1, (B, 224, 224, 3)).astype(np.float32)
    plan = nets.CnnPlan('inception_v3', (224, 224), pool_after_projection=True, fuse_chains=False)
    enc = nets.CnnEncoder(plan, cnn_params, B, 'bf16', DEV)
    heads = [i for i, o in enumerate(plan.ops) if o['kind'] == 0 and o.get('group', 0) and o['KH'] == 1 and o['KW'] == 1 and o['depth'] == 0
             and (i == 0 or plan.ops[i - 1].get('group', 0) != o['group'])]
    assert len(heads) >= 9

    def forward_with(tile):
        for i in heads:
            enc._ops[i].tile = tile
        enc._build_group_args(); enc._drop_graphs()
        im, fm = enc.forward(dev(x))
        sync()
        return im.clone(), fm.clone()
    im0, fm0 = forward_with(44)
    for tile in (56, 57, 58, 59, 60, 61):        # 59..61: the walk of a pixel tile shared by two workgroups (paired walk)
        im1, fm1 = forward_with(tile)
        assert torch.equal(fm1, fm0) and torch.equal(im1, im0), 'walk tile %d differs' % tile
    # a group over different inputs (depth 1: 1x7 | 7x1 of two branches) is not eligible
    other = [i for i, o in enumerate(plan.ops) if o['kind'] == 0 and o.get('group', 0) and o['KH'] * o['KW'] == 7
             and plan.ops[i - 1].get('group', 0) != o['group']]
    enc._ops[other[0]].tile = 56
    with pytest.raises(L.ComicHipError):
        enc._build_group_args()
    enc._ops[other[0]].tile = 0
    enc._build_group_args()


def test_inception_v3_fused_pools_weight_stationary_1x1(cnn_params):
    """Second forward-only rewrite (bf16): MaxPool_3a / MaxPool_5a folded into the loads of the 1x1 convs behind them
    and the thin 1x1 groups of Mixed_5b-d on the weight-stationary kernel (csrc/conv_ws.hip; reference
    inception_v3.py:111-114,124-199): against the oracle, and bit for bit against the plain forward-only plan with
    every conv on an im2col tile (same k order per accumulator, exact max).  Ragged batch: the last 64-pixel tile of
    every layer is partial."""
    B = 3
    x = np.random.default_rng(12).uniform(-1, 1, (B, 224, 224, 3)).astype(np.float32)
    pa = nets.CnnPlan('inception_v3', (224, 224), pool_after_projection=True, fuse_chains=False)
    pb = nets.CnnPlan('inception_v3', (224, 224), pool_after_projection=True, fuse_pools=True, fuse_chains=True)
    assert sum(1 for o in pb.ops if o['kind'] == 2) == sum(1 for o in pa.ops if o['kind'] == 2) - 2
    # MaxPool_5a folded into the four 1x1 convs of Mixed_5b; Conv2d_1a -> 2a -> 2b -> MaxPool_3a is one streaming op (kind 9)
    assert sum(1 for o in pb.ops if o.get('flags', 0) & 2) == 4 and sum(1 for o in pb.ops if o['kind'] == 9) == 1
    assert pb.macs == pa.macs and pb.weights == pa.weights
    ea = nets.CnnEncoder(pa, cnn_params, B, 'bf16', DEV)
    eb = nets.CnnEncoder(pb, cnn_params, B, 'bf16', DEV, weights_from=ea)
    for i, o in enumerate(pa.ops):
        if o['kind'] == 0:
            ea._ops[i].tile = 3                    # 64x64 im2col tile everywhere
    for i, o in enumerate(pb.ops):                 # batch 3 is below the default's size threshold: ask for the kernel
        if o['kind'] == 0 and o['KH'] == 1 and o['Cin'] in (256, 288) and o['Ho'] == 25:
            eb._ops[i].tile = nets.L.WS_TILE
    for e, plan in ((ea, pa), (eb, pb)):           # the row-walking pool + BN + ReLU kernel (default only at large batches)
        for i, o in enumerate(plan.ops):
            if o['kind'] == 7:
                e._ops[i].tile = 1
    ci = [i for i, o in enumerate(pb.ops) if o.get('tile') == nets.L.CHAIN_TILE][0]
    eb._ops[ci].tile = 3                           # linked convs on another tile id are refused, not run side by side
    with pytest.raises(L.ComicHipError):
        eb._build_group_args()
    eb._ops[ci].tile = nets.L.CHAIN_TILE
    ea._build_group_args()
    eb._build_group_args()
    ima, fma = (t.clone() for t in ea.forward(dev(x)))
    imb, fmb = eb.forward(dev(x))
    sync()
    net_ref, ep = cnn_ref.inception_v3(cnn_params, x, act_dtype='bf16')
    for name in ('MaxPool_3a_3x3', 'Conv2d_3b_1x1', 'Mixed_5b', 'Mixed_5c', 'Mixed_5d', 'Mixed_6c', 'Mixed_7b'):
        assert_close(eb.end_point(name).float().cpu().numpy(), ep[name], 3e-2, name)
        assert torch.equal(eb.end_point(name).view(torch.int16), ea.end_point(name).view(torch.int16)), name
    assert torch.equal(fma, fmb) and torch.equal(ima, imb)
    with pytest.raises(ValueError):
        nets.CnnEncoder(pb, cnn_params, B, 'f32', DEV)
    with pytest.raises(ValueError):
        nets.CnnPlan('inception_v3', (224, 224), fuse_pools=True)


@pytest.mark.parametrize('B', [3, 70])
def test_fused_branch_chains_are_bit_identical(cnn_params, B):
    """Third forward-only rewrite (`CnnPlan(fuse_chains=True)`, tile CHAIN_TILE, csrc/conv_img.hip conv_img_chain_kernel): the
    1x7 / 7x1 convs of a Mixed_6b-e branch (inception_v3.py:262-345) as ONE launch per block, a workgroup per (image, branch),
    the intermediate 12x12 maps rewritten in place in the LDS.  Against the same plan with one launch per conv depth: every
    block output from Mixed_6b on, the feature map and the pooled embedding bit for bit (same operands, k order and epilogue
    arithmetic per value), eagerly and replayed from a graph; against the oracle at the bf16 tolerance."""
    x = np.random.default_rng(31 + B).uniform(-1, 1, (B, 224, 224, 3)).astype(np.float32)
    sep = nets.CnnPlan('inception_v3', (224, 224), pool_after_projection=True, fuse_pools=True, fuse_chains=False)
    fus = nets.CnnPlan('inception_v3', (224, 224), pool_after_projection=True, fuse_pools=True, fuse_chains=True)
    assert fus.fuse_chains and not sep.fuse_chains and len(fus.ops) == len(sep.ops)
    ch = [o for o in fus.ops if o.get('tile') == nets.L.CHAIN_TILE]
    assert len(ch) == 26 and sum(1 for o in ch if o.get('flags', 0) & nets.L.OP_CHAIN_LINK) == 17       # (Mixed_7a: 1x7 -> 7x1)
    assert len({o['group'] for o in ch}) == 5 and {o['Cin'] for o in ch} == {128, 160, 192}
    assert not any(o.get('tile') == nets.L.CHAIN_TILE for o in sep.ops)
    assert fus.macs == sep.macs and fus.weights == sep.weights
    e0 = nets.CnnEncoder(sep, cnn_params, B, 'bf16', DEV)
    e1 = nets.CnnEncoder(fus, cnn_params, B, 'bf16', DEV, weights_from=e0)
    im0, fm0 = (t.clone() for t in e0.forward(dev(x)))
    im1, fm1 = (t.clone() for t in e1.forward(dev(x)))
    sync()
    for name in ('Mixed_6a', 'Mixed_6b', 'Mixed_6c', 'Mixed_6d', 'Mixed_6e', 'Mixed_7a'):
        assert torch.equal(e1.end_point(name).view(torch.int16), e0.end_point(name).view(torch.int16)), name
    assert torch.equal(fm1, fm0) and torch.equal(im1, im0)
    for _ in range(2):
        im2, fm2 = e1.forward(dev(x), use_graph=True)
    assert torch.equal(fm2, fm0) and torch.equal(im2, im0)
    net_ref, ep = cnn_ref.inception_v3(cnn_params, x[:2], act_dtype='bf16')
    for name in ('Mixed_6b', 'Mixed_6e'):
        assert_close(e1.end_point(name)[:2].float().cpu().numpy(), ep[name], 3e-2, name + ' (fused chains)')
    # the autotuner leaves the chain launches alone and keeps the result
    e1.autotune(reps=1)
    assert all(e1._ops[i].tile == nets.L.CHAIN_TILE for i, o in enumerate(fus.ops) if o.get('tile') == nets.L.CHAIN_TILE)
    im3, fm3 = e1.forward(dev(x))
    sync()
    assert torch.equal(fm3, fm0) and torch.equal(im3, im0)


def test_fused_chains_of_a_trainable_plan_keep_every_intermediate_map(cnn_params):
    """Plans without the forward-only rewrites (cnn_finetune: the backward reads every conv's output) run the fused chain
    launches with COMIC_OP_CHAIN_KEEP: each linked conv also stores its map.  EVERY buffer of the plan -- the 17 intermediate
    12x12 maps among them -- bit for bit the buffer of the same plan with one launch per conv depth; an fp32 encoder over
    the same plan runs its convs one by one (no chain launch on the exact-fp32 path)."""
    B = 5
    x = np.random.default_rng(77).uniform(-1, 1, (B, 224, 224, 3)).astype(np.float32)
    sep = nets.CnnPlan('inception_v3', (224, 224), fuse_chains=False)
    fus = nets.CnnPlan('inception_v3', (224, 224), fuse_chains=True)
    ch = [o for o in fus.ops if o.get('tile') == nets.L.CHAIN_TILE]
    keep = nets.L.OP_CHAIN_LINK | nets.L.OP_CHAIN_KEEP
    assert fus.fuse_chains and fus.keep_chain_maps and len(ch) == 26 and sum(1 for o in ch if o.get('flags', 0) == keep) == 17
    assert fus.buffers == sep.buffers and fus.weights == sep.weights
    e0 = nets.CnnEncoder(sep, cnn_params, B, 'bf16', DEV)
    e1 = nets.CnnEncoder(fus, cnn_params, B, 'bf16', DEV, weights_from=e0)
    for b in e1.bufs:
        b.view(torch.uint8).fill_(0x5A)            # a kept map that is NOT written would keep this pattern
    e0.forward(dev(x)); e1.forward(dev(x))
    sync()
    for bi, (b0, b1) in enumerate(zip(e0.bufs, e1.bufs)):
        if bi != fus.input:
            assert torch.equal(b0.view(torch.uint8), b1.view(torch.uint8)), 'buffer %d %s' % (bi, fus.buffers[bi])
    e32 = nets.CnnEncoder(fus, cnn_params, 2, 'f32', DEV)
    im32, fm32 = e32.forward(dev(x[:2]))
    sync()
    net_ref, ep = cnn_ref.inception_v3(cnn_params, x[:2])
    assert_close(fm32.cpu().numpy().reshape(ep['Mixed_7c'].shape), ep['Mixed_7c'], 1e-3, 'fp32 encoder over a chain plan')


def test_chain_launch_refuses_what_the_kernel_does_not_cover(cnn_params):
    """A COMIC_CHAIN_TILE group whose ops are not a chain (a link flag on the last op, a linked conv that does not feed the next
    op, another map size) is an error of the call, not a wrong answer."""
    B = 2
    fus = nets.CnnPlan('inception_v3', (224, 224), pool_after_projection=True, fuse_pools=True, fuse_chains=True)
    e1 = nets.CnnEncoder(fus, cnn_params, B, 'bf16', DEV)
    ch = [i for i, o in enumerate(fus.ops) if o.get('tile') == nets.L.CHAIN_TILE]
    last = [i for i in ch if not fus.ops[i].get('flags', 0) & nets.L.OP_CHAIN_LINK][-1]
    x = dev(np.zeros((B, 224, 224, 3), np.float32))
    e1._ops[last].flags |= nets.L.OP_CHAIN_LINK
    with pytest.raises(L.ComicHipError):
        e1.forward(x)
    e1._ops[last].flags &= ~nets.L.OP_CHAIN_LINK
    first = ch[0]
    keep = e1._ops[first].dst
    e1._ops[first].dst = e1._ops[first].src
    with pytest.raises(L.ComicHipError):
        e1.forward(x)
    e1._ops[first].dst = keep
    e1.forward(x)
    sync()


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


def _cnn_grads_device(enc, t):
    """{variable name: gradient} in the slim layout from the packed fp32 gradient buffers."""
    return enc.export_grads()


@pytest.mark.parametrize('dtype,tol', [('f32', 1e-3), ('bf16', 3e-2)])
def test_inception_v1_forward_224(dtype, tol):
    """The reference's default backbone (train.py:56,65): Inception-V1, feature map Mixed_4f
    (14x14x832, an INNER end point handed over in fp32), net = 7x7 average of Mixed_5c; SAME-padded
    7x7 stride-2 stem and max pools, stride-1 max pools inside the blocks, 24-channel reduces padded
    to the MFMA tile."""
    B = 2
    params = cnn_ref.randomize_bn(cnn_ref.init_params_v1(0), seed=1)
    x = np.random.default_rng(3).uniform(-1, 1, (B, 224, 224, 3)).astype(np.float32)
    plan = nets.CnnPlan('inception_v1', (224, 224), 'Mixed_4f')
    assert sum(int(np.prod(v)) for v in plan.param_shapes().values()) == 5607184       # inception_v1_test.py:124-132
    enc = nets.CnnEncoder(plan, params, B, dtype, DEV)
    im, fm = enc.forward(dev(x))
    sync()
    net_ref, ep = cnn_ref.inception_v1(params, x, act_dtype=dtype)
    assert fm.shape == (B, 196, 832) and im.shape == (B, 1024) and fm.dtype == torch.float32
    for name in ('Conv2d_1a_7x7', 'MaxPool_2a_3x3', 'Conv2d_2c_3x3', 'Mixed_3c', 'Mixed_4b', 'Mixed_4e', 'Mixed_5b'):
        got = enc.end_point(name).float().cpu().numpy()
        assert_close(got[..., :ep[name].shape[-1]], ep[name], tol, '%s %s' % (name, dtype))
    assert_close(fm.cpu().numpy().reshape(B, 14, 14, 832), ep['Mixed_4f'], tol, 'Mixed_4f ' + dtype)
    assert_close(im.cpu().numpy(), net_ref.reshape(B, -1), tol, 'net ' + dtype)
    # the row-walking stride-1 max-pool (default only at large batches): the same bits as the per-pixel kernel
    keep = {n: enc.end_point(n).clone() for n in ('Mixed_3b', 'Mixed_4c', 'Mixed_5c')}
    n_forced = 0
    for i, o in enumerate(plan.ops):
        if o['kind'] == 2 and o['SH'] == 1:
            enc._ops[i].tile = 1
            n_forced += 1
    assert n_forced == 9
    im2, fm2 = enc.forward(dev(x))
    sync()
    assert torch.equal(fm2, fm) and torch.equal(im2, im)
    for n, v in keep.items():
        assert torch.equal(enc.end_point(n), v), n


def test_inception_v1_backward_224_f32():
    """cnn_finetune on the default backbone: gradients enter at the INNER feature map (Mixed_4f) and at
    the pooled vector; d weights / d beta of all 57 convs vs the oracle's reverse pass (fp32)."""
    B = 2
    params = cnn_ref.randomize_bn(cnn_ref.init_params_v1(0), seed=1)
    rng = np.random.default_rng(13)
    x = rng.uniform(-1, 1, (B, 224, 224, 3)).astype(np.float32)
    d_net, d_fm = _seeds(rng, B, 196, 832)
    d_net = (1 + 0.5 * rng.standard_normal((B, 1024))).astype(np.float32)
    enc = nets.CnnEncoder(nets.CnnPlan('inception_v1', (224, 224), 'Mixed_4f'), params, B, 'f32', DEV)
    enc.forward(dev(x))
    enc.backward(dev(d_fm), dev(d_net))
    sync()
    got = enc.export_grads()
    # the oracle differentiates through the device's activations (same ReLU masks / pool arg-maxima: every block has a
    # 3x3 stride-1 max-pool branch whose arg-max flips between two free-running forwards reach every layer)
    worst = [0.0]
    want, _, _ = cnn_ref.inception_v1_grads(params, x, d_net, d_fm, override=_device_activation_hook(enc, 1e-4, worst))
    assert set(got) == set(want) and len(want) == 114
    errs = sorted((rel_err(got[k], want[k]), k) for k in want)
    assert errs[-1][0] < F32_RTOL, errs[-1]
    assert errs[len(errs) // 2][0] < 1e-4, errs[len(errs) // 2]


def _device_activation_hook(enc, layer_tol, worst):
    """override hook for cnn_ref._Net: compares the oracle's output of every conv with the DEVICE's activation (max-norm,
    `layer_tol`; the largest error is kept in worst[0]) and hands the device's values to the rest of the oracle's pass,
    so the reverse pass sees the kernels' own ReLU masks and pool arg-maxima."""
    plan = enc.plan
    by_name = {}
    for o in plan.ops:
        if o['kind'] in (0, 1):
            by_name[plan.weights[o['weight']][0] + '/weights'] = o

    def hook(wn, y):
        o = by_name[wn]
        cout = plan.weights[o['weight']][4]
        b = enc.bufs[o['dst']]
        act = b[..., o['dst_coff']:o['dst_coff'] + cout].float()
        if getattr(plan, 'x3', False) and b.dtype != torch.float32:     # [hi | lo | hi] regions: the value is hi + lo
            C3 = b.shape[-1] // 3
            act = act + b[..., C3 + o['dst_coff']:C3 + o['dst_coff'] + cout].float()
        act = act.cpu().numpy()
        e = rel_err(act, y)
        worst[0] = max(worst[0], e)
        assert e <= layer_tol, '%s: forward rel err %.3e > %.1e' % (wn, e, layer_tol)
        return act
    return hook


def _seeds(rng, B, M, C):
    """Gradients of (net, feature map) with a non-zero mean: with zero-mean seeds a variable's
    gradient is a sqrt(N)-sized random sum, and the handful of ReLU masks / pool arg-maxima that
    differ between two forward passes dominates its relative error."""
    return ((1 + 0.5 * rng.standard_normal((B, C))).astype(np.float32),
            ((1 + 0.5 * rng.standard_normal((B, M, C))) / M).astype(np.float32))


def test_inception_v3_backward_224_f32():
    """cnn_finetune: d(conv weights), d(BN beta) of all 94 convs from seeded gradients of the two
    encoder outputs, against the oracle's reverse pass, 1e-3 per variable (max-norm) -- the north star's bound.  The
    oracle's pass runs layer by layer on the DEVICE's activations (each first checked against the oracle's own output
    of that layer at 1e-4), so both sides differentiate through the same ReLU masks and pool arg-maxima; a free-running
    oracle forward differs in ~1e-6 of the masks of the 109x109 stem maps, which alone moved single variables by 2e-3."""
    B = 2
    params = cnn_ref.randomize_bn(cnn_ref.init_params(0, 224), seed=1)
    rng = np.random.default_rng(11)
    x = rng.uniform(-1, 1, (B, 224, 224, 3)).astype(np.float32)
    d_net, d_fm = _seeds(rng, B, 25, 2048)
    enc = nets.CnnEncoder(nets.CnnPlan('inception_v3', (224, 224)), params, B, 'f32', DEV)
    enc.forward(dev(x))
    t = enc.backward(dev(d_fm), dev(d_net))
    sync()
    got = _cnn_grads_device(enc, t)
    worst = [0.0]
    want, _, _ = cnn_ref.inception_v3_grads(params, x, d_net, d_fm, override=_device_activation_hook(enc, 1e-4, worst))
    assert set(got) == set(want) and len(want) == 188
    errs = sorted((rel_err(got[k], want[k]), k) for k in want)
    assert errs[-1][0] < F32_RTOL, errs[-1]
    assert errs[len(errs) // 2][0] < 1e-5, errs[len(errs) // 2]
    # the trainable copies export back to the checkpoint layout unchanged
    ex = enc.export_params()
    for k in ('InceptionV3/Conv2d_1a_3x3/weights', 'InceptionV3/Mixed_6b/Branch_2/Conv2d_0c_1x7/weights',
              'InceptionV3/Mixed_7a/Branch_0/Conv2d_1a_3x3/BatchNorm/beta'):
        np.testing.assert_array_equal(ex[k], params[k])
    # optimiser slots travel per variable (TF-format checkpoints): export -> import into cleared buffers -> same buffers
    tr_ = enc._train
    slots = {'sc/' + k: v for k, v in enc.export_slots(tr_.dw, tr_.dbeta, 'Adam').items()}
    assert slots['sc/InceptionV3/Mixed_7c/Branch_0/Conv2d_0a_1x1/weights/Adam'].shape == (1, 1, 2048, 320)
    w1, b1 = tr_.dw.like(), tr_.dbeta.like()
    w1.data.fill_(7.0), b1.data.fill_(7.0)
    assert not enc.import_slots(w1, b1, slots, 'Adam_1', 'sc/')          # absent slots: nothing loaded
    assert enc.import_slots(w1, b1, slots, 'Adam', 'sc/')
    for i in range(len(enc.plan.weights)):
        assert torch.equal(w1.view('w%d' % i), tr_.dw.view('w%d' % i)), i
        assert torch.equal(b1.view('b%d' % i), tr_.dbeta.view('b%d' % i)), i


def test_inception_v3_backward_224_bf16x3_meets_the_fp32_bar():
    """cnn_finetune on the bf16x3 plan (csrc/conv.hip conv_backward_x3: d conv as [hi | lo | hi] regions, the weight
    gradient as three bf16 products x_hi dz_hi + x_lo dz_hi + x_hi dz_lo, backward-data as the x3 conv of d conv against
    [Wt_hi | Wt_hi | Wt_lo] into fp32 gradient buffers, pools through hi + lo): d weights / d beta of all 94 convs at the bar
    of the exact-fp32 plan, 1e-3 per variable (test_inception_v3_backward_224_f32), where the plain bf16 plan is at 4e-2.
    The oracle differentiates through the device's activations (each within 1e-4 of its own layer output)."""
    B = 2
    params = cnn_ref.randomize_bn(cnn_ref.init_params(0, 224), seed=1)
    rng = np.random.default_rng(11)
    x = rng.uniform(-1, 1, (B, 224, 224, 3)).astype(np.float32)
    d_net, d_fm = _seeds(rng, B, 25, 2048)
    enc = nets.CnnEncoder(nets.CnnPlan('inception_v3', (224, 224), x3=True), params, B, 'bf16x3', DEV)
    enc.forward(dev(x))
    t = enc.backward(dev(d_fm), dev(d_net))
    sync()
    assert all(g is None or g.dtype == torch.float32 for g in t.gbufs)
    got = _cnn_grads_device(enc, t)
    worst = [0.0]
    want, _, _ = cnn_ref.inception_v3_grads(params, x, d_net, d_fm, override=_device_activation_hook(enc, 1e-4, worst))
    assert set(got) == set(want) and len(want) == 188
    errs = sorted((rel_err(got[k], want[k]), k) for k in want)
    print('bf16x3 backward: worst variable %.2e (%s), median %.2e, worst layer output %.2e' % (
        errs[-1][0], errs[-1][1], errs[len(errs) // 2][0], worst[0]))
    assert errs[-1][0] < F32_RTOL, errs[-1]
    assert errs[len(errs) // 2][0] < 1e-4, errs[len(errs) // 2]
    # a second step over the same buffers (gradient buffers cleared, filters packed ahead by the table launch): same gradients
    enc.forward(dev(x))
    enc.backward(dev(d_fm), dev(d_net))
    sync()
    again = _cnn_grads_device(enc, t)
    for k in want:
        assert rel_err(again[k], got[k]) < 1e-5, k


_CHAIN = [('c', 'c1', 32, (3, 3), 2, 'VALID'), ('c', 'c2', 64, (3, 3), 1, 'SAME'), ('max',),
          ('c', 'c3', 96, (1, 7), 1, 'SAME'), ('avg',), ('c', 'c4', 64, (3, 3), 2, 'VALID'),
          ('c', 'c5', 48, (5, 5), 1, 'SAME')]


def _chain_oracle(params, x, d_net, d_fm, act_dtype, override=None):
    n = cnn_ref._Net(params, None, act_dtype=act_dtype, run=True, tape=True, override=override)
    n.scope.append('Chain')
    h = np.asarray(x, np.float32)
    for op in _CHAIN:
        if op[0] == 'c':
            h = n.conv(h, op[2], op[3], op[4], op[5], op[1])
        elif op[0] == 'max':
            h = n.max_pool(h, 3, 2, 'VALID')
        else:
            h = n.avg_pool(h, 3, 1, 'SAME')
    pooled = n.avg_pool(h, (h.shape[1], h.shape[2]), 1, 'VALID')
    g = n.backward([(pooled, d_net.reshape(pooled.shape)), (h, d_fm.reshape(h.shape))])
    return h, pooled, g


@pytest.mark.parametrize('dtype,tol', [('f32', 1e-3), ('bf16', 4e-2), ('bf16x3', 1e-3)])
@pytest.mark.parametrize('B,size', [(3, 63), (2, 70)])
def test_cnn_backward_chain(dtype, tol, B, size):
    """Every backward kernel on a shallow stack (stem conv, 3x3 SAME, max pool, 1x7, avg pool,
    stride-2 VALID with odd and even input sizes, 5x5 into the fp32 feature map, head pool), both
    plan dtypes, ragged batch: d weights / d beta against the oracle's reverse pass over a
    forward that emulates the plan's storage type."""
    rng = np.random.default_rng(5 + B)
    plan = nets.CnnPlan('chain', (size, size), layers=_CHAIN, x3=dtype == 'bf16x3')
    params = cnn_ref.randomize_bn(plan.init_params(seed=3), seed=4)
    x = rng.uniform(-1, 1, (B, size, size, 3)).astype(np.float32)
    Hf, Wf, Cf, _ = plan.buffers[plan.fm]
    d_net, d_fm = _seeds(rng, B, Hf * Wf, Cf)
    enc = nets.CnnEncoder(plan, params, B, dtype, DEV)
    im, fm = enc.forward(dev(x))
    t = enc.backward(dev(d_fm), dev(d_net))
    sync()
    # (bf16x3: fp32-class activations, 5e-6 from the oracle's -- enough to flip a ReLU mask or a max-pool arg-max here and
    # there, each worth 1e-3 ... 1e-2 of a small layer's gradient: the oracle differentiates through the device's
    # activations, as in the whole-network tests)
    hook = _device_activation_hook(enc, 1e-4, [0.0]) if dtype == 'bf16x3' else None
    h, pooled, want = _chain_oracle(params, x, d_net, d_fm, 'f32' if dtype == 'bf16x3' else dtype, hook)
    assert_close(fm.cpu().numpy().reshape(h.shape), h, tol, 'chain fm ' + dtype)
    assert_close(im.cpu().numpy(), pooled.reshape(B, -1), tol, 'chain pooled ' + dtype)
    got = _cnn_grads_device(enc, t)
    assert set(got) == set(want)
    for k in sorted(want):
        assert_close(got[k], want[k], tol, '%s %s' % (k, dtype))


def test_inception_v3_backward_224_bf16_sanity():
    """Whole-network bf16 backward (the path the cnn_finetune extra times) against the fp32-arithmetic oracle.  The
    oracle's pass consumes the device's bf16 activations layer by layer (each within 2e-2 of the oracle's bf16-emulating
    output of that layer: one layer of bf16 products, not 45), so the masks agree and every variable is held to the
    per-kernel bf16 bound of test_cnn_backward_chain (4e-2) instead of the 0.2 a free-running forward needed."""
    B = 2
    params = cnn_ref.randomize_bn(cnn_ref.init_params(0, 224), seed=1)
    rng = np.random.default_rng(11)
    x = rng.uniform(-1, 1, (B, 224, 224, 3)).astype(np.float32)
    d_net, d_fm = _seeds(rng, B, 25, 2048)
    enc = nets.CnnEncoder(nets.CnnPlan('inception_v3', (224, 224)), params, B, 'bf16', DEV)
    enc.forward(dev(x))
    t = enc.backward(dev(d_fm), dev(d_net))
    sync()
    got = _cnn_grads_device(enc, t)
    worst = [0.0]
    want, _, _ = cnn_ref.inception_v3_grads(params, x, d_net, d_fm, act_dtype='bf16',
                                            override=_device_activation_hook(enc, 2e-2, worst))
    errs = sorted((rel_err(got[k], want[k]), k) for k in want)
    assert errs[-1][0] < 4e-2, errs[-1]
    assert errs[len(errs) // 2][0] < 1e-2, errs[len(errs) // 2]
    assert all(np.isfinite(v).all() for v in got.values())


@pytest.mark.parametrize('dtype,tol', [('f32', 2e-5), ('bf16', 2e-2)])
def test_backward_with_fused_activation_gradients_matches_the_unfused_chain(dtype, tol):
    """comic_cnn_backward_sched hands the activation gradient of a conv with ONE reader (the inner convs of the Inception
    branches) to the epilogue of that reader's backward-data launch.  Same gradients as the chain with the separate
    act_grad launches (backward(act_fusion=False) = COMIC_CNN_BWD_NO_ACT_FUSION of that call): fp32 plan to rounding of the atomics' order, bf16 plan to the one
    bf16 rounding of the intermediate gradient the fused form skips."""
    B = 2
    params = cnn_ref.randomize_bn(cnn_ref.init_params(0, 224), seed=1)
    rng = np.random.default_rng(21)
    x = rng.uniform(-1, 1, (B, 224, 224, 3)).astype(np.float32)
    d_net, d_fm = _seeds(rng, B, 25, 2048)
    enc = nets.CnnEncoder(nets.CnnPlan('inception_v3', (224, 224)), params, B, dtype, DEV)
    lib = L.load()
    out = {}
    for fused in (0, 1):
        enc.forward(dev(x))
        t = enc.backward(dev(d_fm), dev(d_net), act_fusion=bool(fused))
        sync()
        assert t.sched is not None                   # the scheduled (three-lane) backward is the one that fuses
        out[fused] = _cnn_grads_device(enc, t)
    errs = sorted((rel_err(out[1][k], out[0][k]), k) for k in out[0])
    assert errs[-1][0] < tol, errs[-3:]
    moved = sum(1 for k in out[0] if not np.array_equal(out[0][k], out[1][k]))
    if dtype == 'bf16':
        assert moved > 20                                # the fused form did run (it skips a bf16 rounding)


@pytest.mark.parametrize('dtype,tol', [('bf16', 2e-2), ('bf16x3', 2e-5)])
def test_bucketed_backward_matches_the_scheduled_backward(dtype, tol):
    """The data-parallel form of the cnn_finetune backward (comic_cnn_backward bucket by bucket, the callback after each one:
    nets.CnnEncoder.backward(buckets=, on_bucket=)) against the scheduled three-lane pass of the single-GPU step, on the bf16
    plan and on the bf16x3 plan (fp32 gradient buffers): the same gradients up to the summation order of the atomics and --
    bf16 -- the rounding of the intermediate gradients the fused activation gradients skip; every bucket's callback sees its
    ranges complete."""
    B = 2
    params = cnn_ref.randomize_bn(cnn_ref.init_params(0, 224), seed=1)
    rng = np.random.default_rng(23)
    x = rng.uniform(-1, 1, (B, 224, 224, 3)).astype(np.float32)
    d_net, d_fm = _seeds(rng, B, 25, 2048)
    enc = nets.CnnEncoder(nets.CnnPlan('inception_v3', (224, 224), x3=dtype == 'bf16x3'), params, B, dtype, DEV)
    enc.forward(dev(x))
    t = enc.backward(dev(d_fm), dev(d_net))
    sync()
    assert t.sched is not None
    want = _cnn_grads_device(enc, t)
    seen = []
    enc.forward(dev(x))
    buckets = enc.grad_buckets(4)
    t = enc.backward(dev(d_fm), dev(d_net), buckets, lambda ts, bk: seen.append(bk))
    sync()
    got = _cnn_grads_device(enc, t)
    assert seen == list(buckets) and len(buckets) >= 3
    errs = sorted((rel_err(got[k], want[k]), k) for k in want)
    assert errs[-1][0] < tol, errs[-3:]


# ----------------------------------------------------------------------------- decoder ----
def _spec_and_cfg(**kw):
    base = dict(D=128, E=64, V=258, C=192, Cg=192, H=8, M=25)
    base.update(kw)
    spec = cdec.DecoderSpec(**base)
    cfg = dr.DecoderConfig(rnn_size=spec.D, rnn_word_size=spec.E, attn_num_heads=spec.H,
                           cnn_fm_projection=spec.fm_projection, attn_alignment_method=spec.method,
                           attn_probability_fn=spec.prob, attn_context_layer=spec.context_layer,
                           rnn_init_method=spec.init_method, token_type=spec.token_type, softmax_size=spec.V,
                           fm_channels=spec.C, im_embed_size=spec.Cg, start_id=spec.start_id, end_id=spec.end_id,
                           rnn_name=spec.rnn_name)
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
    for k in p:                                                    # --rnn_name LN_LSTM / GRU
        if k == 'b_c' or (k.startswith('cln_') and k.endswith('b')):
            p[k] = (0.1 * rng.standard_normal(p[k].shape)).astype(np.float32)
        if k.startswith('cln_') and k.endswith('g'):
            p[k] = (1 + 0.1 * rng.standard_normal(p[k].shape)).astype(np.float32)
    return p


TRAIN_VARIANTS = [
    dict(),                                                                    # COMIC: radix, tied, add_LN, 8 heads
    dict(D=512, E=256, C=2048, Cg=2048),                                       # full COMIC-256 geometry
    dict(fm_projection='independent', prob='sigmoid'),
    dict(fm_projection=None, H=1, token_type='word', V=300, init_method='project_hidden',
         start_id=298, end_id=299),                                            # InstaPIC-style baseline (config 5)
    dict(fm_projection=None, context_layer=True, method='dot', H=4),
    dict(method='dot', H=2, M=64),
    # persistent time loop (decoder_persist.hip; D = 512 only): the other memory / alignment / probability forms
    dict(D=512, E=256, fm_projection='independent', prob='sigmoid', H=4),
    dict(D=512, E=128, method='dot', H=16, M=64),                               # W_q columns from L2 (keys fill the LDS)
    dict(D=512, E=256, M=64),                                                   # 299-pixel map: own-rows backward loop, add_LN
    dict(D=512, E=256, C=832, Cg=1024, M=196, H=4),                             # large-memory loops, 1 / 4 heads a channel quarter
    dict(D=512, E=128, C=832, Cg=1024, M=196, H=16, method='dot'),
    dict(D=512, E=256, C=832, Cg=1024, M=130, H=16),                            # M not a multiple of 4: ragged own-row quarters
    # the reference CLI's default geometry (train.py:56,65): Inception-V1 Mixed_4f, 14 x 14 x 832 -> M = 196; the
    # attention kernels run in their split form (several workgroups per batch row, decoder.hip)
    dict(D=512, E=256, C=832, Cg=1024, M=196),
    dict(D=512, E=256, C=832, Cg=1024, M=196, fm_projection='independent', prob='sigmoid'),
    # --rnn_name LN_LSTM / GRU (model_base.py:622-629; cells.hip), both state initialisations
    dict(rnn_name='LN_LSTM'),
    dict(rnn_name='LN_LSTM', D=512, E=256, init_method='project_hidden', fm_projection='independent'),
    dict(rnn_name='GRU'),
    dict(rnn_name='GRU', D=512, E=256, init_method='project_hidden', method='dot', H=2),
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
    if kw in (dict(D=512, E=256, C=832, Cg=1024, M=196), dict(D=512, E=256, M=64), dict(D=512, E=256, C=2048, Cg=2048)):
        # these geometries meet the oracle THROUGH both persistent time loops (Inception-V1 Mixed_4f, the 299-pixel map, COMIC-256)
        assert dec.lib.comic_decoder_train_path() == 3
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


@pytest.mark.parametrize('geo', [dict(C=832, Cg=1024, M=196), dict(M=64)])
def test_large_memory_loops_graph_replay_equals_eager(geo):
    """The large-memory forms of the persistent loops under hipGraph capture / replay, on a second decoder with the same
    seed: the same bits as eager launches -- also says that the backward loop's d keys, added to memory with float atomics
    by ONE writer per address in step order, come out the same in every run."""
    spec, cfg = _spec_and_cfg(**dict(dict(D=512, E=256, C=2048, Cg=2048), **geo))
    p = _rand_params(cfg, 5)
    a, b = cdec.Decoder(spec, p, DEV, seed=3), cdec.Decoder(spec, p, DEV, seed=3)
    fm, im, caps = _batch(spec, 64, 20, 51)
    for it in range(3):                                   # call 1 eager, call 2 captures, call 3 replays
        ra = a.train_step(dev(fm), dev(im), caps, training=True, seed=50 + it, use_graph=False)
        rb = b.train_step(dev(fm), dev(im), caps, training=True, seed=50 + it, use_graph=True)
        sync()
        assert a.lib.comic_decoder_train_path() == 3
        assert float(ra['loss']) == float(rb['loss']), it
        assert torch.equal(a.grads.data, b.grads.data), it


def test_split_train_step_equals_whole_step():
    """SCST: the update's forward pass is enqueued before the rewards exist (Decoder.train_step(phase='fwd'), then phase='bwd'
    with the rewards; COMIC_DEC_PHASE_FWD / _BWD over one workspace).  Same kernels in the same order: the same bits as the
    one-call step, eagerly and replayed from the two phase graphs, with the dropout masks generated on the device."""
    spec, cfg = _spec_and_cfg(D=512, E=256, C=2048, Cg=2048)
    B, Lc = 40, 24
    p = _rand_params(cfg, 5)
    fm, im, caps = _batch(spec, B, Lc, 41)
    rewards = np.random.default_rng(3).standard_normal(B).astype(np.float32)
    whole, split = cdec.Decoder(spec, p, DEV, seed=7), cdec.Decoder(spec, p, DEV, seed=7)
    for it in range(3):                                   # call 1 eager, call 2 captures, call 3 replays
        a = whole.train_step(dev(fm), dev(im), caps, rewards=rewards, training=True, seed=100 + it, use_graph=True)
        assert split.train_step(dev(fm), dev(im), caps, training=True, seed=100 + it, use_graph=True, phase='fwd') is None
        b = split.train_step(None, None, caps, rewards=rewards, training=True, use_graph=True, phase='bwd')
        sync()
        assert whole.lib.comic_decoder_train_path() == 3
        assert float(a['loss']) == float(b['loss']) and float(a['map_loss']) == float(b['map_loss']), it
        assert torch.equal(whole.grads.data, split.grads.data), it
        assert torch.equal(a['logits'], b['logits']), it


@pytest.mark.parametrize('B,geo', [(64, {}), (23, {}), (64, dict(C=832, Cg=1024, M=196)), (23, dict(C=832, Cg=1024, M=196)),
                                   (64, dict(M=64)), (23, dict(M=64, H=16)),
                                   (80, dict(C=832, Cg=1024, M=196))])          # five groups: two launches each way
def test_persistent_time_loop_equals_step_launches(B, geo, monkeypatch):
    """The one-launch forward time loop (decoder_persist.hip: 64 workgroups per 16 batch rows, sc1 hand-offs, counter
    barriers) against the per-step launch chain it replaces, at the bench geometry with every dropout on: same saved
    activations up to fp32 summation order.  Two different batches through the same buffers: a stale hand-off (a byte
    of the previous launch read in place of this one's) would show in the second."""
    # geo M = 196 (Inception-V1 Mixed_4f, the reference CLI's default map): the forward loop in its large-memory form (a
    # workgroup holds its channel quarter of the keys, the LayerNorm sums cross the four quarters), the backward loop in its
    # own-rows form with up to eight rows a wave (d keys added to memory step by step).  geo M = 64 (the 8 x 8 map of
    # 299-pixel inputs): the backward loop in its own-rows form (a workgroup holds its 16 memory rows, the softmax
    # backward's dot products cross the four quarters).  Path 3 everywhere.
    spec, cfg = _spec_and_cfg(**dict(dict(D=512, E=256, C=2048, Cg=2048), **geo))
    Lc = 30
    dec = cdec.Decoder(spec, _rand_params(cfg, 5), DEV)
    for seed in (31, 32):
        fm, im, caps = _batch(spec, B, Lc, seed)
        _, _, _, lens = dr.process_inputs(caps, cfg.token_type)
        masks = dr.make_dropout_masks(cfg, B, int(lens.max()), spec.M, seed)
        got = {}
        for mode in ('11', '10', '00'):                    # forward + backward loops | forward loop only | launches
            monkeypatch.setenv('COMIC_PERSIST', mode[0])
            monkeypatch.setenv('COMIC_PERSIST_BWD', mode[1])
            res = dec.train_step(dev(fm), dev(im), caps, masks=masks, training=True, want_input_grads=True)
            sync()
            assert dec.lib.comic_decoder_train_path() == {'11': 3, '10': 1, '00': 0}[mode]   # the loops really ran
            got[mode] = dict(logits=res['logits'].cpu().numpy(), maps=res['attn_maps'].cpu().numpy(),
                             loss=float(res['loss']), map_loss=float(res['map_loss']), dfm=res['dfm'].cpu().numpy(),
                             dim=res['dim_embed'].cpu().numpy(), g=dec.grads.to_numpy())
        ref = got['00']
        for mode in ('11', '10'):
            g = got[mode]
            assert np.isfinite(g['loss']) and np.isfinite(g['map_loss']), mode
            assert_close(g['logits'], ref['logits'], 2e-5, mode + ' logits')
            assert_close(g['maps'], ref['maps'], 2e-5, mode + ' attention maps')
            assert_close(g['dfm'], ref['dfm'], 1e-4, mode + ' d feature map')
            assert_close(g['dim'], ref['dim'], 1e-4, mode + ' d image embedding')
            for k in ref['g']:
                assert_close(g['g'][k], ref['g'][k], 1e-4, mode + ' grad ' + k)
            assert abs(g['loss'] - ref['loss']) <= 1e-5 * abs(ref['loss'])


@pytest.mark.parametrize('B', [64, 224, -64])
def test_persistent_loops_match_oracle_at_bench_geometry(B, monkeypatch):
    """The persistent forward and backward time loops against the ORACLE (not against the per-step launches) at the
    geometry bench.py times -- COMIC-256: D = 512, E = 256, C = 2048, M = 25, 8 heads, tied, T' = 29, every dropout on
    with injected masks -- at batch 64 (four 16-row groups in one launch) and at the SCST step's 224 hypotheses (four
    consecutive launches).  Same 1e-3 bar as the small cases, max-norm and element-wise."""
    if B < 0:      # the backward loop in its own-rows form (the form of memories of more than 64 rows) at M = 25
        B = -B
        monkeypatch.setenv('COMIC_BWD_OWN_ROWS', '1')
    spec, cfg = _spec_and_cfg(D=512, E=256, C=2048, Cg=2048)
    Lc = 30                                               # row 0 has the longest caption: T' = Lc - 1 = 29
    p = _rand_params(cfg, 13)
    fm, im, caps = _batch(spec, B, Lc, 17)
    _, _, _, lens = dr.process_inputs(caps, cfg.token_type)
    assert int(lens.max()) == 29
    masks = dr.make_dropout_masks(cfg, B, int(lens.max()), spec.M, 19)
    cfg.l2_decay = 0.0
    out = dr.train_forward(p, cfg, fm, im, caps, masks, None)
    grads, dfm, dim = dr.train_backward(p, cfg, out)
    dec = cdec.Decoder(spec, p, DEV)
    res = dec.train_step(dev(fm), dev(im), caps, masks=masks, training=True, want_input_grads=True)
    sync()
    assert dec.lib.comic_decoder_train_path() == 3        # both loops ran as persistent launches
    assert_close(res['logits'].cpu().numpy(), out['logits'], F32_RTOL, 'logits')
    assert_close(res['attn_maps'].cpu().numpy(), out['attn_maps'], F32_RTOL, 'attn_maps')
    assert abs(float(res['loss']) - float(out['xe'])) <= F32_RTOL * abs(float(out['xe']))
    assert abs(float(res['map_loss']) - float(out['map_loss'])) <= F32_RTOL * abs(float(out['map_loss'])) + 1e-7
    g = dec.grads.to_numpy()
    for k in grads:
        assert_close(g[k], grads[k], F32_RTOL, 'grad ' + k)
    assert_close(res['dfm'].cpu().numpy(), dfm, F32_RTOL, 'dfm')
    assert_close(res['dim_embed'].cpu().numpy(), dim, F32_RTOL, 'dim_embed')


def test_persistent_loop_timeout_voids_the_step():
    """A bounded wait of a persistent loop that expires must not train on garbage: the executor's last
    launch turns the step's losses into NaN and every gradient into zeros (comic_persist_gate), so the fused Adam that
    follows without a host check applies no gradient, and the host sees NaN at its next look at the loss.  The timeout
    is injected into ONE call (COMIC_DEC_INJECT_TIMEOUT in that call's descriptor); the step before and the step after are healthy."""
    spec, cfg = _spec_and_cfg(D=512, E=256, C=2048, Cg=2048)
    B, Lc = 32, 12
    dec = cdec.Decoder(spec, _rand_params(cfg, 6), DEV)
    fm, im, caps = _batch(spec, B, Lc, 43)
    _, _, _, lens = dr.process_inputs(caps, cfg.token_type)
    masks = dr.make_dropout_masks(cfg, B, int(lens.max()), spec.M, 43)
    good = dec.train_step(dev(fm), dev(im), caps, masks=masks, training=True, want_input_grads=True)
    sync()
    assert dec.lib.comic_decoder_train_path() == 3
    want = (float(good['loss']), dec.grads.data.clone(), good['dfm'].clone())
    assert np.isfinite(want[0]) and float(want[1].abs().max()) > 0
    bad = dec.train_step(dev(fm), dev(im), caps, masks=masks, training=True, want_input_grads=True, inject_timeout=True)
    sync()
    assert np.isnan(float(bad['loss'])) and np.isnan(float(bad['map_loss']))
    assert float(dec.grads.flat.abs().max()) == 0.0
    assert float(bad['dfm'].abs().max()) == 0.0 and float(bad['dim_embed'].abs().max()) == 0.0
    # the voided step is LOUD and a no-op: the gradient buffer's status word says "voided", the parameter
    # buffer's sticky count went up, and the gated optimiser leaves parameters AND its moments untouched
    assert float(dec.grads.status) == 1.0 and dec.voided_steps() == 1
    from comic_amd import optim
    for opt in (optim.AdamTF(dec.params), optim.MomentumTF(dec.params)):
        opt.m.data.fill_(0.25); opt.v.data.fill_(0.5)
        before = (dec.params.data.clone(), opt.m.data.clone(), opt.v.data.clone())
        opt.step(dec.grads, 1e-2)
        sync()
        assert torch.equal(dec.params.data, before[0]) and torch.equal(opt.m.data, before[1]) and torch.equal(opt.v.data, before[2])
    again = dec.train_step(dev(fm), dev(im), caps, masks=masks, training=True, want_input_grads=True)   # the flag was that call's alone
    sync()
    assert float(again['loss']) == want[0] and torch.equal(dec.grads.flat, want[1][:dec.grads.numel]) and torch.equal(again['dfm'], want[2])
    assert float(dec.grads.status) == 0.0 and dec.voided_steps() == 1      # healthy again; the count stays
    opt = optim.AdamTF(dec.params)
    before = dec.params.data.clone()
    opt.step(dec.grads, 1e-2)
    sync()
    assert not torch.equal(dec.params.flat, before[:dec.params.numel])   # ... and a healthy step's update is applied


def test_persistent_loops_beside_resident_kernels_of_another_stream():
    """What a collective's kernels do to a training step (VERDICT r3, N > 1 risk): workgroups of ANOTHER stream stay
    resident on some CUs (comic_debug_occupy_cus: 24 workgroups for ~3 ms, about what RCCL's channels hold) while the
    persistent loops -- which need a workgroup on every CU -- are launched.  The loops must wait for those CUs and then
    complete: same bits as the undisturbed step, no voided step (a bounded wait of 2^20 polls is seconds, not
    milliseconds)."""
    spec, cfg = _spec_and_cfg(D=512, E=256, C=2048, Cg=2048)
    B, Lc = 64, 14
    dec = cdec.Decoder(spec, _rand_params(cfg, 8), DEV)
    fm, im, caps = _batch(spec, B, Lc, 47)
    _, _, _, lens = dr.process_inputs(caps, cfg.token_type)
    masks = dr.make_dropout_masks(cfg, B, int(lens.max()), spec.M, 47)
    quiet = dec.train_step(dev(fm), dev(im), caps, masks=masks, training=True)
    sync()
    assert dec.lib.comic_decoder_train_path() == 3
    want = (float(quiet['loss']), dec.grads.flat.clone())
    side = torch.cuda.Stream()
    for n_wg, usec in ((24, 3000), (256, 1500), (512, 500)):
        with torch.cuda.stream(side):
            L.check(dec.lib.comic_debug_occupy_cus(n_wg, usec, side.cuda_stream))
        res = dec.train_step(dev(fm), dev(im), caps, masks=masks, training=True)
        sync()
        assert float(res['loss']) == want[0] and torch.equal(dec.grads.flat, want[1]), (n_wg, usec)
        assert float(dec.grads.status) == 0.0 and dec.voided_steps() == 0, (n_wg, usec)


@pytest.mark.parametrize('kw,B,Lc', [
    (dict(E=64, H=4, M=1), 1, 4),                                  # one row, one memory row, one x block per wave
    (dict(E=512, H=16, M=28), 17, 9),                              # second group holds a single row; widest x third
    (dict(E=128, M=7, method='dot'), 33, 6),                       # three groups, dot scores
    (dict(E=256, M=25, fm_projection='independent'), 50, 12),      # untied values: forward loop only
    (dict(E=256, M=25, prob='sigmoid'), 64, 5),                    # sigmoid probability: forward loop only
    (dict(E=256, M=25), 100, 7),                                   # 7 groups: two consecutive launches (4 + 3 groups)
    (dict(E=256, M=25), 224, 6),                                   # the SCST step's 224 hypotheses: four launches
])
def test_persistent_time_loops_odd_shapes(kw, B, Lc, monkeypatch):
    """Shapes at the edges of what the persistent loops accept (ragged groups, a single row, one memory row, every x-third
    width, 4 and 16 heads, configurations only the forward loop covers) against the per-step launches."""
    spec, cfg = _spec_and_cfg(D=512, C=96, Cg=64, **kw)
    dec = cdec.Decoder(spec, _rand_params(cfg, 8), DEV)
    fm, im, caps = _batch(spec, B, Lc, 51)
    _, _, _, lens = dr.process_inputs(caps, cfg.token_type)
    masks = dr.make_dropout_masks(cfg, B, int(lens.max()), spec.M, 52)
    got = {}
    for mode in ('11', '00'):
        monkeypatch.setenv('COMIC_PERSIST', mode[0])
        monkeypatch.setenv('COMIC_PERSIST_BWD', mode[1])
        res = dec.train_step(dev(fm), dev(im), caps, masks=masks, training=True, want_input_grads=True)
        sync()
        got[mode] = (dec.lib.comic_decoder_train_path(), res['logits'].cpu().numpy(), res['attn_maps'].cpu().numpy(),
                     res['dfm'].cpu().numpy(), dec.grads.to_numpy(), float(res['loss']))
    full = spec.fm_projection == 'tied' and spec.prob == 'softmax'
    assert got['11'][0] == (3 if full else 1) and got['00'][0] == 0
    assert_close(got['11'][1], got['00'][1], 2e-5, 'logits')
    assert_close(got['11'][2], got['00'][2], 2e-5, 'attention maps')
    assert_close(got['11'][3], got['00'][3], 1e-4, 'd feature map')
    for k in got['00'][4]:
        assert_close(got['11'][4][k], got['00'][4][k], 1e-4, 'grad ' + k)
    assert abs(got['11'][5] - got['00'][5]) <= 1e-5 * abs(got['00'][5])


def test_persistent_time_loops_replay_from_a_hipgraph():
    """The step with both persistent loops captured in a hipGraph (CaptionModel.run_train_step's path) and replayed
    equals the eager step: the sentinel fill, the loops and their error check are all nodes of the graph."""
    spec, cfg = _spec_and_cfg(D=512, E=256, C=2048, Cg=2048)
    B, Lc = 32, 14
    dec = cdec.Decoder(spec, _rand_params(cfg, 6), DEV)
    fm, im, caps = _batch(spec, B, Lc, 41)
    _, _, _, lens = dr.process_inputs(caps, cfg.token_type)
    masks = dr.make_dropout_masks(cfg, B, int(lens.max()), spec.M, 41)
    ref = dec.train_step(dev(fm), dev(im), caps, masks=masks, training=True)
    sync()
    want = (float(ref['loss']), float(ref['map_loss']), dec.grads.data.clone())
    for _ in range(3):                                   # eager, capture, replay
        res = dec.train_step(dev(fm), dev(im), caps, masks=masks, training=True, use_graph=True)
        sync()
        assert float(res['loss']) == want[0] and float(res['map_loss']) == want[1]
        assert torch.equal(dec.grads.data, want[2])


def test_train_step_inputs_survive_host_run_ahead():
    """The host issues steps faster than the GPU executes them: every step must train on ITS captions (the
    pinned staging buffers of the async host-to-device copies are rewritten only after the copy that last used
    them has executed).  A long kernel is queued first so that all four steps are issued before the first runs."""
    spec, cfg = _spec_and_cfg()
    B, Lc = 6, 11
    fm, im, caps_a = _batch(spec, B, Lc, 21)
    _, _, caps_b = _batch(spec, B, Lc, 22)
    _, _, caps_c = _batch(spec, B, Lc, 23)
    for c in (caps_b, caps_c):                       # same shape key (B, T, T') as caps_a: same context, same staging slots
        c[0] = caps_a[0]
    dec = cdec.Decoder(spec, _rand_params(cfg, 2), DEV)
    fmd, imd = dev(fm), dev(im)
    want = []
    for c in (caps_a, caps_b, caps_c, caps_a):
        want.append(float(dec.train_step(fmd, imd, c, training=False)['loss']))       # float(): synchronises
    assert len({round(w, 6) for w in want[:3]}) == 3
    big = torch.randn(8192, 8192, device=DEV)
    sync()
    for _ in range(40):
        big = torch.mm(big, big) * 1e-4              # ~100 ms of queued work
    got = [dec.train_step(fmd, imd, c, training=False)['loss'].clone() for c in (caps_a, caps_b, caps_c, caps_a)]
    sync()
    assert [float(g) for g in got] == want


@pytest.mark.parametrize('group', [1, 2, 3])
def test_pipelined_xe_steps_equal_serial_steps(group):
    """Frozen-CNN pipelining (trainer.submit_images / xe_step_pending): the encoder forward of the next step -- with
    encoder_group S, ONE forward over the batches of the next S steps -- runs on a second stream under the
    decoder step.  Six steps on six different image batches must give the losses and the final parameters of the
    serial xe_step sequence, bit for bit (same kernels on the same rows: the CNN is batch-independent)."""
    from comic_amd import trainer
    B, size, Lc, steps = 3, 63, 9, 6
    plan = nets.CnnPlan('chain', (size, size), layers=_CHAIN)
    cnn_p = cnn_ref.randomize_bn(plan.init_params(seed=3), seed=4)
    Hf, Wf, Cf, _ = plan.buffers[plan.fm]
    spec, cfg = _spec_and_cfg(C=Cf, Cg=Cf, M=Hf * Wf)
    p = _rand_params(cfg, 3)
    rng = np.random.default_rng(23)
    xs = [dev(rng.uniform(-1, 1, (B, size, size, 3)).astype(np.float32)) for _ in range(steps)]
    caps = [_batch(spec, B, Lc, 30 + i)[2] for i in range(steps)]
    for c in caps:
        c[0] = caps[0][0]                            # same (B, T, T') key: one decoder context
    kw = dict(lr_start=1e-2, lr_end=1e-2, max_step=10, plan=plan)
    ser = trainer.CaptionTrainer(cnn_p, spec, p, B, (size, size), 'f32', DEV, **kw)
    want = [float(ser.xe_step(xs[i], caps[i], training=False)['loss']) for i in range(steps)]
    assert len({round(w, 6) for w in want}) == steps
    tr = trainer.CaptionTrainer(cnn_p, spec, p, B, (size, size), 'f32', DEV, encoder_group=group, **kw)
    tr.enable_overlap()
    groups = [torch.cat(xs[g:g + group]) for g in range(0, steps, group)]
    tr.submit_images(groups[0])
    got = []
    for i in range(steps):
        g, j = divmod(i, group)
        nxt = groups[g + 1] if g + 1 < len(groups) else None
        got.append(tr.xe_step_pending(caps[i], next_images=nxt if group > 1 else (xs[i + 1] if i + 1 < steps else None),
                                      training=False)['loss'].clone())
    sync()
    assert [float(v) for v in got] == want
    np.testing.assert_array_equal(tr.decoder.params.data.cpu().numpy(), ser.decoder.params.data.cpu().numpy())


@pytest.mark.parametrize('dtype', ['f32', 'bf16x3'])
def test_cnn_finetune_step_end_to_end(dtype):
    """train_mode cnn_finetune on a shallow stack: CNN forward -> decoder XE step -> CNN backward
    -> TF-Adam on decoder AND CNN variables, against the oracle chain (cnn_ref reverse pass fed by
    decoder_ref's input gradients, then adam_tf_update); two steps, so the second one runs on the
    refreshed weights (bf16x3: the [W_hi | W_hi | W_lo] forward copy and the backward-data filters repacked from the
    masters).  fp32 plan and the bf16x3 plan, both at the fp32 bar."""
    from comic_amd import trainer
    B, size, Lc = 3, 63, 9
    plan = nets.CnnPlan('chain', (size, size), layers=_CHAIN, x3=dtype == 'bf16x3')
    cnn_p = cnn_ref.randomize_bn(plan.init_params(seed=3), seed=4)
    Hf, Wf, Cf, _ = plan.buffers[plan.fm]
    spec, cfg = _spec_and_cfg(C=Cf, Cg=Cf, M=Hf * Wf, l2_decay=0.0)
    cfg.l2_decay = 0.0
    p = _rand_params(cfg, 3)
    rng = np.random.default_rng(17)
    x = rng.uniform(-1, 1, (B, size, size, 3)).astype(np.float32)
    _, _, caps = _batch(spec, B, Lc, 7)
    lr, eps = 1e-2, 1e-2
    tr = trainer.CaptionTrainer(cnn_p, spec, p, B, (size, size), dtype, DEV, lr_start=lr, lr_end=lr, max_step=10,
                                adam_epsilon=eps, plan=plan)
    tr.use_graph = False
    tr.enable_cnn_finetune()
    # oracle state
    names = sorted(k for k in cnn_p if k.endswith('weights') or k.endswith('beta'))
    ow = {k: cnn_p[k].copy() for k in cnn_p}
    om = {k: np.zeros_like(cnn_p[k]) for k in names}
    ov = {k: np.zeros_like(cnn_p[k]) for k in names}
    dm = {k: np.zeros_like(v) for k, v in p.items()}
    dv = {k: np.zeros_like(v) for k, v in p.items()}
    dp_ = {k: v.copy() for k, v in p.items()}
    for step in (1, 2):
        res = tr.finetune_step(dev(x), caps, training=False)
        sync()
        # the oracle walks the device's activations of THIS step (same masks / arg-maxima, see _device_activation_hook)
        n = cnn_ref._Net(ow, None, act_dtype='f32', run=True, tape=True,
                         override=_device_activation_hook(tr.encoder, 1e-3, [0.0]))
        n.scope.append('Chain')
        h = x
        for op in _CHAIN:
            h = n.conv(h, op[2], op[3], op[4], op[5], op[1]) if op[0] == 'c' else (
                n.max_pool(h, 3, 2, 'VALID') if op[0] == 'max' else n.avg_pool(h, 3, 1, 'SAME'))
        pooled = n.avg_pool(h, (h.shape[1], h.shape[2]), 1, 'VALID')
        out = dr.train_forward(dp_, cfg, h.reshape(B, Hf * Wf, Cf), pooled.reshape(B, Cf), caps, None, None)
        grads, dfm, dim = dr.train_backward(dp_, cfg, out)
        assert abs(float(res['loss']) - float(out['xe'])) <= F32_RTOL * abs(float(out['xe'])) + 1e-6, step
        g = n.backward([(pooled, dim.reshape(pooled.shape)), (h, dfm.reshape(h.shape))])
        for k in names:
            dr.adam_tf_update(ow[k], np.asarray(g[k], np.float32), om[k], ov[k], step, lr, eps=eps)
        for k in grads:
            dr.adam_tf_update(dp_[k], grads[k], dm[k], dv[k], step, lr, eps=eps)
    got = tr.encoder.export_params()
    for k in names:
        # compare the UPDATE (w - w0): the variables themselves barely move in two steps
        assert_close(got[k] - cnn_p[k], ow[k] - cnn_p[k], F32_RTOL, 'finetune update ' + k)
    gd = tr.decoder.params.to_numpy()
    # (W_q moves by 7e-6 on values of 0.1: the subtraction alone carries 1e-3 of fp32 rounding; the bf16x3 features add their
    # 5e-6 -- 1.08e-3 seen once)
    for k in ('K', 'W_q', 'W_m', 'W_o'):
        assert_close(gd[k] - p[k], dp_[k] - p[k], F32_RTOL if dtype == 'f32' else 2e-3, 'decoder update ' + k)


def test_known_answer_param_count_on_device():
    """README.md:222 '4.3 M' -> 4 297 987 (SURVEY §8 a-P); InceptionV3 variant 5 707 011."""
    s1 = cdec.DecoderSpec(C=832, Cg=1024, M=196)
    s3 = cdec.DecoderSpec()
    assert sum(int(np.prod(v)) if v else 1 for v in s1.param_shapes().values()) == 4297987
    assert sum(int(np.prod(v)) if v else 1 for v in s3.param_shapes().values()) == 5707011


@pytest.mark.parametrize('kw', [dict(), dict(fm_projection=None, H=1, token_type='word', V=300,
                                            init_method='project_hidden', start_id=298, end_id=299),
                                # large vocabulary: the beam step runs split over several workgroups per entry
                                dict(fm_projection=None, H=1, token_type='word', V=9000,
                                     init_method='project_hidden', start_id=8998, end_id=8999),
                                # Inception-V1 Mixed_4f (M = 196): the attention step in its split form
                                dict(C=832, Cg=1024, M=196),
                                # context layer (attention state = W_a ctx), independent value projection, sigmoid
                                dict(fm_projection='independent', context_layer=True, prob='sigmoid', H=4),
                                dict(fm_projection=None, context_layer=True, method='dot', H=4),
                                # --rnn_name LN_LSTM / GRU
                                dict(rnn_name='LN_LSTM'), dict(rnn_name='GRU'),
                                dict(rnn_name='GRU', init_method='project_hidden', context_layer=True)])
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
    # the loop ends on the device once every row has emitted EOS: a strong EOS bias ends it after 1-3 steps; the
    # executed prefix must equal the oracle's (which stops there too), called eagerly and replayed from the graph
    pe = dict(p); pe['b_o'] = p['b_o'].copy(); pe['b_o'][spec.end_id] = 9.0
    dece = cdec.Decoder(spec, pe, DEV)
    e_ids, e_logits, e_map = beam_ref.greedy_decode(pe, cfg, fm, im, max_steps)
    assert e_ids.shape[1] < max_steps, 'the early-exit case did not exit early'
    for _ in range(3):
        ids2, amap2, logits2 = dece.greedy(dev(fm), dev(im), max_steps, want_logits=True)
        np.testing.assert_array_equal(ids2, e_ids)
        assert_close(logits2.cpu().numpy(), e_logits, F32_RTOL, 'greedy logits (early exit)')
        assert_close(amap2.cpu().numpy(), e_map, F32_RTOL, 'greedy attention maps (early exit)')
    # the SCST step's non-draining fetch of the beam ids, with the greedy loop enqueued behind it and fetched later
    fb = dec.beam_search_ids(dev(fm), dev(im), 3, max_steps)
    fg = dec.greedy(dev(fm), dev(im), max_steps, defer=True)
    np.testing.assert_array_equal(fb(), dec.beam_search(dev(fm), dev(im), 3, max_steps, want_attention=False)['predicted_ids'])
    np.testing.assert_array_equal(fg()[0], g_ids)
    for W, eos_bias in ((3, 1.5), (7, 1.5), (3, 9.0)):
        if eos_bias != 1.5:
            # every beam ends within a few steps: the remaining launches of the fixed-length loop return at once
            # (device-side early exit, common.h ComicStop) and the outputs are those of the executed steps
            p = dict(p); p['b_o'] = p['b_o'].copy(); p['b_o'][spec.end_id] = eos_bias
            dec = cdec.Decoder(spec, p, DEV)
        pred, scores, hist, dbg = beam_ref.beam_search_decode(p, cfg, fm, im, W, max_steps, return_debug=True)
        res = dec.beam_search(dev(fm), dev(im), W, max_steps)
        if eos_bias != 1.5:
            assert res['step_ids'].shape[0] < max_steps
        np.testing.assert_array_equal(res['step_ids'], dbg['step_ids'])
        np.testing.assert_array_equal(res['parent_ids'], dbg['parent_ids'])
        np.testing.assert_array_equal(res['predicted_ids'], pred)              # bit-exact beam ids
        np.testing.assert_array_equal(res['lengths'], dbg['lengths'])
        fin = np.isfinite(scores)
        assert_close(np.where(fin, res['scores'], 0), np.where(fin, scores, 0), 1e-4, 'beam scores')
        assert_close(res['attn_hist'], hist, F32_RTOL, 'beam alignment history')


@pytest.mark.parametrize('B,W,V,D', [(5, 3, 9000, 128), (50, 3, 8962, 128), (32, 8, 4300, 256), (7, 5, 25599, 512),
                                     (20, 3, 25599, 512)])
def test_beam_streaming_logits_step_matches_oracle(B, W, V, D, monkeypatch):
    """Large-vocabulary beam step as the streaming projection + per-chunk top-k launch and its merge
    (csrc/beam_logits.hip) and, above 32 rows, the LSTM step as one streaming pass over the packed kernel
    (csrc/lstm_stream.hip), against the oracle and against the GEMM + statistics + top-k launches with the
    per-row-tile LSTM kernel: ids, parents and lengths bit-exact, scores at 1e-4.  Row counts on one and on two 16-row tiles per wave, a last chunk with fewer
    live columns than the beam width (V = 8962 = 70 * 128 + 2), beam 8, and the bench vocabulary at D = 512 with
    waves that only feed the LDS ring (35 and 60 rows: a counted wait that trusted LDS-DMA and register loads to retire in
    issue order gave about one wrong workgroup per launch exactly there)."""
    spec, cfg = _spec_and_cfg(fm_projection=None, H=1, token_type='word', V=V, D=D, init_method='project_hidden',
                              start_id=V - 2, end_id=V - 1)
    p = _rand_params(cfg, 9)
    fm, im, _ = _batch(spec, B, 6, 23)
    max_steps = 10
    for eos_bias in (1.5, 9.0):
        pe = dict(p); pe['b_o'] = p['b_o'].copy(); pe['b_o'][spec.end_id] = eos_bias
        dec = cdec.Decoder(spec, pe, DEV)
        res = None
        for _ in range(3):                                # eager, captured, replayed
            res = dec.beam_search(dev(fm), dev(im), W, max_steps)
        dec.beam_search(dev(fm), dev(im), W, max_steps, use_graph=False)
        # the streaming launches really ran: bit 0 the logits step, bit 1 the LSTM step (more than 32 rows)
        assert dec.lib.comic_decoder_beam_path() == (3 if B * W > 32 else 1)
        monkeypatch.setenv('COMIC_BEAM_LOGITS', '0')
        monkeypatch.setenv('COMIC_LSTM_STREAM', '0')
        dec0 = cdec.Decoder(spec, pe, DEV)
        res0 = dec0.beam_search(dev(fm), dev(im), W, max_steps, use_graph=False)
        assert dec0.lib.comic_decoder_beam_path() == 0
        monkeypatch.delenv('COMIC_BEAM_LOGITS')
        monkeypatch.delenv('COMIC_LSTM_STREAM')
        for k in ('step_ids', 'parent_ids', 'predicted_ids', 'lengths'):
            np.testing.assert_array_equal(res[k], res0[k], err_msg=k)
        fin = np.isfinite(res0['scores'])
        assert_close(np.where(fin, res['scores'], 0), np.where(fin, res0['scores'], 0), 1e-4, 'beam scores vs the GEMM path')
        if B * W * V * max_steps <= 150 * 9000 * 10:
            pred, scores, hist, dbg = beam_ref.beam_search_decode(pe, cfg, fm, im, W, max_steps, return_debug=True)
            np.testing.assert_array_equal(res['step_ids'], dbg['step_ids'])
            np.testing.assert_array_equal(res['parent_ids'], dbg['parent_ids'])
            np.testing.assert_array_equal(res['predicted_ids'], pred)
            np.testing.assert_array_equal(res['lengths'], dbg['lengths'])
            fin = np.isfinite(scores)
            assert_close(np.where(fin, res['scores'], 0), np.where(fin, scores, 0), 1e-4, 'beam scores')


def test_variational_recurrent_dropout_masks():
    """--rnn_recurr_dropout (model_base.py:645; DropoutWrapper(variational_recurrent=True), [TF-1.9] noise of shape
    [1, size]): the generated input / output masks are ONE row for every batch row and time step, the init call uses the
    input row, the attention dropout stays per step; two steps draw different rows; the step itself is the one the
    injected-mask oracle tests cover (loss and gradients finite, equal to a re-run with the same masks injected)."""
    spec, cfg = _spec_and_cfg(D=128, E=64)
    spec.recurrent_dropout = True
    B = 6
    fm, im, caps = _batch(spec, B, 9, 4)
    dec = cdec.Decoder(spec, _rand_params(cfg, 2), DEV)
    r1 = dec.train_step(dev(fm), dev(im), caps, training=True, use_graph=False)
    ctx = [c for c in dec._ctx.values() if c.masks is not None][0]
    m = {k: v.clone() for k, v in ctx.masks.items()}
    for k in ('inp', 'out'):
        row = m[k][0, 0]
        assert bool((m[k] == row).all()), k
        assert 0 < float((row == 0).float().mean()) < 1          # a real Bernoulli row
    assert bool((m['init_in'] == m['inp'][0, 0]).all())
    assert not bool((m['alpha'] == m['alpha'][0, 0]).all())
    g1 = dec.grads.data.clone()
    r2 = dec.train_step(dev(fm), dev(im), caps, masks={k: v for k, v in m.items()}, training=True, use_graph=False)
    assert abs(r1['loss'] - r2['loss']) <= 1e-6 * abs(r2['loss']) and bool(torch.equal(g1, dec.grads.data))
    dec.train_step(dev(fm), dev(im), caps, training=True, use_graph=False)
    assert not bool((ctx.masks['inp'][0, 0] == m['inp'][0, 0]).all())
    for _ in range(3):                                             # captured and replayed: the broadcast is part of the graph
        r3 = dec.train_step(dev(fm), dev(im), caps, training=True, use_graph=True)
        assert np.isfinite(float(r3['loss']))
        ctxg = [c for c in dec._ctx.values() if c.masks is not None][0]
        assert bool((ctxg.masks['out'] == ctxg.masks['out'][0, 0]).all())
        assert bool((ctxg.masks['init_in'] == ctxg.masks['inp'][0, 0]).all())


@pytest.mark.parametrize('kw,w', [(dict(), 0.7), (dict(fm_projection=None, H=1, token_type='word', V=300,
                                                     init_method='project_hidden', start_id=298, end_id=299), 1.0),
                                  (dict(C=832, Cg=1024, M=196), -0.5)])
def test_beam_length_penalty_matches_oracle(kw, w):
    """rnn_decoder_beam_search with length_penalty_weight != 0 (ops_rnn.py:96, infer.py:65): candidates ranked by
    total / ((5 + length) / 6)^w, EOS and finished beams do not grow; ids / parents / lengths bit-exact against the
    oracle, penalised scores at 1e-4; and a different result than without the penalty (the option does something)."""
    spec, cfg = _spec_and_cfg(**kw)
    p = _rand_params(cfg, 5)
    p['b_o'][spec.end_id] = 3.0                 # EOS competitive from the first steps: beams of different lengths
    B, max_steps = 5, 14
    fm, im, _ = _batch(spec, B, 6, 21)
    dec = cdec.Decoder(spec, p, DEV)
    plain = dec.beam_search(dev(fm), dev(im), 3, max_steps)
    differs = False
    for W in (3, 7):
        pred, scores, hist, dbg = beam_ref.beam_search_decode(p, cfg, fm, im, W, max_steps, return_debug=True,
                                                              length_penalty_weight=w)
        for _ in range(3):
            res = dec.beam_search(dev(fm), dev(im), W, max_steps, length_penalty_weight=w)
        np.testing.assert_array_equal(res['step_ids'], dbg['step_ids'])
        np.testing.assert_array_equal(res['parent_ids'], dbg['parent_ids'])
        np.testing.assert_array_equal(res['predicted_ids'], pred)
        np.testing.assert_array_equal(res['lengths'], dbg['lengths'])
        fin = np.isfinite(scores) & (np.abs(scores) < 1e30)
        assert_close(np.where(fin, res['scores'], 0), np.where(fin, scores, 0), 1e-4, 'penalised beam scores')
        if W == 3:
            differs = res['step_ids'].shape != plain['step_ids'].shape or (res['step_ids'] != plain['step_ids']).any() \
                or not np.allclose(res['scores'], plain['scores'])
    assert differs, 'the length penalty changed nothing'


@pytest.mark.parametrize('B,W,V', [(20, 3, 25599), (7, 5, 25599), (50, 3, 25599), (32, 7, 258)])
def test_beam_streaming_step_race_screen(B, W, V, monkeypatch):
    """Race screen of the hand-synchronised streaming kernels (LDS-DMA rings with counted / drained waits, cross-workgroup
    completion counters): 40 decodes of one input, eager and replayed, must all equal the round-2 launch chain's result.
    A wait that is one request short shows as a rare wrong workgroup (about 1 of 200 per launch when it existed), which
    a single comparison passes most of the time."""
    if V == 258:
        spec, cfg = _spec_and_cfg(D=512, E=256, C=2048, Cg=2048)
    else:
        spec, cfg = _spec_and_cfg(fm_projection=None, H=1, token_type='word', V=V, D=512, init_method='project_hidden',
                                  start_id=V - 2, end_id=V - 1)
    p = _rand_params(cfg, 9)
    p['b_o'][spec.end_id] = 1.5
    fm, im, _ = _batch(spec, B, 6, 23)
    monkeypatch.setenv('COMIC_BEAM_LOGITS', '0')
    monkeypatch.setenv('COMIC_LSTM_STREAM', '0')
    ref = cdec.Decoder(spec, p, DEV).beam_search(dev(fm), dev(im), W, 8, use_graph=False, want_attention=False)
    monkeypatch.delenv('COMIC_BEAM_LOGITS')
    monkeypatch.delenv('COMIC_LSTM_STREAM')
    dec = cdec.Decoder(spec, p, DEV)
    for i in range(40):
        res = dec.beam_search(dev(fm), dev(im), W, 8, use_graph=(i % 4 != 0), want_attention=False)
        for k in ('step_ids', 'parent_ids', 'lengths'):
            np.testing.assert_array_equal(res[k], ref[k], err_msg='%s, call %d' % (k, i))


@pytest.mark.parametrize('B,W,kw', [(32, 7, dict(D=512, E=256, C=2048, Cg=2048)), (5, 8, dict()), (9, 4, dict(V=1000, H=4))])
def test_beam_small_vocabulary_step_matches_launch_chain(B, W, kw, monkeypatch):
    """Small-vocabulary beam step with a beam's logits in a wave's registers (beam_step_small_kernel: radix-256, the SCST
    rollout geometry 32 x 7 at COMIC-256 size; beam 8; V = 1000) against comic_beam_step's kernel + the all-finished
    launch: ids, parents, lengths bit-exact, scores at 1e-5; early exit at the same step; eager, captured, replayed.
    (The oracle comparison of this path is test_greedy_and_beam_match_oracle.)"""
    spec, cfg = _spec_and_cfg(**kw)
    p = _rand_params(cfg, 11)
    fm, im, _ = _batch(spec, B, 6, 29)
    max_steps = 12
    for eos_bias in (1.5, 9.0):
        pe = dict(p); pe['b_o'] = p['b_o'].copy(); pe['b_o'][spec.end_id] = eos_bias
        dec = cdec.Decoder(spec, pe, DEV)
        for _ in range(3):
            res = dec.beam_search(dev(fm), dev(im), W, max_steps)
        dec.beam_search(dev(fm), dev(im), W, max_steps, use_graph=False)
        assert dec.lib.comic_decoder_beam_path() & 4
        monkeypatch.setenv('COMIC_BEAM_LOGITS', '0')
        dec0 = cdec.Decoder(spec, pe, DEV)
        res0 = dec0.beam_search(dev(fm), dev(im), W, max_steps, use_graph=False)
        assert not dec0.lib.comic_decoder_beam_path() & 4
        monkeypatch.delenv('COMIC_BEAM_LOGITS')
        for k in ('step_ids', 'parent_ids', 'predicted_ids', 'lengths'):
            np.testing.assert_array_equal(res[k], res0[k], err_msg=k)
        fin = np.isfinite(res0['scores'])
        assert_close(np.where(fin, res['scores'], 0), np.where(fin, res0['scores'], 0), 1e-5, 'beam scores vs comic_beam_step')
        assert_close(res['attn_hist'], res0['attn_hist'], 1e-5, 'alignment history')


@pytest.mark.parametrize('kw,B', [(dict(D=512, E=256), 5), (dict(D=512, E=128, method='dot', H=4, M=7), 37),
                                  (dict(D=512, E=64, fm_projection='independent', prob='sigmoid', H=16), 64)])
def test_persistent_greedy_loop_matches_oracle(kw, B, monkeypatch):
    """Greedy decoding as ONE persistent launch (decoder_persist.hip, GREEDY: logit columns and their maxima in the
    query phase, ids reduced by every workgroup, x third from the embedding table) against the oracle and against the
    per-step launches: ids bit-exact, logits / attention maps at 1e-3; with a strong EOS bias the loop leaves early in
    every group at the same step and the executed prefix is the oracle's; eager, captured and replayed."""
    spec, cfg = _spec_and_cfg(**kw)
    p = _rand_params(cfg, 5)
    p['b_o'][spec.end_id] = 1.5
    fm, im, _ = _batch(spec, B, 6, 21)
    max_steps = 12
    for eos_bias in (1.5, 9.0):
        pe = dict(p); pe['b_o'] = p['b_o'].copy(); pe['b_o'][spec.end_id] = eos_bias
        g_ids, g_logits, g_map = beam_ref.greedy_decode(pe, cfg, fm, im, max_steps)
        if eos_bias == 9.0:
            assert g_ids.shape[1] < max_steps, 'the early-exit case did not exit early'
        dec = cdec.Decoder(spec, pe, DEV)
        for _ in range(3):
            ids, amap, logits = dec.greedy(dev(fm), dev(im), max_steps, want_logits=True)
            np.testing.assert_array_equal(ids, g_ids)
            assert_close(logits.cpu().numpy(), g_logits, F32_RTOL, 'greedy logits')
            assert_close(amap.cpu().numpy(), g_map, F32_RTOL, 'greedy attention maps')
        dec.greedy(dev(fm), dev(im), max_steps, use_graph=False)
        assert dec.lib.comic_decoder_greedy_path() == 1            # the persistent loop really ran
        monkeypatch.setenv('COMIC_PERSIST', '0')
        ids0, amap0, logits0 = cdec.Decoder(spec, pe, DEV).greedy(dev(fm), dev(im), max_steps, want_logits=True)
        monkeypatch.delenv('COMIC_PERSIST')
        np.testing.assert_array_equal(ids0, ids)
        assert_close(logits.cpu().numpy(), logits0.cpu().numpy(), 2e-5, 'persistent vs per-step logits')
        assert_close(amap.cpu().numpy(), amap0.cpu().numpy(), 2e-5, 'persistent vs per-step attention maps')


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


def test_fullsize_word_baseline_beam3_properties():
    """BASELINE configs[4] at full size (word tokens, V = 25 599, 1 head, no feature-map projection, batch 50, beam 3,
    30 steps) -- beyond what the oracle finishes in seconds, so size-independent properties: ids inside the vocabulary,
    the final beams ordered by score, parents inside the beam, beam-1 == greedy up to the first EOS, determinism."""
    V = 25599
    spec, cfg = _spec_and_cfg(D=512, E=256, C=2048, Cg=2048, V=V, H=1, fm_projection=None, token_type='word',
                              start_id=V - 2, end_id=V - 1)
    p = _rand_params(cfg, 9)
    p['b_o'][spec.end_id] = 2.5
    B, steps = 50, 30
    fm, im, _ = _batch(spec, B, 6, 33)
    dec = cdec.Decoder(spec, p, DEV)
    r = dec.beam_search(dev(fm), dev(im), 3, steps, want_attention=False)
    T = r['predicted_ids'].shape[0]
    assert 1 <= T <= steps and r['predicted_ids'].shape[1:] == (B, 3)
    assert r['predicted_ids'].min() >= 0 and r['predicted_ids'].max() < V
    assert r['parent_ids'].min() >= 0 and r['parent_ids'].max() < 3
    sc = r['scores'][T - 1]                                            # cumulative log-probabilities of the final beams
    assert np.all(np.diff(sc, axis=1) <= 1e-6), 'beams are not ordered by score'
    fin = np.isfinite(r['scores'])
    assert np.all(np.diff(np.where(fin, r['scores'], -1e30)[:, :, 0], axis=0) <= 1e-5), 'best score must not increase'
    r2 = dec.beam_search(dev(fm), dev(im), 3, steps, want_attention=False)
    np.testing.assert_array_equal(r['predicted_ids'], r2['predicted_ids'])
    g_ids, _, _ = dec.greedy(dev(fm), dev(im), steps)
    b1 = dec.beam_search(dev(fm), dev(im), 1, steps, want_attention=False)['predicted_ids'][:, :, 0].T     # [B, T1]
    for b in range(B):
        eos = np.flatnonzero(g_ids[b] == spec.end_id)
        n = min((eos[0] + 1) if len(eos) else g_ids.shape[1], b1.shape[1], g_ids.shape[1])
        np.testing.assert_array_equal(b1[b, :n], g_ids[b, :n], err_msg='row %d' % b)


def test_fullsize_scst_step_properties():
    """BASELINE configs[3] at full size: the reward-weighted step on 32 x 7 = 224 hypothesis rows (COMIC-256, 5x5x2048
    map).  Deterministic bit for bit; the gradient is LINEAR in the rewards (model_base.py:342-347: mean_b(xent_b *
    reward_b); map loss off here); rows with reward 0 contribute nothing; every gradient finite."""
    spec, cfg = _spec_and_cfg(D=512, E=256, C=2048, Cg=2048, map_loss_scale=0.0)
    rows = 224
    fm, im, caps = _batch(spec, rows, 22, 5)
    rng = np.random.default_rng(6)
    r1 = rng.standard_normal(rows).astype(np.float32)
    r2 = rng.standard_normal(rows).astype(np.float32)
    dec = cdec.Decoder(spec, _rand_params(cfg, 2), DEV)
    dec.spec.l2_decay = 0.0

    def grads(r):
        dec.train_step(dev(fm), dev(im), caps, rewards=r, training=False)
        return dec.grads.data.clone()
    ga, gb, gab = grads(r1), grads(r2), grads(r1 + r2)
    assert torch.equal(ga, grads(r1)), 'the SCST step is not deterministic'
    assert torch.isfinite(gab).all()
    den = float(gab.abs().max())
    assert float((ga + gb - gab).abs().max()) <= 1e-4 * den, 'gradient is not linear in the rewards'
    half = r1.copy(); half[112:] = 0
    g_half = grads(half)
    dec.train_step(dev(fm[:112]), dev(im[:112]), caps[:112], rewards=r1[:112] * np.float32(112 / 224), training=False)
    assert float((g_half - dec.grads.data).abs().max()) <= 1e-4 * float(g_half.abs().max()), 'zero-reward rows leak'


def test_fullsize_cnn_finetune_step_properties():
    """BASELINE configs[2] at its per-GPU size (batch 32, 224 x 224, bf16 activations, fp32 masters): two trainers from
    the same state take the same step (fp32 atomics in dW: 1e-5), the loss falls over three steps on a fixed batch,
    every updated variable is finite and has moved."""
    from comic_amd import trainer
    B = 32
    cnn_p = cnn_ref.randomize_bn(cnn_ref.init_params(0, 224), seed=1)
    spec, cfg = _spec_and_cfg(D=512, E=256, C=2048, Cg=2048)
    p = _rand_params(cfg, 4)
    rng = np.random.default_rng(8)
    x = dev(rng.uniform(-1, 1, (B, 224, 224, 3)).astype(np.float32))
    _, _, caps = _batch(spec, B, 24, 9)
    outs = []
    for rep in range(2):
        tr = trainer.CaptionTrainer(cnn_p, spec, p, B, (224, 224), 'bf16', DEV, lr_start=1e-3, lr_end=1e-3, max_step=10)
        tr.use_graph = False
        tr.enable_cnn_finetune()
        losses = [float(tr.finetune_step(x, caps, training=False)['loss']) for _ in range(3)]
        sync()
        outs.append((losses, tr.encoder.w_master.data.clone(), tr.decoder.params.data.clone()))
        del tr
    (l0, w0, d0), (l1, w1, d1) = outs
    assert all(np.isfinite(l0)) and l0[2] < l0[0], l0
    # (the order of the fp32 atomics in dW differs run to run at 1e-8; where that flips the bf16 rounding of a refreshed weight the
    # third step's loss moves by 2e-6 relative, the variables by 3e-6 / 1e-5 absolute: tools/ft_repro.py, 40 pairs.  Differences
    # of 1e-4 ... 1e-3 here were a RACE of the two chain lanes of the scheduled backward -- the head's global pool inside the
    # last block's fork region -- that this bound caught once the lanes really ran side by side.)
    assert abs(l0[2] - l1[2]) <= 1e-4 * abs(l0[2])
    assert torch.isfinite(w0).all() and torch.isfinite(d0).all()
    assert float((w0 - w1).abs().max()) <= 1e-4 * float(w0.abs().max())
    assert float((d0 - d1).abs().max()) <= 1e-4 * float(d0.abs().max())
    w_init = nets.CnnEncoder(nets.CnnPlan('inception_v3', (224, 224)), cnn_p, 1, 'bf16', DEV).w_master.data
    assert float((w0 - w_init).abs().max()) > 0


@pytest.mark.gpu
@pytest.mark.parametrize('kw', [dict(), dict(fm_projection=None, H=1, token_type='word', V=9000,
                                             init_method='project_hidden', start_id=8998, end_id=8999)])
def test_sampled_decode_matches_oracle_on_the_same_noise(kw):
    """rnn_decoder_search(greedy_search=False) (ops_rnn.py:158-166, SampleEmbeddingHelper): with the SAME Gumbel noise the
    device's draws equal the oracle's id for id (argmax of logits + noise); the returned logits stay the undisturbed
    projection (the distribution of the draws is checked on the oracle, tests/test_oracle_decoder.py)."""
    spec, cfg = _spec_and_cfg(**kw)
    p = _rand_params(cfg, 5)
    p['b_o'][spec.end_id] = 1.5
    B, max_steps = 5, 12
    fm, im, _ = _batch(spec, B, 6, 21)
    dec = cdec.Decoder(spec, p, DEV)
    rng = np.random.default_rng(3)
    gum = (-np.log(-np.log(rng.uniform(1e-9, 1.0, (max_steps, B, spec.V))))).astype(np.float32)
    o_ids, o_logits, o_map = beam_ref.greedy_decode(p, cfg, fm, im, max_steps, gumbel=gum)
    ids, amap, logits = dec.sample(dev(fm), dev(im), max_steps, noise=dev(gum), want_logits=True)
    np.testing.assert_array_equal(ids, o_ids)
    assert_close(logits.cpu().numpy(), o_logits, F32_RTOL, 'sampled-decode logits')
    assert_close(amap.cpu().numpy(), o_map, F32_RTOL, 'sampled-decode attention maps')
    g_ids, _, _ = beam_ref.greedy_decode(p, cfg, fm, im, max_steps)
    assert not np.array_equal(o_ids[:, :g_ids.shape[1]], g_ids[:, :o_ids.shape[1]]), 'the noise never changed a token'
    # generated noise: different seeds give different rollouts, one seed the same rollout
    a1, _, _ = dec.sample(dev(fm), dev(im), max_steps, seed=1)
    a2, _, _ = dec.sample(dev(fm), dev(im), max_steps, seed=1)
    b1, _, _ = dec.sample(dev(fm), dev(im), max_steps, seed=2)
    np.testing.assert_array_equal(a1, a2)
    assert a1.shape != b1.shape or not np.array_equal(a1, b1)






def test_two_beam_searches_in_flight_give_the_ids_of_one_at_a_time():
    """Decoder.beam_search_ids(slot=): two buffer sets let the decode loops of two batches run on two streams at once
    (CaptionModel.infer_pipelined); ids, batch by batch, equal the one-at-a-time loop -- word vocabulary (streaming logits)
    and radix vocabulary, several rounds so both slots replay their captured graphs."""
    for spec, B, W in ((cdec.DecoderSpec(V=25599, H=1, fm_projection=None, token_type='word', start_id=25597, end_id=25598), 6, 3),
                       (cdec.DecoderSpec(), 5, 3)):
        dec = cdec.Decoder(spec, None, DEV, seed=12)
        rng = np.random.default_rng(3)
        feats = [(dev(rng.standard_normal((B, spec.M, spec.C)).astype(np.float32)),
                  dev(rng.standard_normal((B, spec.Cg)).astype(np.float32))) for _ in range(6)]
        want = [dec.beam_search_ids(fm, im, W, 12)() for fm, im in feats]
        lanes = [torch.cuda.Stream(), torch.cuda.Stream()]
        pend, got = [None, None], []
        for i, (fm, im) in enumerate(feats):
            k = i % 2
            if pend[k] is not None:
                got.append(pend[k]())
            lanes[k].wait_stream(torch.cuda.current_stream())
            with torch.cuda.stream(lanes[k]):
                pend[k] = dec.beam_search_ids(fm, im, W, 12, slot=k)
        n = len(feats)
        got.append(pend[n % 2]())
        got.append(pend[1 - n % 2]())
        sync()
        assert len(got) == len(want)
        for a, b in zip(got, want):
            np.testing.assert_array_equal(a, b)
