"""Data-parallel path on CPU: two processes over gloo run the DataParallel helper (row
sharding, global token count, flat-gradient all-reduce) with the oracle standing in for the
GPU step; the rank-mean gradient must equal the single-process gradient of the global batch
(XE is normalised by the GLOBAL non-pad token count)."""
import os
import socket

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

from oracle import decoder_ref as dr


def _free_port():
    s = socket.socket()
    s.bind(('127.0.0.1', 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _case():
    cfg = dr.DecoderConfig(rnn_size=32, rnn_word_size=16, attn_num_heads=4, softmax_size=18, fm_channels=24,
                           im_embed_size=24, radix_base=16, start_id=16, end_id=17, l2_decay=0.0,
                           rnn_map_loss_scale=0.0)
    rng = np.random.default_rng(0)
    B, M, L = 8, 5, 9
    fm = rng.standard_normal((B, M, 24)); im = rng.standard_normal((B, 24))
    caps = np.full((B, L), -1, np.int64)
    for b in range(B):
        n = int(rng.integers(2, L - 1))
        caps[b, 0] = 16; caps[b, 1:1 + n] = rng.integers(0, 16, n); caps[b, 1 + n] = 17
    p = dr.init_params(cfg, 3, np.float64)
    return cfg, p, fm, im, caps


def _worker(rank, world, port, ret):
    os.environ.update(MASTER_ADDR='127.0.0.1', MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world))
    dist.init_process_group('gloo', rank=rank, world_size=world)
    from comic_amd.trainer import DataParallel
    dp = DataParallel(dist)
    cfg, p, fm, im, caps = _case()
    lo, hi = dp.shard(fm.shape[0])
    assert (lo, hi) == (rank * 4, rank * 4 + 4)
    out = dr.train_forward(p, cfg, fm[lo:hi], im[lo:hi], caps[lo:hi])
    grads, _, _ = dr.train_backward(p, cfg, out)
    local_tokens = float((caps[lo:hi, 1:] >= 0).sum())
    g_tokens = dp.global_tokens(local_tokens, 'cpu')
    denom = g_tokens / world + 1e-12                         # what Decoder.train_step(xe_denom=...) uses
    flat = torch.from_numpy(np.concatenate([(grads[k] * (local_tokens + 1e-12) / denom).reshape(-1) for k in sorted(grads)]))
    scale = dp.average_(flat)
    flat = flat * scale
    if rank == 0:
        ret['flat'] = flat.numpy()
        ret['tokens'] = g_tokens
    dist.barrier()
    dist.destroy_process_group()


def _bucket_worker(rank, world, port, ret):
    """Bucketed exchange (reduce_async per bucket, last block first, + wait_all) against ONE flat all-reduce of the same
    buffers, and the device-side XE denominator, on CPU tensors over gloo."""
    os.environ.update(MASTER_ADDR='127.0.0.1', MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world))
    dist.init_process_group('gloo', rank=rank, world_size=world)
    from comic_amd import nets
    from comic_amd.decoder import FlatParams
    from comic_amd.trainer import DataParallel
    dp = DataParallel(dist)
    plan = nets.CnnPlan('inception_v3', (224, 224))
    ws, bs = nets.flat_layout(plan)
    W, Bt = FlatParams(ws, 'cpu'), FlatParams(bs, 'cpu')
    buckets = nets.plan_grad_buckets(plan, W, Bt, 6)
    g = torch.Generator().manual_seed(100 + rank)
    dw = torch.randn(W.numel, generator=g)
    db = torch.randn(Bt.numel, generator=g)
    flat_w, flat_b = dw.clone(), db.clone()
    dp.average_(flat_w)
    dp.average_(flat_b)
    for bk in buckets:                                    # the order the backward finishes them
        dp.reduce_async(dw[bk[2][0]:bk[2][1]])
        dp.reduce_async(db[bk[3][0]:bk[3][1]])
    scale = dp.wait_all()
    wm = torch.zeros(40)
    wm[:13 + 5 * rank] = 1.0
    den = dp.global_xe_denominator(wm)
    if rank == 0:
        ret['equal'] = bool(torch.equal(dw, flat_w) and torch.equal(db, flat_b))
        ret['scale'] = scale
        ret['den'] = float(den)
        ret['n_buckets'] = len(buckets)
    dist.barrier()
    dist.destroy_process_group()


def _chunk_worker(rank, world, port, ret):
    """The decoder's chunked gradient exchange (DataParallel.chunk_bounds: ranges on variable boundaries, the last range --
    with the status word behind the variables -- first) against ONE flat all-reduce of the same buffer, on CPU tensors over
    gloo; and the optimiser's range-by-range update order (`before_range` called once per range, in exchange order)."""
    os.environ.update(MASTER_ADDR='127.0.0.1', MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world))
    dist.init_process_group('gloo', rank=rank, world_size=world)
    from comic_amd.decoder import FlatParams, DecoderSpec
    from comic_amd.trainer import DataParallel
    dp = DataParallel(dist)
    G = FlatParams(DecoderSpec(M=25, C=2048, Cg=2048).param_shapes(), 'cpu', status_tail=True)
    g = torch.Generator().manual_seed(300 + rank)
    G.data.copy_(torch.randn(G.data.numel(), generator=g))
    G.data[G.numel] = float(rank)                      # the status word: rank 1 voids the step
    flat = G.data.clone()
    dp.average_(flat)
    bounds = dp.chunk_bounds(G, 4)
    tail = G.data.numel() - G.numel
    for j, (lo, hi) in enumerate(bounds[::-1]):
        dp.reduce_async(G.data[lo:hi + (tail if j == 0 else 0)])
    dp.wait_all()
    if rank == 0:
        ret['equal'] = bool(torch.equal(G.data, flat))
        ret['status'] = float(G.data[G.numel])
        ret['bounds'] = [tuple(b) for b in bounds]
        ret['numel'] = G.numel
        ret['cuts'] = sorted(G.offsets.values())
    dist.barrier()
    dist.destroy_process_group()


@pytest.mark.timeout(300)
def test_chunked_decoder_exchange_equals_flat_all_reduce():
    world, port = 2, _free_port()
    mgr = mp.Manager()
    ret = mgr.dict()
    mp.spawn(_chunk_worker, args=(world, port, ret), nprocs=world, join=True)
    assert ret['equal'] and ret['status'] == 1.0
    b = ret['bounds']
    assert 2 <= len(b) <= 4 and b[0][0] == 0 and b[-1][1] == ret['numel']
    assert all(b[i][1] == b[i + 1][0] for i in range(len(b) - 1)) and all(lo in ret['cuts'] for lo, _ in b)
    sizes = [hi - lo for lo, hi in b]
    assert max(sizes) < 0.6 * ret['numel']              # no chunk carries most of the buffer (K alone is 46 %)


@pytest.mark.timeout(300)
def test_bucketed_exchange_equals_flat_all_reduce():
    world, port = 2, _free_port()
    mgr = mp.Manager()
    ret = mgr.dict()
    mp.spawn(_bucket_worker, args=(world, port, ret), nprocs=world, join=True)
    assert ret['equal'] and ret['scale'] == 0.5 and 4 <= ret['n_buckets'] <= 6
    assert abs(ret['den'] - ((13 + 18) / 2 + 1e-12)) < 1e-9


@pytest.mark.timeout(300)
def test_rank_mean_gradient_equals_global_batch_gradient():
    world, port = 2, _free_port()
    mgr = mp.Manager()
    ret = mgr.dict()
    mp.spawn(_worker, args=(world, port, ret), nprocs=world, join=True)
    cfg, p, fm, im, caps = _case()
    out = dr.train_forward(p, cfg, fm, im, caps)
    grads, _, _ = dr.train_backward(p, cfg, out)
    ref = np.concatenate([grads[k].reshape(-1) for k in sorted(grads)])
    assert ret['tokens'] == float((caps[:, 1:] >= 0).sum())
    np.testing.assert_allclose(ret['flat'], ref, rtol=1e-9, atol=1e-12)


def test_single_process_dataparallel_is_identity():
    from comic_amd.trainer import DataParallel
    dp = DataParallel(None)
    assert dp.world == 1 and dp.shard(64) == (0, 64) and dp.global_tokens(5.0, 'cpu') == 5.0
    t = torch.ones(4)
    assert dp.average_(t) == 1.0 and torch.equal(t, torch.ones(4))
