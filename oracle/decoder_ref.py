"""CPU restatement (numpy) of the reference's attention-LSTM decoder: forward for
training (teacher forcing), losses, analytic backward, TF-form Adam, cosine LR.

TEST INFRASTRUCTURE -- see oracle/__init__.py.  PARITY UNPINNED (tensorflow==1.9.0
un-vendored): follows the reference call sites + TF-1.9 semantics;
cross-checked against an independent torch-CPU/autograd formulation in
tests/test_oracle_decoder.py.

Reference anchors
  _process_inputs ............ src/model_base.py:501-528
  keys / values .............. common/ops_rnn.py:440-477   (MultiHeadAttV3.__init__)
  LSTM cell + dropout ........ src/model_base.py:606-648   (BasicLSTMCell, DropoutWrapper)
  rnn init ................... src/model_base.py:651-689
  embeddings ................. src/model_base.py:557-594
  wrapper step ............... common/ops_rnn.py:660-755
  add_LN / dot score ......... common/ops_rnn.py:531-565, :611-632 ; common/ops.py:241-275
  training loop .............. common/ops_rnn.py:183-243   (impute_finished=True, last-step padding)
  post-process ............... src/model_base.py:272-314
  losses ..................... src/model_base.py:325-417 ; common/ops.py:184-190
  optimiser / LR ............. src/model_base.py:809-820, :852-883

Short parameter keys -> reference variable (scope Model/decoder/rnn_decoder/):
  W_init  rnn_init_input/projection/weight | rnn_initial_state/weight
  K, b    .../basic_lstm_cell/{kernel,bias}
          rnn_name LN_LSTM: .../layer_norm_basic_lstm_cell/kernel (no bias) and cln_{i,j,f,o,c}{g,b} =
          .../layer_norm_basic_lstm_cell/{input,transform,forget,output,state}/{gamma,beta}
          rnn_name GRU: K, b = .../gru_cell/gates/{kernel,bias}; K_c, b_c = .../gru_cell/candidate/{kernel,bias}
  W_m     memory_layer/kernel        W_v  value_layer/kernel (independent only)
  W_q     query_layer/kernel         v    attention_v
  ln_g, ln_b  LN_tanh/{gamma,beta}   tau  softmax_temperature
  W_a     a_layer/kernel (context layer only)
  W_o, b_o  output_projection/{kernel,bias}     emb  embedding_map
"""
from __future__ import annotations

import math
from dataclasses import dataclass, field

import numpy as np

LN_EPS = 1e-12  # tf.contrib.layers.layer_norm variance_epsilon
LN_LSTM_NORMS = ('i', 'j', 'f', 'o', 'c')     # input, transform, forget, output, state


@dataclass
class DecoderConfig:
    rnn_size: int = 512            # D
    rnn_word_size: int = 256       # E
    attn_num_heads: int = 8        # H
    cnn_fm_projection: str | None = 'tied'      # 'tied' | 'independent' | None
    attn_alignment_method: str = 'add_LN'       # 'add_LN' | 'dot'
    attn_probability_fn: str = 'softmax'        # 'softmax' | 'sigmoid'
    attn_context_layer: bool = False
    rnn_init_method: str = 'first_input'        # 'first_input' | 'project_hidden'
    token_type: str = 'radix'
    radix_base: int = 256
    softmax_size: int = 258        # V
    fm_channels: int = 2048        # C
    im_embed_size: int = 2048      # C_g
    dropout_rnn_in: float = 0.35
    dropout_rnn_out: float = 0.35
    attn_keep_prob: float = 0.9
    rnn_map_loss_scale: float = 1.0
    l2_decay: float = 1e-5
    start_id: int = 256
    end_id: int = 257
    rnn_name: str = 'LSTM'         # 'LSTM' | 'LN_LSTM' | 'GRU'   (model_base.py:606-632)

    @property
    def attn_size(self):           # A   (model_base.py:611-615)
        if self.cnn_fm_projection is None and not self.attn_context_layer:
            return self.fm_channels
        return self.rnn_size

    @property
    def value_channels(self):      # channels of the (un-split) values tensor
        return self.fm_channels if self.cnn_fm_projection is None else self.rnn_size


def xavier_uniform(rng, shape, dtype=np.float32):
    """slim.xavier_initializer() [TF-1.9]: U(+-sqrt(6/(fan_in+fan_out)))."""
    if len(shape) > 1:
        fan_in, fan_out = shape[-2], shape[-1]
    else:
        fan_in = fan_out = shape[-1]
    lim = math.sqrt(6.0 / (fan_in + fan_out))
    return rng.uniform(-lim, lim, shape).astype(dtype)


def init_params(cfg: DecoderConfig, seed=0, dtype=np.float32):
    rng = np.random.default_rng(seed)
    D, E, A, V, C, Cg = (cfg.rnn_size, cfg.rnn_word_size, cfg.attn_size, cfg.softmax_size,
                         cfg.fm_channels, cfg.im_embed_size)
    p = {}
    if cfg.rnn_init_method == 'first_input':
        p['W_init'] = xavier_uniform(rng, (Cg, E + A), dtype)
    else:
        p['W_init'] = xavier_uniform(rng, (Cg, D), dtype)
    if cfg.rnn_name == 'GRU':       # [TF-1.9] GRUCell: gates bias starts at 1.0, candidate bias at 0
        p['K'] = xavier_uniform(rng, (E + A + D, 2 * D), dtype)
        p['b'] = np.ones(2 * D, dtype)
        p['K_c'] = xavier_uniform(rng, (E + A + D, D), dtype)
        p['b_c'] = np.zeros(D, dtype)
    elif cfg.rnn_name == 'LN_LSTM':  # [TF-1.9] LayerNormBasicLSTMCell(layer_norm=True): no bias, gain 1 / shift 0
        p['K'] = xavier_uniform(rng, (E + A + D, 4 * D), dtype)
        for n in LN_LSTM_NORMS:
            p['cln_%sg' % n] = np.ones(D, dtype)
            p['cln_%sb' % n] = np.zeros(D, dtype)
    else:
        p['K'] = xavier_uniform(rng, (E + A + D, 4 * D), dtype)
        p['b'] = np.zeros(4 * D, dtype)
    p['W_m'] = xavier_uniform(rng, (C, D), dtype)
    if cfg.cnn_fm_projection == 'independent':
        p['W_v'] = xavier_uniform(rng, (C, D), dtype)
    p['W_q'] = xavier_uniform(rng, (D, D), dtype)
    if cfg.attn_alignment_method == 'add_LN':
        p['v'] = xavier_uniform(rng, (D,), dtype)
        p['ln_g'] = np.ones(D, dtype)
        p['ln_b'] = np.zeros(D, dtype)
        p['tau'] = np.array(5.0, dtype)
    if cfg.attn_context_layer:
        p['W_a'] = xavier_uniform(rng, (cfg.value_channels, D), dtype)
    p['W_o'] = xavier_uniform(rng, (D, V), dtype)
    p['b_o'] = np.zeros(V, dtype)
    p['emb'] = xavier_uniform(rng, (V, E), dtype)
    return p


def count_params(p):
    return int(sum(np.asarray(v).size for v in p.values()))


# --------------------------------------------------------------------------- #
# small numerics
# --------------------------------------------------------------------------- #
def sigmoid(x):
    return 1.0 / (1.0 + np.exp(-x))


def softmax(x, axis=-1):
    e = np.exp(x - x.max(axis=axis, keepdims=True))
    return e / e.sum(axis=axis, keepdims=True)


def log_softmax(x, axis=-1):
    z = x - x.max(axis=axis, keepdims=True)
    return z - np.log(np.exp(z).sum(axis=axis, keepdims=True))


def dropout(x, mask, keep):
    """tf.nn.dropout [TF-1.9]: div(x, keep_prob) * binary_tensor."""
    if mask is None:
        return x
    return (x / x.dtype.type(keep)) * mask


def process_inputs(captions, token_type='radix'):
    """model_base.py:501-528 -> (inputs [B,T], targets [B,T], masks [B,T], lens [B])."""
    sent = np.asarray(captions, np.int64)
    masks = np.sign((sent[:, 1:] + 1).astype(np.float32))
    lens = masks.sum(axis=1).astype(np.int32)
    if token_type == 'word':
        sent = np.maximum(sent, 0)
        inputs = sent[:, :-1]
    else:
        inputs = sent[:, :-1]
        sent = np.maximum(sent, 0)
    targets = sent[:, 1:]
    return inputs, targets, masks, lens


def embed(emb, ids):
    """one_hot(ids) @ embedding_map: negative ids give the zero vector
    (model_base.py:523-526, :587-593); word tokens are clipped upstream."""
    ids = np.asarray(ids)
    out = emb[np.maximum(ids, 0)]
    return out * (ids >= 0)[..., None].astype(emb.dtype)


def split_heads(x, H):
    """[B,L,C] -> [B,H,L,C/H]   (ops_rnn.py:246-261)."""
    B, L, C = x.shape
    return x.reshape(B, L, H, C // H).transpose(0, 2, 1, 3)


def memory_projections(p, cfg, fm):
    """keys [B,M,D] and un-split values [B,M,Cv]   (ops_rnn.py:440-477)."""
    keys = fm @ p['W_m']
    if cfg.cnn_fm_projection == 'tied':
        values = keys
    elif cfg.cnn_fm_projection == 'independent':
        values = fm @ p['W_v']
    else:
        values = fm
    return keys, values


def lstm_cell(p, xin, c, h):
    """BasicLSTMCell, forget_bias 1.0, gate order i,j,f,o."""
    D = c.shape[-1]
    g = np.concatenate([xin, h], axis=1) @ p['K'] + p['b']
    i, j, f, o = g[:, :D], g[:, D:2 * D], g[:, 2 * D:3 * D], g[:, 3 * D:]
    si, sf, so, tj = sigmoid(i), sigmoid(f + 1.0), sigmoid(o), np.tanh(j)
    c2 = c * sf + si * tj
    tc = np.tanh(c2)
    h2 = tc * so
    return c2, h2, (si, tj, sf, so, tc)


def ln_lstm_cell(p, xin, c, h):
    """tf.contrib.rnn.LayerNormBasicLSTMCell(num_units) [TF-1.9 contrib/rnn/python/ops/rnn_cell.py]: layer_norm=True,
    forget_bias 1.0, norm_gain 1 / norm_shift 0 initialisers, dropout_keep_prob 1: concat = [x,h] K (NO bias);
    i, j, f, o each through layers.layer_norm (scopes input / transform / forget / output); new_c =
    c*sigmoid(f + 1) + sigmoid(i)*tanh(j), then LN (scope state) -- the NORMALISED value is the new cell state --
    new_h = tanh(new_c) * sigmoid(o)."""
    D = c.shape[-1]
    g = np.concatenate([xin, h], axis=1) @ p['K']
    pre, xh, rs = [], [], []
    for k, n in enumerate('ijfo'):
        z = g[:, k * D:(k + 1) * D]
        y, mean, rstd = layer_norm_tf(z, p['cln_%sg' % n], p['cln_%sb' % n])
        pre.append(y); xh.append((z - mean) * rstd); rs.append(rstd)
    si, tj, sf, so = sigmoid(pre[0]), np.tanh(pre[1]), sigmoid(pre[2] + 1.0), sigmoid(pre[3])
    craw = c * sf + si * tj
    c2, mean, rstd = layer_norm_tf(craw, p['cln_cg'], p['cln_cb'])
    xh.append((craw - mean) * rstd); rs.append(rstd)
    tc = np.tanh(c2)
    return c2, tc * so, (si, tj, sf, so, tc, xh, rs)


def gru_cell(p, xin, c, h):
    """tf.contrib.rnn.GRUCell [TF-1.9 rnn_cell_impl.GRUCell.call]: [r,u] = sigmoid([x,h] W_g + b_g);
    cand = tanh([x, r*h] W_c + b_c); new_h = u*h + (1-u)*cand.  The state is h alone; `c` rides along untouched."""
    D = h.shape[-1]
    ru = sigmoid(np.concatenate([xin, h], axis=1) @ p['K'] + p['b'])
    r, u = ru[:, :D], ru[:, D:]
    cand = np.tanh(np.concatenate([xin, r * h], axis=1) @ p['K_c'] + p['b_c'])
    return c, u * h + (1 - u) * cand, (r, u, cand)


def rnn_cell(p, cfg, xin, c, h):
    return {'LSTM': lstm_cell, 'LN_LSTM': ln_lstm_cell, 'GRU': gru_cell}[cfg.rnn_name](p, xin, c, h)


def layer_norm_tf(z, g, b):
    """tf.contrib.layers.layer_norm [TF-1.9]: moments over the last axis (biased var),
    nn.batch_normalization(x, mean, var, beta, gamma, 1e-12) = x*inv + (beta - mean*inv),
    inv = rsqrt(var + eps) * gamma."""
    mean = z.mean(axis=-1, keepdims=True)
    var = ((z - mean) ** 2).mean(axis=-1, keepdims=True)
    rstd = 1.0 / np.sqrt(var + z.dtype.type(LN_EPS))
    inv = rstd * g
    return z * inv + (b - mean * inv), mean, rstd


def attention_scores(p, cfg, keys, q):
    """-> (alpha [B,H,M], cache).  MultiHeadAddLN / MultiHeadDot + probability fn."""
    B, M, D = keys.shape
    H = cfg.attn_num_heads
    cache = {}
    if cfg.attn_alignment_method == 'add_LN':
        z = keys + q[:, None, :]
        zhat, mean, rstd = layer_norm_tf(z, p['ln_g'], p['ln_b'])
        th = np.tanh(zhat)
        raw = (th * p['v']).reshape(B, M, H, D // H).sum(axis=3).transpose(0, 2, 1)  # [B,H,M]
        s = raw / p['tau']
        cache.update(z=z, mean=mean, rstd=rstd, th=th, raw=raw)
    elif cfg.attn_alignment_method == 'dot':
        raw = (keys * q[:, None, :]).reshape(B, M, H, D // H).sum(axis=3).transpose(0, 2, 1)
        s = raw / keys.dtype.type(math.sqrt(D / H))
        cache.update(raw=raw)
    else:
        raise ValueError('Invalid alignment method.')
    if cfg.attn_probability_fn == 'softmax':
        alpha = softmax(s, axis=-1)
    elif cfg.attn_probability_fn == 'sigmoid':           # model_base.py:599-603
        sg = sigmoid(s)
        alpha = sg / sg.sum(axis=-1, keepdims=True)
        cache['sg'] = sg
    else:
        raise ValueError('Invalid probability fn.')
    cache['s'] = s
    return alpha, cache


def context(cfg, alpha_d, values):
    """ctx[b, h*dv:(h+1)*dv] = sum_m alpha[b,h,m] * values[b,m,h*dv:(h+1)*dv]."""
    B, M, Cv = values.shape
    H = cfg.attn_num_heads
    vs = values.reshape(B, M, H, Cv // H)
    return np.einsum('bhm,bmhd->bhd', alpha_d, vs).reshape(B, Cv)


def rnn_init(p, cfg, im_embed, mask_in=None):
    """model_base.py:651-689."""
    B = im_embed.shape[0]
    D = cfg.rnn_size
    dt = im_embed.dtype
    cache = {}
    if cfg.rnn_init_method == 'project_hidden':
        h0 = im_embed @ p['W_init']
        c0 = np.zeros_like(h0)
    else:
        x = im_embed @ p['W_init']
        u = dropout(x, mask_in, 1.0 - cfg.dropout_rnn_in)
        c0, h0, gc = rnn_cell(p, cfg, u, np.zeros((B, D), dt), np.zeros((B, D), dt))
        cache.update(u=u, gates=gc)
    return c0, h0, cache


def decoder_step(p, cfg, keys, values, x_t, c, h, att, masks=None):
    """One MultiHeadAttentionWrapperV3.call (ops_rnn.py:660-755).
    masks: None (inference) or {'in': [B,E+A], 'out': [B,D], 'alpha': [B,H,M]}."""
    m = masks or {}
    xin = np.concatenate([x_t, att], axis=1)
    u = dropout(xin, m.get('in'), 1.0 - cfg.dropout_rnn_in)
    c2, h2, gc = rnn_cell(p, cfg, u, c, h)
    y = dropout(h2, m.get('out'), 1.0 - cfg.dropout_rnn_out)
    q = y @ p['W_q']
    alpha, ac = attention_scores(p, cfg, keys, q)
    alpha_d = dropout(alpha, m.get('alpha'), cfg.attn_keep_prob)
    ctx = context(cfg, alpha_d, values)
    att2 = ctx @ p['W_a'] if cfg.attn_context_layer else ctx
    cache = dict(u=u, h_prev=h, c_prev=c, gates=gc, c2=c2, h2=h2, y=y, q=q, alpha=alpha,
                 alpha_d=alpha_d, ctx=ctx, att_cache=ac)
    return y, c2, h2, att2, alpha_d, cache


# --------------------------------------------------------------------------- #
# training forward (rnn_decoder_training + post-process + losses)
# --------------------------------------------------------------------------- #
def make_dropout_masks(cfg, B, Tp, M, seed=0, dtype=np.float32):
    """Bernoulli keep masks for a training pass (TF's RNG stream cannot be reproduced;
    parity tests inject these same masks into the HIP path)."""
    rng = np.random.default_rng(seed)
    E, A, D, H = cfg.rnn_word_size, cfg.attn_size, cfg.rnn_size, cfg.attn_num_heads
    ki, ko, ka = 1 - cfg.dropout_rnn_in, 1 - cfg.dropout_rnn_out, cfg.attn_keep_prob
    return dict(
        init_in=(rng.random((B, E + A)) < ki).astype(dtype),
        inp=(rng.random((Tp, B, E + A)) < ki).astype(dtype),
        out=(rng.random((Tp, B, D)) < ko).astype(dtype),
        alpha=(rng.random((Tp, B, H, M)) < ka).astype(dtype))


def train_forward(p, cfg, fm, im_embed, captions, masks=None, rewards=None):
    """Teacher-forced decode + losses.

    Returns dict with logits [B,T,V], ids [B,T], attn_maps [B,H,T',M], xe (scalar loss:
    XE `sequence_loss`, or SCST reward-weighted when `rewards` is given), map_loss,
    and a cache for `train_backward`.
    """
    dt = fm.dtype
    inputs, targets, wmask, lens = process_inputs(captions, cfg.token_type)
    wmask = wmask.astype(dt)
    B, T = inputs.shape
    Tp = int(lens.max())
    D, H = cfg.rnn_size, cfg.attn_num_heads
    M = fm.shape[1]
    keys, values = memory_projections(p, cfg, fm)
    c, h, init_cache = rnn_init(p, cfg, im_embed, None if masks is None else masks['init_in'])
    att = np.zeros((B, cfg.attn_size), dt)
    emb_t = embed(p['emb'], inputs).transpose(1, 0, 2)          # [T,B,E]  time-major
    V = cfg.softmax_size
    logits = np.zeros((T, B, V), dt)
    alphas = np.zeros((Tp, B, H, M), dt)
    steps = []
    for t in range(Tp):
        fin = (t >= lens)                                       # finished BEFORE this step (A.6)
        sm = None if masks is None else {'in': masks['inp'][t], 'out': masks['out'][t],
                                         'alpha': masks['alpha'][t]}
        y, c2, h2, att2, alpha_d, cache = decoder_step(p, cfg, keys, values, emb_t[t], c, h, att, sm)
        lg = y @ p['W_o'] + p['b_o']
        keep = (~fin)[:, None].astype(dt)
        logits[t] = lg * keep                                   # impute_finished: zero outputs
        alphas[t] = alpha_d                                     # history is NOT imputed
        cache.update(fin=fin, att_prev=att)
        c = np.where(fin[:, None], c, c2)                       # ... and state frozen
        h = np.where(fin[:, None], h, h2)
        att = np.where(fin[:, None], att, att2)
        steps.append(cache)
    ids_tp = logits[:Tp].argmax(axis=2)
    if Tp < T:                                                  # ops_rnn.py:235-241
        logits[Tp:] = logits[Tp - 1]
    ids = np.concatenate([ids_tp, np.repeat(ids_tp[-1:], T - Tp, axis=0)], axis=0)
    logits_bt = logits.transpose(1, 0, 2)                       # [B,T,V]
    attn_maps = alphas.transpose(1, 2, 0, 3)                    # [B,H,T',M]

    # sequence_loss
    lsm = log_softmax(logits_bt, axis=-1)
    xent = -np.take_along_axis(lsm, targets[..., None], axis=2)[..., 0] * wmask
    if rewards is None:
        denom = wmask.sum() + dt.type(1e-12)
        xe = xent.sum() / denom
    else:
        denom = wmask.sum(axis=1) + dt.type(1e-12)
        xe = ((xent.sum(axis=1) / denom) * np.asarray(rewards, dt)).mean()
    # doubly stochastic attention loss: sum over HEADS (axis=1)  (model_base.py:356-365)
    flat = attn_maps.sum(axis=1)
    map_loss = ((1.0 - flat) ** 2).mean() * dt.type(cfg.rnn_map_loss_scale)
    out = dict(logits=logits_bt, ids=ids.T.astype(np.int32), attn_maps=attn_maps, xe=xe,
               map_loss=map_loss, log_softmax=lsm)
    out['cache'] = dict(keys=keys, values=values, fm=fm, im_embed=im_embed, init=init_cache,
                        steps=steps, emb_t=emb_t, inputs=inputs, targets=targets, wmask=wmask,
                        lens=lens, Tp=Tp, masks=masks, rewards=rewards, denom=denom, flat=flat)
    return out


def l2_loss(p, decay):
    """ops.l2_regulariser over every trainable var: decay * sum(w^2)/2."""
    return sum(float(decay) * 0.5 * float((np.asarray(v, np.float64) ** 2).sum()) for v in p.values())


def total_loss(p, cfg, out):
    return float(out['xe']) + float(out['map_loss']) + l2_loss(p, cfg.l2_decay)


# --------------------------------------------------------------------------- #
# analytic backward of train_forward  (TF autodiff restated)
# --------------------------------------------------------------------------- #
def _lstm_backward(p, cache_gates, c_prev, dc2, dh2, D):
    si, tj, sf, so, tc = cache_gates
    dso = dh2 * tc
    dtc = dh2 * so
    dc2 = dc2 + dtc * (1 - tc * tc)
    dsf = dc2 * c_prev
    dc_prev = dc2 * sf
    dsi = dc2 * tj
    dtj = dc2 * si
    dg = np.concatenate([dsi * si * (1 - si), dtj * (1 - tj * tj),
                         dsf * sf * (1 - sf), dso * so * (1 - so)], axis=1)
    return dg, dc_prev


def _ln_backward(dy, xhat, rstd, gamma):
    """y = xhat*gamma + beta -> (dz, dgamma, dbeta), rows independent."""
    dxh = dy * gamma
    dz = rstd * (dxh - dxh.mean(axis=-1, keepdims=True) - xhat * (dxh * xhat).mean(axis=-1, keepdims=True))
    return dz, (dy * xhat).sum(axis=0), dy.sum(axis=0)


def _cell_backward(p, cfg, grads, cache_gates, xh_rows, h_prev, c_prev, dc2, dh2):
    """Backward of one cell call: accumulates the cell's parameter gradients, returns
    (d [x ; h] of the cell input rows `xh_rows` = [u ; h_prev], d c_prev)."""
    D = h_prev.shape[-1]
    EA = xh_rows.shape[1] - D
    if cfg.rnn_name == 'LSTM':
        dg, dc_prev = _lstm_backward(p, cache_gates, c_prev, dc2, dh2, D)
        grads['K'] += xh_rows.T @ dg
        grads['b'] += dg.sum(axis=0)
        return dg @ p['K'].T, dc_prev
    if cfg.rnn_name == 'LN_LSTM':
        si, tj, sf, so, tc, xh, rs = cache_gates
        dso = dh2 * tc
        dcn = dc2 + dh2 * so * (1 - tc * tc)                       # d (normalised new c)
        dcraw, gg, gb = _ln_backward(dcn, xh[4], rs[4], p['cln_cg'])
        grads['cln_cg'] += gg; grads['cln_cb'] += gb
        dpre = [dcraw * tj * si * (1 - si), dcraw * si * (1 - tj * tj), dcraw * c_prev * sf * (1 - sf),
                dso * so * (1 - so)]
        dz = []
        for k, n in enumerate('ijfo'):
            z, gg, gb = _ln_backward(dpre[k], xh[k], rs[k], p['cln_%sg' % n])
            grads['cln_%sg' % n] += gg; grads['cln_%sb' % n] += gb
            dz.append(z)
        dg = np.concatenate(dz, axis=1)
        grads['K'] += xh_rows.T @ dg
        return dg @ p['K'].T, dcraw * sf
    r, u, cand = cache_gates                                       # GRU
    dpc = dh2 * (1 - u) * (1 - cand * cand)
    xh2 = np.concatenate([xh_rows[:, :EA], r * h_prev], axis=1)
    grads['K_c'] += xh2.T @ dpc
    grads['b_c'] += dpc.sum(axis=0)
    dxh2 = dpc @ p['K_c'].T
    drh = dxh2[:, EA:]
    dpg = np.concatenate([drh * h_prev * r * (1 - r), dh2 * (h_prev - cand) * u * (1 - u)], axis=1)
    grads['K'] += xh_rows.T @ dpg
    grads['b'] += dpg.sum(axis=0)
    dxh = dpg @ p['K'].T
    dxh[:, :EA] += dxh2[:, :EA]
    dxh[:, EA:] += drh * r + dh2 * u
    return dxh, dc2


def _attention_backward(p, cfg, keys, q, ac, alpha, dalpha, grads):
    """-> (dkeys, dq); accumulates d(v, ln_g, ln_b, tau) into grads."""
    B, M, D = keys.shape
    H = cfg.attn_num_heads
    d = D // H
    s = ac['s']
    if cfg.attn_probability_fn == 'softmax':
        ds = alpha * (dalpha - (alpha * dalpha).sum(axis=-1, keepdims=True))
    else:
        sg = ac['sg']
        S = sg.sum(axis=-1, keepdims=True)
        dsg = dalpha / S - (dalpha * sg).sum(axis=-1, keepdims=True) / (S * S)
        ds = dsg * sg * (1 - sg)
    if cfg.attn_alignment_method == 'add_LN':
        tau = p['tau']
        draw = ds / tau
        grads['tau'] += -(ds * ac['raw']).sum() / (tau * tau)
        draw_k = np.repeat(draw.transpose(0, 2, 1), d, axis=2)          # [B,M,D]
        th = ac['th']
        grads['v'] += (draw_k * th).sum(axis=(0, 1))
        dzhat = draw_k * p['v'] * (1 - th * th)
        xh = (ac['z'] - ac['mean']) * ac['rstd']
        grads['ln_g'] += (dzhat * xh).sum(axis=(0, 1))
        grads['ln_b'] += dzhat.sum(axis=(0, 1))
        dxh = dzhat * p['ln_g']
        dz = ac['rstd'] * (dxh - dxh.mean(axis=-1, keepdims=True)
                           - xh * (dxh * xh).mean(axis=-1, keepdims=True))
        return dz, dz.sum(axis=1)
    scale = keys.dtype.type(math.sqrt(D / H))
    draw_k = np.repeat((ds / scale).transpose(0, 2, 1), d, axis=2)
    return draw_k * q[:, None, :], (draw_k * keys).sum(axis=1)


def train_backward(p, cfg, out):
    """Gradients of xe + map_loss + L2 w.r.t. every parameter, plus d_fm, d_im_embed."""
    cc = out['cache']
    dt = cc['fm'].dtype
    keys, values, fm = cc['keys'], cc['values'], cc['fm']
    B, M, _ = keys.shape
    D, E, A, H = cfg.rnn_size, cfg.rnn_word_size, cfg.attn_size, cfg.attn_num_heads
    Tp, lens, masks = cc['Tp'], cc['lens'], cc['masks']
    T = cc['targets'].shape[1]
    grads = {k: np.zeros_like(v) for k, v in p.items()}

    # d logits  (softmax - onehot) * w / denom ; padded steps T'..T-1 alias step T'-1
    sm = np.exp(out['log_softmax'])
    oh = np.zeros_like(sm)
    np.put_along_axis(oh, cc['targets'][..., None], 1.0, axis=2)
    if cc['rewards'] is None:
        coef = cc['wmask'] / cc['denom']
    else:
        coef = cc['wmask'] / cc['denom'][:, None] * (np.asarray(cc['rewards'], dt) / B)[:, None]
    dlogits = ((sm - oh) * coef[..., None]).transpose(1, 0, 2)          # [T,B,V]
    dl = dlogits[:Tp].copy()
    if Tp < T:
        dl[Tp - 1] += dlogits[Tp:].sum(axis=0)

    # d alpha_d from the map loss (same for every head)
    n_el = B * Tp * M
    dflat = (2.0 * (cc['flat'] - 1.0) / n_el * cfg.rnn_map_loss_scale).astype(dt)   # [B,T',M]

    dkeys = np.zeros_like(keys)
    dvalues = np.zeros_like(values)
    dc = np.zeros((B, D), dt)
    dh = np.zeros((B, D), dt)
    datt = np.zeros((B, A), dt)
    demb_t = np.zeros((Tp, B, E), dt)
    Cv = values.shape[2]
    for t in reversed(range(Tp)):
        st = cc['steps'][t]
        fin = st['fin'][:, None]
        live = (~st['fin'])[:, None].astype(dt)
        # state select: s_next = fin ? s_prev : s_new
        dc2, dh2, datt2 = dc * live, dh * live, datt * live
        dc, dh, datt = dc * (1 - live), dh * (1 - live), datt * (1 - live)
        # logits (zeroed for finished rows)
        dlg = dl[t] * live
        grads['W_o'] += st['y'].T @ dlg
        grads['b_o'] += dlg.sum(axis=0)
        dy = dlg @ p['W_o'].T
        # attention / context
        if cfg.attn_context_layer:
            grads['W_a'] += st['ctx'].T @ datt2
            dctx = datt2 @ p['W_a'].T
        else:
            dctx = datt2
        vs = values.reshape(B, M, H, Cv // H)
        dctx_h = dctx.reshape(B, H, Cv // H)
        dalpha_d = np.einsum('bhd,bmhd->bhm', dctx_h, vs) + dflat[:, t][:, None, :]
        dvalues += np.einsum('bhm,bhd->bmhd', st['alpha_d'], dctx_h).reshape(B, M, Cv)
        if masks is not None:
            dalpha = (dalpha_d / dt.type(cfg.attn_keep_prob)) * masks['alpha'][t]
        else:
            dalpha = dalpha_d
        dk, dq = _attention_backward(p, cfg, keys, st['q'], st['att_cache'], st['alpha'], dalpha, grads)
        dkeys += dk
        grads['W_q'] += st['y'].T @ dq
        dy = dy + dq @ p['W_q'].T
        if masks is not None:
            dh2 = dh2 + (dy / dt.type(1 - cfg.dropout_rnn_out)) * masks['out'][t]
        else:
            dh2 = dh2 + dy
        dxh, dc_prev = _cell_backward(p, cfg, grads, st['gates'], np.concatenate([st['u'], st['h_prev']], axis=1),
                                      st['h_prev'], st['c_prev'], dc2, dh2)
        du, dh_prev = dxh[:, :E + A], dxh[:, E + A:]
        if masks is not None:
            dxin = (du / dt.type(1 - cfg.dropout_rnn_in)) * masks['inp'][t]
        else:
            dxin = du
        demb_t[t] = dxin[:, :E]
        datt = datt + dxin[:, E:]
        dc = dc + dc_prev
        dh = dh + dh_prev
    # embeddings (ids < 0 have no row)
    ids = cc['inputs'].T[:Tp]                                    # [T',B]
    valid = ids >= 0
    np.add.at(grads['emb'], ids[valid], demb_t[valid])
    # rnn init
    im = cc['im_embed']
    if cfg.rnn_init_method == 'project_hidden':
        grads['W_init'] += im.T @ dh
        dim = dh @ p['W_init'].T
    else:
        ic = cc['init']
        zeros = np.zeros((B, D), dt)
        dxh, _ = _cell_backward(p, cfg, grads, ic['gates'], np.concatenate([ic['u'], zeros], axis=1), zeros, zeros, dc, dh)
        du = dxh[:, :E + A]
        if masks is not None:
            dx = (du / dt.type(1 - cfg.dropout_rnn_in)) * masks['init_in']
        else:
            dx = du
        grads['W_init'] += im.T @ dx
        dim = dx @ p['W_init'].T
    # memory projections
    C = fm.shape[2]
    fm2 = fm.reshape(B * M, C)
    if cfg.cnn_fm_projection == 'tied':
        dk_tot = dkeys + dvalues
        grads['W_m'] += fm2.T @ dk_tot.reshape(B * M, D)
        dfm = dk_tot @ p['W_m'].T
    elif cfg.cnn_fm_projection == 'independent':
        grads['W_m'] += fm2.T @ dkeys.reshape(B * M, D)
        grads['W_v'] += fm2.T @ dvalues.reshape(B * M, D)
        dfm = dkeys @ p['W_m'].T + dvalues @ p['W_v'].T
    else:
        grads['W_m'] += fm2.T @ dkeys.reshape(B * M, D)
        dfm = dkeys @ p['W_m'].T + dvalues
    # L2 on every trainable variable (model_base.py:408-417)
    if cfg.l2_decay > 0:
        for k in grads:
            grads[k] = grads[k] + dt.type(cfg.l2_decay) * p[k]
    return grads, dfm, dim


# --------------------------------------------------------------------------- #
# optimiser
# --------------------------------------------------------------------------- #
def cosine_lr(step, max_step, lr_start, lr_end):
    """model_base.py:809-820 (fp32 arithmetic after the int/int true-division)."""
    s = np.float32(step / max_step)
    s = np.float32(1.0) + np.cos(np.minimum(np.float32(1.0), s) * np.float32(math.pi), dtype=np.float32)
    return np.float32(np.float32(lr_start - lr_end) * s / np.float32(2) + np.float32(lr_end))


def adam_tf_update(w, g, m, v, t, lr, beta1=0.9, beta2=0.999, eps=1e-2):
    """tf.train.AdamOptimizer.ApplyAdam [TF-1.9]; t = 1 for the first update.
    lr_t = lr*sqrt(1-b2^t)/(1-b1^t); w -= lr_t * m / (sqrt(v) + eps)."""
    f = np.float32
    lr_t = f(lr) * f(math.sqrt(1.0 - beta2 ** t)) / f(1.0 - beta1 ** t)
    m[...] = m + (g - m) * f(1 - beta1)
    v[...] = v + (g * g - v) * f(1 - beta2)
    w[...] = w - (m * lr_t) / (np.sqrt(v) + f(eps))
    return w, m, v


def clip_by_norm(g, clip_norm):
    """slim.learning.clip_gradient_norms -> tf.clip_by_norm on ONE variable's gradient [TF-1.9 slim / clip_ops]
    (model_base.py:394-401, create_train_op(clip_gradient_norm=c)): g * clip / max(||g||_2, clip)."""
    g = np.asarray(g, np.float32)
    norm = np.sqrt(np.sum(g.astype(np.float64) ** 2))
    return (g * np.float32(clip_norm / max(norm, clip_norm))).astype(np.float32)


def momentum_tf_update(w, g, accum, lr, momentum=0.9):
    """tf.train.MomentumOptimizer.ApplyMomentum, use_nesterov=False [TF-1.9] (model_base.py:867-880):
    accum = momentum*accum + g; w -= lr*accum."""
    f = np.float32
    accum[...] = accum * f(momentum) + g
    w[...] = w - f(lr) * accum
    return w, accum


# --------------------------------------------------------------------------- #
# legacy encoder head (model_base.py:80-91; common/ops.py:200-275)
# --------------------------------------------------------------------------- #
def legacy_head_forward(p, net, eps=1e-12):
    """im_embed = tanh(layer_norm(net)) . W  (LayerNorm over the last axis, biased variance, eps 1e-12 [TF-1.9
    tf.contrib.layers.layer_norm]; ops.linear without bias).  -> (im_embed, cache)."""
    x = np.asarray(net, np.float64)
    mean = x.mean(axis=1, keepdims=True)
    var = ((x - mean) ** 2).mean(axis=1, keepdims=True)
    xhat = (x - mean) / np.sqrt(var + eps)
    z = np.tanh(xhat * p['ln_gamma'].astype(np.float64) + p['ln_beta'].astype(np.float64))
    return (z @ p['W'].astype(np.float64)).astype(np.float32), dict(z=z, xhat=xhat)


def legacy_head_backward(p, cache, d_im):
    """-> {ln_gamma, ln_beta, W} gradients (the gradient w.r.t. `net` is not needed: the CNN is frozen with --legacy)."""
    d = np.asarray(d_im, np.float64)
    z, xhat = cache['z'], cache['xhat']
    dz = d @ p['W'].astype(np.float64).T
    t = dz * (1.0 - z * z)
    return dict(W=(z.T @ d).astype(np.float32), ln_gamma=(t * xhat).sum(axis=0).astype(np.float32),
                ln_beta=t.sum(axis=0).astype(np.float32))
