"""Second, independent CPU formulation of the hot path on torch (oneDNN convolutions, autograd): TEST INFRASTRUCTURE.

Two uses, both outside the product: (i) tests/test_oracle_decoder.py and tests/test_oracle_cnn.py cross-check the numpy
oracle against it (the TF-1.9 graph arithmetic cannot be run here, so the oracle is pinned for
internal consistency by a formulation that shares none of its arithmetic); (ii) bench.py's `cpu_baseline` times it as
the "framework CPU path" stand-in BASELINE.md section 3.2(b) names (the literal TF-1 binary is absent).
Reference lines restated: common/nets/inception_v3.py:100-415 (through oracle.cnn_ref's layer walk),
common/nets/inception_utils.py:32-82, common/ops_rnn.py:660-755, src/model_base.py:325-417.
"""
import math

import numpy as np
import torch
import torch.nn.functional as F

from . import cnn_ref


def _same_pad(size, k, stride):
    tot = max((-(-size // stride) - 1) * stride + k - size, 0)
    return tot // 2, tot - tot // 2              # TF SAME: the extra pixel goes after


class _TorchNet(cnn_ref._Net):
    """oracle.cnn_ref's network walk (layer order, scopes, variable names) with torch NCHW tensors behind it."""

    def __init__(self, params):
        super().__init__(params, None, act_dtype='f32', run=True)
        self.tw = {}           # weights name -> (w OIHW (requires grad), scale, mean, beta (requires grad))

    def conv(self, x, cout, k, stride=1, padding='VALID', scope=None):
        kh, kw = (k, k) if isinstance(k, int) else k
        self.scope.append(scope)
        wn, bn = self.name('weights'), self.name('BatchNorm/beta')
        mn, vn = self.name('BatchNorm/moving_mean'), self.name('BatchNorm/moving_variance')
        self.scope.pop()
        if wn not in self.tw:
            w = torch.from_numpy(np.ascontiguousarray(self.params[wn].transpose(3, 2, 0, 1)))       # HWIO -> OIHW
            scale = (1.0 / np.sqrt(self.params[vn] + np.float32(cnn_ref.BN_EPS))).astype(np.float32)
            self.tw[wn] = (w.requires_grad_(True), torch.from_numpy(scale), torch.from_numpy(self.params[mn]),
                           torch.from_numpy(self.params[bn].copy()).requires_grad_(True))
        w, scale, mean, beta = self.tw[wn]
        if padding == 'SAME':
            (pt, pb), (pl, pr) = _same_pad(x.shape[2], kh, stride), _same_pad(x.shape[3], kw, stride)
            x = F.pad(x, (pl, pr, pt, pb))
        y = F.conv2d(x, w, stride=stride)
        return torch.relu((y - mean[None, :, None, None]) * scale[None, :, None, None] + beta[None, :, None, None])

    def max_pool(self, x, k=3, stride=2, padding='VALID'):
        if padding == 'SAME':
            (pt, pb), (pl, pr) = _same_pad(x.shape[2], k, stride), _same_pad(x.shape[3], k, stride)
            x = F.pad(x, (pl, pr, pt, pb), value=float('-inf'))
        return F.max_pool2d(x, k, stride)

    def avg_pool(self, x, k=3, stride=1, padding='SAME'):
        kh, kw = (k, k) if isinstance(k, int) else k
        if padding == 'SAME':          # stride 1, odd kernels: symmetric padding, divisor = taps inside the image
            return F.avg_pool2d(x, (kh, kw), stride, (kh // 2, kw // 2), count_include_pad=False)
        return F.avg_pool2d(x, (kh, kw), stride)

    def concat(self, xs):
        return torch.cat(xs, dim=1)


def torch_encoder(params, images, fm_name='Mixed_7c'):
    """ModelBase._encoder (non-legacy, model_base.py:56-104) on torch-CPU.  images [B,H,W,3] numpy ->
    (im_embed [B,C_g], fm [B,M,C], net): torch tensors with the CNN's autograd graph behind them."""
    net = _TorchNet(params)
    x = torch.from_numpy(np.ascontiguousarray(np.asarray(images, np.float32).transpose(0, 3, 1, 2)))
    ep = {}
    y = cnn_ref._inception_v3_base(net, x, ep)
    pooled = net.avg_pool(y, (min(y.shape[2], 8), min(y.shape[3], 8)), 1, 'VALID')     # inception_v3.py:520-532
    fm = ep[fm_name]
    B, C, H, W = fm.shape
    return pooled.reshape(B, -1), fm.permute(0, 2, 3, 1).reshape(B, H * W, C), net


def torch_forward(p, cfg, fm, im, caps, masks, rewards=None, dtype=None):
    """Independent formulation of the decoder step: torch ops, F.layer_norm, F.cross_entropy, autograd
    (common/ops_rnn.py:660-755, src/model_base.py:325-417 restated a second time).  `fm` / `im` may be torch tensors
    with a graph behind them (torch CNN): gradients then flow into it."""
    dt = dtype or torch.float64
    tp = {k: torch.tensor(v, dtype=dt, requires_grad=True) for k, v in p.items()}
    fm_t = fm if torch.is_tensor(fm) else torch.tensor(fm, dtype=dt, requires_grad=True)
    im_t = im if torch.is_tensor(im) else torch.tensor(im, dtype=dt, requires_grad=True)
    D, E, H = cfg.rnn_size, cfg.rnn_word_size, cfg.attn_num_heads
    caps_t = torch.tensor(caps)
    wmask = torch.sign((caps_t[:, 1:] + 1).to(dt))
    lens = wmask.sum(1).long()
    inputs = caps_t[:, :-1]
    if cfg.token_type == 'word':
        inputs = inputs.clamp(min=0)
    targets = caps_t.clamp(min=0)[:, 1:]
    B, T = inputs.shape
    Tp = int(lens.max())
    M = fm.shape[1]
    mk = (lambda k: None) if masks is None else (lambda k: torch.tensor(masks[k], dtype=dt))

    def drop(x, m, keep):
        return x if m is None else x / keep * m

    keys = fm_t @ tp['W_m']
    if cfg.cnn_fm_projection == 'tied':
        values = keys
    elif cfg.cnn_fm_projection == 'independent':
        values = fm_t @ tp['W_v']
    else:
        values = fm_t
    Cv = values.shape[-1]

    def lstm(xin, c, h):
        if cfg.rnn_name == 'GRU':
            r, u = torch.sigmoid(torch.cat([xin, h], 1) @ tp['K'] + tp['b']).chunk(2, dim=1)
            cand = torch.tanh(torch.cat([xin, r * h], 1) @ tp['K_c'] + tp['b_c'])
            return c, u * h + (1 - u) * cand
        if cfg.rnn_name == 'LN_LSTM':
            ln = lambda z, n: F.layer_norm(z, (D,), tp['cln_%sg' % n], tp['cln_%sb' % n], eps=1e-12)
            i, j, f, o = (torch.cat([xin, h], 1) @ tp['K']).chunk(4, dim=1)
            c2 = ln(c * torch.sigmoid(ln(f, 'f') + 1.0) + torch.sigmoid(ln(i, 'i')) * torch.tanh(ln(j, 'j')), 'c')
            return c2, torch.tanh(c2) * torch.sigmoid(ln(o, 'o'))
        g = torch.cat([xin, h], 1) @ tp['K'] + tp['b']
        i, j, f, o = g.chunk(4, dim=1)
        c2 = c * torch.sigmoid(f + 1.0) + torch.sigmoid(i) * torch.tanh(j)
        return c2, torch.tanh(c2) * torch.sigmoid(o)

    z0 = torch.zeros(B, D, dtype=dt)
    if cfg.rnn_init_method == 'project_hidden':
        h, c = im_t @ tp['W_init'], z0
    else:
        c, h = lstm(drop(im_t @ tp['W_init'], mk('init_in'), 1 - cfg.dropout_rnn_in), z0, z0)
    att = torch.zeros(B, cfg.attn_size, dtype=dt)
    onehot = F.one_hot(inputs.clamp(min=0), cfg.softmax_size).to(dt) * (inputs >= 0)[..., None]
    emb = onehot @ tp['emb']
    logits, alphas = [], []
    mi, mo, ma = mk('inp'), mk('out'), mk('alpha')
    for t in range(Tp):
        fin = (t >= lens)[:, None]
        u = drop(torch.cat([emb[:, t], att], 1), None if mi is None else mi[t], 1 - cfg.dropout_rnn_in)
        c2, h2 = lstm(u, c, h)
        y = drop(h2, None if mo is None else mo[t], 1 - cfg.dropout_rnn_out)
        q = y @ tp['W_q']
        if cfg.attn_alignment_method == 'add_LN':
            zz = F.layer_norm(keys + q[:, None, :], (D,), tp['ln_g'], tp['ln_b'], eps=1e-12)
            sc = (torch.tanh(zz) * tp['v']).view(B, M, H, D // H).sum(-1).permute(0, 2, 1) / tp['tau']
        else:
            sc = (keys * q[:, None, :]).view(B, M, H, D // H).sum(-1).permute(0, 2, 1) / math.sqrt(D / H)
        if cfg.attn_probability_fn == 'softmax':
            al = torch.softmax(sc, -1)
        else:
            sg = torch.sigmoid(sc)
            al = sg / sg.sum(-1, keepdim=True)
        al = drop(al, None if ma is None else ma[t], cfg.attn_keep_prob)
        ctx = torch.matmul(al[:, :, None, :], values.view(B, M, H, Cv // H).permute(0, 2, 1, 3))
        ctx = ctx.squeeze(2).reshape(B, Cv)
        att2 = ctx @ tp['W_a'] if cfg.attn_context_layer else ctx
        lg = y @ tp['W_o'] + tp['b_o']
        logits.append(torch.where(fin, torch.zeros_like(lg), lg))
        alphas.append(al)
        c, h, att = torch.where(fin, c, c2), torch.where(fin, h, h2), torch.where(fin, att, att2)
    logits = torch.stack(logits + [logits[-1]] * (T - Tp), 1)           # [B,T,V]
    amap = torch.stack(alphas, 2)                                        # [B,H,T',M]
    ce = F.cross_entropy(logits.reshape(B * T, -1), targets.reshape(-1), reduction='none').view(B, T) * wmask
    if rewards is None:
        xe = ce.sum() / (wmask.sum() + 1e-12)
    else:
        xe = ((ce.sum(1) / (wmask.sum(1) + 1e-12)) * torch.tensor(rewards, dtype=dt)).mean()
    map_loss = ((1.0 - amap.sum(1)) ** 2).mean() * cfg.rnn_map_loss_scale
    l2 = sum(cfg.l2_decay * 0.5 * (v ** 2).sum() for v in tp.values())
    return xe, map_loss, l2, logits, amap, tp, fm_t, im_t


def torch_train_step(cnn_params, dec_params, cfg, images, caps, lr=1e-3, eps=1e-2, state=None):
    """One decoder-mode XE step on torch-CPU in fp32: InceptionV3 forward (frozen, no graph), decoder forward + autograd
    backward, TF-Adam update of the decoder parameters (in place).  -> (loss, state)."""
    with torch.no_grad():
        im, fm, _ = torch_encoder(cnn_params, images)
    xe, map_loss, l2, _, _, tp, _, _ = torch_forward(dec_params, cfg, fm.detach(), im.detach(), caps, None, dtype=torch.float32)
    (xe + map_loss + l2).backward()
    state = state or dict(t=0, m={k: np.zeros_like(v) for k, v in dec_params.items()},
                          v={k: np.zeros_like(v) for k, v in dec_params.items()})
    state['t'] += 1
    t, b1, b2 = state['t'], 0.9, 0.999
    lr_t = lr * math.sqrt(1 - b2 ** t) / (1 - b1 ** t)
    for k in dec_params:
        g = tp[k].grad.numpy()
        state['m'][k] += (g - state['m'][k]) * np.float32(1 - b1)
        state['v'][k] += (g * g - state['v'][k]) * np.float32(1 - b2)
        dec_params[k] -= np.float32(lr_t) * state['m'][k] / (np.sqrt(state['v'][k]) + np.float32(eps))
    return float(xe.detach()), state
