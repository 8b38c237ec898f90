"""CPU restatement (numpy, fp32) of the reference's InceptionV3 encoder.

TEST INFRASTRUCTURE -- see oracle/__init__.py.  PARITY UNPINNED: the arithmetic
lives in tensorflow==1.9.0 (README.md:48), which is absent here; this follows
the reference's call sites and TF-1.9 semantics:

  * network structure ......... common/nets/inception_v3.py:100-415 (base),
                                 :419-544 (head, ``num_classes=None`` returns at :531-532)
  * conv + BN(inference) + ReLU  common/nets/inception_utils.py:32-82
                                 (no bias, no gamma, eps 1e-3, is_training=False:
                                 src/model_base.py:72-77)
  * feature map / im_embed ..... src/model_base.py:93-104

Pins available from the reference's own tests (checked in tests/test_oracle_cnn.py):
end-point shapes at 299 (inception_v3_test.py:93-123) and the parameter count
21 802 784 (inception_v3_test.py:125-133).

Layout: activations NHWC, weights HWIO (TF layout), all float32.  ``act_dtype='bf16'``
emulates the product's bf16 storage: every conv input and every weight is rounded
to bf16 (RNE) before an fp32-accumulated contraction, and every layer output is
rounded to bf16 when stored.
"""
from __future__ import annotations

import math
from collections import OrderedDict

import numpy as np

BN_EPS = 1e-3  # inception_utils.py:36


# --------------------------------------------------------------------------- #
# dtype helpers
# --------------------------------------------------------------------------- #
def bf16_round(x: np.ndarray) -> np.ndarray:
    """Round fp32 -> bf16 (round-to-nearest-even), returned as fp32."""
    x = np.ascontiguousarray(x, dtype=np.float32)
    u = x.view(np.uint32)
    # NaN-safe enough for the oracle (inputs are finite)
    r = ((u >> 16) & 1) + np.uint32(0x7FFF)
    y = ((u + r) & np.uint32(0xFFFF0000)).astype(np.uint32)
    return y.view(np.float32)


def _q(x, act_dtype):
    return bf16_round(x) if act_dtype == 'bf16' else x


# --------------------------------------------------------------------------- #
# TF padding arithmetic
# --------------------------------------------------------------------------- #
def same_pad(in_size: int, k: int, s: int):
    out = -(-in_size // s)
    total = max((out - 1) * s + k - in_size, 0)
    return out, total // 2, total - total // 2


def out_size(in_size, k, s, padding):
    if padding == 'SAME':
        return same_pad(in_size, k, s)
    return (in_size - k) // s + 1, 0, 0


# --------------------------------------------------------------------------- #
# primitive ops
# --------------------------------------------------------------------------- #
def conv2d(x, w, stride=1, padding='VALID'):
    """NHWC x HWIO -> NHWC, fp32 accumulate (im2col + GEMM)."""
    B, H, W, C = x.shape
    kh, kw, ci, co = w.shape
    assert ci == C
    Ho, pt, pb = out_size(H, kh, stride, padding)
    Wo, pl, pr = out_size(W, kw, stride, padding)
    if pt or pb or pl or pr:
        x = np.pad(x, ((0, 0), (pt, pb), (pl, pr), (0, 0)))
    if kh == 1 and kw == 1 and stride == 1:
        return (x.reshape(-1, C) @ w.reshape(C, co)).reshape(B, Ho, Wo, co)
    cols = np.empty((B, Ho, Wo, kh, kw, C), np.float32)
    for i in range(kh):
        for j in range(kw):
            cols[:, :, :, i, j, :] = x[:, i:i + (Ho - 1) * stride + 1:stride,
                                       j:j + (Wo - 1) * stride + 1:stride, :]
    y = cols.reshape(B * Ho * Wo, kh * kw * C) @ w.reshape(kh * kw * C, co)
    return y.reshape(B, Ho, Wo, co)


def batch_norm_inference(x, beta, mean, var):
    """FusedBatchNorm, is_training=False, no gamma (inception_utils.py:56-66)."""
    return (x - mean) * (np.float32(1.0) / np.sqrt(var + np.float32(BN_EPS))) + beta


def max_pool(x, k=3, stride=2, padding='VALID'):
    B, H, W, C = x.shape
    Ho, pt, pb = out_size(H, k, stride, padding)
    Wo, pl, pr = out_size(W, k, stride, padding)
    if pt or pb or pl or pr:
        x = np.pad(x, ((0, 0), (pt, pb), (pl, pr), (0, 0)), constant_values=-np.inf)
    y = np.full((B, Ho, Wo, C), -np.inf, np.float32)
    for i in range(k):
        for j in range(k):
            y = np.maximum(y, x[:, i:i + (Ho - 1) * stride + 1:stride,
                                j:j + (Wo - 1) * stride + 1:stride, :])
    return y


def avg_pool(x, k=3, stride=1, padding='SAME'):
    """TF AvgPool: SAME padding divides by the number of valid taps."""
    B, H, W, C = x.shape
    kh, kw = (k, k) if isinstance(k, int) else k
    Ho, pt, pb = out_size(H, kh, stride, padding)
    Wo, pl, pr = out_size(W, kw, stride, padding)
    xp = np.pad(x, ((0, 0), (pt, pb), (pl, pr), (0, 0)))
    ones = np.pad(np.ones((1, H, W, 1), np.float32), ((0, 0), (pt, pb), (pl, pr), (0, 0)))
    acc = np.zeros((B, Ho, Wo, C), np.float32)
    cnt = np.zeros((1, Ho, Wo, 1), np.float32)
    for i in range(kh):
        for j in range(kw):
            acc += xp[:, i:i + (Ho - 1) * stride + 1:stride, j:j + (Wo - 1) * stride + 1:stride, :]
            cnt += ones[:, i:i + (Ho - 1) * stride + 1:stride, j:j + (Wo - 1) * stride + 1:stride, :]
    return acc / cnt


# --------------------------------------------------------------------------- #
# backward primitives (cnn_finetune: the CNN variables join the trainable set when
# freeze_scopes == '' -- src/train.py:241-249, src/model_base.py:834-849; the graph is still
# built with is_training=False, model_base.py:76, so BN contributes d beta only)
# --------------------------------------------------------------------------- #
def conv2d_bwd(x, w, dy, stride=1, padding='VALID'):
    """-> (dx, dw) of y = conv2d(x, w)."""
    B, H, W, C = x.shape
    kh, kw, ci, co = w.shape
    Ho, pt, pb = out_size(H, kh, stride, padding)
    Wo, pl, pr = out_size(W, kw, stride, padding)
    xp = np.pad(x, ((0, 0), (pt, pb), (pl, pr), (0, 0))) if (pt or pb or pl or pr) else x
    cols = np.empty((B, Ho, Wo, kh, kw, C), np.float32)
    for i in range(kh):
        for j in range(kw):
            cols[:, :, :, i, j, :] = xp[:, i:i + (Ho - 1) * stride + 1:stride,
                                        j:j + (Wo - 1) * stride + 1:stride, :]
    dy2 = np.ascontiguousarray(dy, np.float32).reshape(B * Ho * Wo, co)
    dw = (cols.reshape(B * Ho * Wo, kh * kw * C).T @ dy2).reshape(kh, kw, ci, co)
    dcols = (dy2 @ w.reshape(kh * kw * C, co).T).reshape(B, Ho, Wo, kh, kw, C)
    dxp = np.zeros_like(xp)
    for i in range(kh):
        for j in range(kw):
            dxp[:, i:i + (Ho - 1) * stride + 1:stride, j:j + (Wo - 1) * stride + 1:stride, :] += dcols[:, :, :, i, j, :]
    return dxp[:, pt:pt + H, pl:pl + W, :], dw


def max_pool_bwd(x, dy, k=3, stride=2, padding='VALID'):
    """MaxPoolGrad: the gradient of a window goes to its FIRST maximum in window scan order
    (kh-major) -- TF's CPU kernel (argmax from the forward pass).  Ties other than between
    ReLU zeros (whose gradient the producing conv masks anyway) do not occur in practice."""
    B, H, W, C = x.shape
    Ho, pt, pb = out_size(H, k, stride, padding)
    Wo, pl, pr = out_size(W, k, stride, padding)
    xp = np.pad(x, ((0, 0), (pt, pb), (pl, pr), (0, 0)), constant_values=-np.inf) if (pt or pb or pl or pr) else x
    win = np.stack([xp[:, i:i + (Ho - 1) * stride + 1:stride, j:j + (Wo - 1) * stride + 1:stride, :]
                    for i in range(k) for j in range(k)], axis=0)          # [k*k, B, Ho, Wo, C]
    arg = np.argmax(win, axis=0)                                          # first maximum
    dxp = np.zeros_like(xp)
    for t in range(k * k):
        i, j = divmod(t, k)
        dxp[:, i:i + (Ho - 1) * stride + 1:stride, j:j + (Wo - 1) * stride + 1:stride, :] += np.where(arg == t, dy, 0)
    return dxp[:, pt:pt + H, pl:pl + W, :]


def avg_pool_bwd(x_shape, dy, k=3, stride=1, padding='SAME'):
    B, H, W, C = x_shape
    kh, kw = (k, k) if isinstance(k, int) else k
    Ho, pt, pb = out_size(H, kh, stride, padding)
    Wo, pl, pr = out_size(W, kw, stride, padding)
    ones = np.pad(np.ones((1, H, W, 1), np.float32), ((0, 0), (pt, pb), (pl, pr), (0, 0)))
    cnt = np.zeros((1, Ho, Wo, 1), np.float32)
    for i in range(kh):
        for j in range(kw):
            cnt += ones[:, i:i + (Ho - 1) * stride + 1:stride, j:j + (Wo - 1) * stride + 1:stride, :]
    g = np.asarray(dy, np.float32) / cnt
    dxp = np.zeros((B, H + pt + pb, W + pl + pr, C), np.float32)
    for i in range(kh):
        for j in range(kw):
            dxp[:, i:i + (Ho - 1) * stride + 1:stride, j:j + (Wo - 1) * stride + 1:stride, :] += g
    return dxp[:, pt:pt + H, pl:pl + W, :]


# --------------------------------------------------------------------------- #
# network builder: used both to create parameters and to run the forward pass
# --------------------------------------------------------------------------- #
class _Net:
    def __init__(self, params=None, rng=None, act_dtype='f32', run=True, tape=False, override=None):
        # override(weights name, y) -> y': hook applied to every conv output.  The backward parity tests substitute
        # the DEVICE's activation there (after comparing it with y), so the oracle's reverse pass sees the same ReLU
        # masks and pool arg-maxima as the kernels and the gradient comparison measures arithmetic, not mask flips
        self.override = override
        self.tape = [] if tape else None     # reverse-mode records: (kind, inputs, output, ctx)
        self.params = params if params is not None else OrderedDict()
        self.create = params is None
        self.rng = rng
        self.act_dtype = act_dtype
        self.run = run
        self.scope = []
        self.macs = 0          # multiply-accumulates per image (conv only)
        self.conv_log = []     # (name, in_shape, out_shape, k, stride, padding)

    def name(self, leaf):
        return '/'.join(self.scope + [leaf])

    def conv(self, x, cout, k, stride=1, padding='VALID', scope=None):
        kh, kw = (k, k) if isinstance(k, int) else k
        self.scope.append(scope)
        cin = x.shape[-1]
        wn, bn = self.name('weights'), self.name('BatchNorm/beta')
        mn, vn = self.name('BatchNorm/moving_mean'), self.name('BatchNorm/moving_variance')
        if self.create:
            # slim.variance_scaling_initializer(): factor 2.0, FAN_IN, truncated normal
            fan_in = kh * kw * cin
            std = math.sqrt(1.3 * 2.0 / fan_in)
            w = self.rng.standard_normal((kh, kw, cin, cout)).astype(np.float32)
            w = np.clip(w, -2, 2) * np.float32(std)
            self.params[wn] = w
            self.params[bn] = np.zeros(cout, np.float32)
            self.params[mn] = np.zeros(cout, np.float32)
            self.params[vn] = np.ones(cout, np.float32)
        w = self.params[wn]
        B, H, W, _ = x.shape
        Ho = out_size(H, kh, stride, padding)[0]
        Wo = out_size(W, kw, stride, padding)[0]
        self.macs += Ho * Wo * kh * kw * cin * cout
        self.conv_log.append((wn, (H, W, cin), (Ho, Wo, cout), (kh, kw), stride, padding))
        if self.run:
            y = conv2d(_q(x, self.act_dtype), _q(w, self.act_dtype), stride, padding)
            y = batch_norm_inference(y, self.params[bn], self.params[mn], self.params[vn])
            y = _q(np.maximum(y, 0), self.act_dtype)
            if self.override is not None:
                y = self.override(wn, y)
            if self.tape is not None:
                self.tape.append(('conv', [x], y, (wn, bn, vn, stride, padding)))
        else:
            y = np.zeros((B, Ho, Wo, cout), np.float32)
        self.scope.pop()
        return y

    def max_pool(self, x, k=3, stride=2, padding='VALID'):
        if not self.run:
            B, H, W, C = x.shape
            return np.zeros((B, out_size(H, k, stride, padding)[0],
                             out_size(W, k, stride, padding)[0], C), np.float32)
        y = max_pool(x, k, stride, padding)
        if self.tape is not None:
            self.tape.append(('max', [x], y, (k, stride, padding)))
        return y

    def concat(self, xs):
        y = np.concatenate(xs, axis=3)
        if self.tape is not None:
            self.tape.append(('cat', list(xs), y, None))
        return y

    def backward(self, seeds):
        """seeds: list of (array produced by this net, gradient).  -> {variable name: gradient}
        for every conv weight and BN beta (fp32 arithmetic on the taped activations)."""
        g = {}
        keep = []

        def add(arr, val):
            k = id(arr)
            if k in g:
                g[k] = g[k] + val
            else:
                g[k] = np.asarray(val, np.float32)
                keep.append(arr)
        for arr, val in seeds:
            add(arr, val)
        out = {}
        for kind, ins, y, ctx in reversed(self.tape):
            dy = g.pop(id(y), None)
            if dy is None:
                continue
            if kind == 'conv':
                wn, bn, vn, stride, padding = ctx
                dz = np.where(y > 0, dy, 0).astype(np.float32)
                out[bn] = out.get(bn, 0) + dz.sum(axis=(0, 1, 2))
                dz = dz * (np.float32(1.0) / np.sqrt(self.params[vn] + np.float32(BN_EPS)))
                dx, dw = conv2d_bwd(_q(ins[0], self.act_dtype), _q(self.params[wn], self.act_dtype), dz, stride, padding)
                out[wn] = out.get(wn, 0) + dw
                add(ins[0], dx)
            elif kind == 'max':
                add(ins[0], max_pool_bwd(ins[0], dy, *ctx))
            elif kind == 'avg':
                add(ins[0], avg_pool_bwd(ins[0].shape, dy, *ctx))
            elif kind == 'cat':
                o = 0
                for x in ins:
                    add(x, dy[..., o:o + x.shape[3]])
                    o += x.shape[3]
        return out

    def avg_pool(self, x, k=3, stride=1, padding='SAME'):
        if not self.run:
            B, H, W, C = x.shape
            kh, kw = (k, k) if isinstance(k, int) else k
            return np.zeros((B, out_size(H, kh, stride, padding)[0],
                             out_size(W, kw, stride, padding)[0], C), np.float32)
        y = _q(avg_pool(x, k, stride, padding), self.act_dtype)
        if self.tape is not None:
            self.tape.append(('avg', [x], y, (k, stride, padding)))
        return y


def _inception_v3_base(net: _Net, x, end_points):
    """common/nets/inception_v3.py:100-415."""
    cat = net.concat
    net.scope.append('InceptionV3')
    # stem, stride 1 / VALID defaults (inception_v3.py:100-137)
    x = net.conv(x, 32, 3, 2, 'VALID', 'Conv2d_1a_3x3'); end_points['Conv2d_1a_3x3'] = x
    x = net.conv(x, 32, 3, 1, 'VALID', 'Conv2d_2a_3x3'); end_points['Conv2d_2a_3x3'] = x
    x = net.conv(x, 64, 3, 1, 'SAME', 'Conv2d_2b_3x3'); end_points['Conv2d_2b_3x3'] = x
    x = net.max_pool(x, 3, 2, 'VALID'); end_points['MaxPool_3a_3x3'] = x
    x = net.conv(x, 80, 1, 1, 'VALID', 'Conv2d_3b_1x1'); end_points['Conv2d_3b_1x1'] = x
    x = net.conv(x, 192, 3, 1, 'VALID', 'Conv2d_4a_3x3'); end_points['Conv2d_4a_3x3'] = x
    x = net.max_pool(x, 3, 2, 'VALID'); end_points['MaxPool_5a_3x3'] = x

    def c(inp, cout, k, scope, stride=1, padding='SAME'):
        return net.conv(inp, cout, k, stride, padding, scope)

    # Mixed_5b/5c/5d (inception_v3.py:141-210).  Note the reference's odd scope
    # names in Mixed_5c Branch_1 ('Conv2d_0b_1x1', 'Conv_1_0c_5x5').
    for blk, pool_c, b1 in (('Mixed_5b', 32, ('Conv2d_0a_1x1', 'Conv2d_0b_5x5')),
                            ('Mixed_5c', 64, ('Conv2d_0b_1x1', 'Conv_1_0c_5x5')),
                            ('Mixed_5d', 64, ('Conv2d_0a_1x1', 'Conv2d_0b_5x5'))):
        net.scope.append(blk)
        net.scope.append('Branch_0'); b0 = c(x, 64, 1, 'Conv2d_0a_1x1'); net.scope.pop()
        net.scope.append('Branch_1')
        br1 = c(x, 48, 1, b1[0]); br1 = c(br1, 64, 5, b1[1]); net.scope.pop()
        net.scope.append('Branch_2')
        br2 = c(x, 64, 1, 'Conv2d_0a_1x1'); br2 = c(br2, 96, 3, 'Conv2d_0b_3x3')
        br2 = c(br2, 96, 3, 'Conv2d_0c_3x3'); net.scope.pop()
        net.scope.append('Branch_3')
        br3 = net.avg_pool(x, 3, 1, 'SAME'); br3 = c(br3, pool_c, 1, 'Conv2d_0b_1x1'); net.scope.pop()
        x = cat([b0, br1, br2, br3]); end_points[blk] = x
        net.scope.pop()

    # Mixed_6a (inception_v3.py:213-228)
    net.scope.append('Mixed_6a')
    net.scope.append('Branch_0'); b0 = c(x, 384, 3, 'Conv2d_1a_1x1', 2, 'VALID'); net.scope.pop()
    net.scope.append('Branch_1')
    br1 = c(x, 64, 1, 'Conv2d_0a_1x1'); br1 = c(br1, 96, 3, 'Conv2d_0b_3x3')
    br1 = c(br1, 96, 3, 'Conv2d_1a_1x1', 2, 'VALID'); net.scope.pop()
    br2 = net.max_pool(x, 3, 2, 'VALID')
    x = cat([b0, br1, br2]); end_points['Mixed_6a'] = x
    net.scope.pop()

    # Mixed_6b..6e (inception_v3.py:231-344)
    for blk, d in (('Mixed_6b', 128), ('Mixed_6c', 160), ('Mixed_6d', 160), ('Mixed_6e', 192)):
        net.scope.append(blk)
        net.scope.append('Branch_0'); b0 = c(x, 192, 1, 'Conv2d_0a_1x1'); net.scope.pop()
        net.scope.append('Branch_1')
        br1 = c(x, d, 1, 'Conv2d_0a_1x1'); br1 = c(br1, d, (1, 7), 'Conv2d_0b_1x7')
        br1 = c(br1, 192, (7, 1), 'Conv2d_0c_7x1'); net.scope.pop()
        net.scope.append('Branch_2')
        br2 = c(x, d, 1, 'Conv2d_0a_1x1'); br2 = c(br2, d, (7, 1), 'Conv2d_0b_7x1')
        br2 = c(br2, d, (1, 7), 'Conv2d_0c_1x7'); br2 = c(br2, d, (7, 1), 'Conv2d_0d_7x1')
        br2 = c(br2, 192, (1, 7), 'Conv2d_0e_1x7'); net.scope.pop()
        net.scope.append('Branch_3')
        br3 = net.avg_pool(x, 3, 1, 'SAME'); br3 = c(br3, 192, 1, 'Conv2d_0b_1x1'); net.scope.pop()
        x = cat([b0, br1, br2, br3]); end_points[blk] = x
        net.scope.pop()

    # Mixed_7a (inception_v3.py:347-365)
    net.scope.append('Mixed_7a')
    net.scope.append('Branch_0')
    b0 = c(x, 192, 1, 'Conv2d_0a_1x1'); b0 = c(b0, 320, 3, 'Conv2d_1a_3x3', 2, 'VALID'); net.scope.pop()
    net.scope.append('Branch_1')
    br1 = c(x, 192, 1, 'Conv2d_0a_1x1'); br1 = c(br1, 192, (1, 7), 'Conv2d_0b_1x7')
    br1 = c(br1, 192, (7, 1), 'Conv2d_0c_7x1'); br1 = c(br1, 192, 3, 'Conv2d_1a_3x3', 2, 'VALID')
    net.scope.pop()
    br2 = net.max_pool(x, 3, 2, 'VALID')
    x = cat([b0, br1, br2]); end_points['Mixed_7a'] = x
    net.scope.pop()

    # Mixed_7b / 7c (inception_v3.py:367-413); 7b Branch_1 second 3x1 conv is
    # scoped 'Conv2d_0b_3x1', 7c's is 'Conv2d_0c_3x1' (reference naming quirk).
    for blk, b1b in (('Mixed_7b', 'Conv2d_0b_3x1'), ('Mixed_7c', 'Conv2d_0c_3x1')):
        net.scope.append(blk)
        net.scope.append('Branch_0'); b0 = c(x, 320, 1, 'Conv2d_0a_1x1'); net.scope.pop()
        net.scope.append('Branch_1')
        br1 = c(x, 384, 1, 'Conv2d_0a_1x1')
        br1 = cat([c(br1, 384, (1, 3), 'Conv2d_0b_1x3'), c(br1, 384, (3, 1), b1b)]); net.scope.pop()
        net.scope.append('Branch_2')
        br2 = c(x, 448, 1, 'Conv2d_0a_1x1'); br2 = c(br2, 384, 3, 'Conv2d_0b_3x3')
        br2 = cat([c(br2, 384, (1, 3), 'Conv2d_0c_1x3'), c(br2, 384, (3, 1), 'Conv2d_0d_3x1')])
        net.scope.pop()
        net.scope.append('Branch_3')
        br3 = net.avg_pool(x, 3, 1, 'SAME'); br3 = c(br3, 192, 1, 'Conv2d_0b_1x1'); net.scope.pop()
        x = cat([b0, br1, br2, br3]); end_points[blk] = x
        net.scope.pop()
    net.scope.pop()
    return x


def _inception_v1_base(net: _Net, x, end_points):
    """common/nets/inception_v1.py:29-266 (conv2d / max_pool2d default stride 1, SAME; conv +
    BN(inference) + ReLU under inception_v1_arg_scope = inception_utils.inception_arg_scope)."""
    net.scope.append('InceptionV1')
    x = net.conv(x, 64, 7, 2, 'SAME', 'Conv2d_1a_7x7'); end_points['Conv2d_1a_7x7'] = x
    x = net.max_pool(x, 3, 2, 'SAME'); end_points['MaxPool_2a_3x3'] = x
    x = net.conv(x, 64, 1, 1, 'SAME', 'Conv2d_2b_1x1'); end_points['Conv2d_2b_1x1'] = x
    x = net.conv(x, 192, 3, 1, 'SAME', 'Conv2d_2c_3x3'); end_points['Conv2d_2c_3x3'] = x
    x = net.max_pool(x, 3, 2, 'SAME'); end_points['MaxPool_3a_3x3'] = x

    def block(x, name, c0, c1a, c1b, c2a, c2b, c3, quirk=False):
        net.scope.append(name)
        net.scope.append('Branch_0'); b0 = net.conv(x, c0, 1, 1, 'SAME', 'Conv2d_0a_1x1'); net.scope.pop()
        net.scope.append('Branch_1')
        b1 = net.conv(x, c1a, 1, 1, 'SAME', 'Conv2d_0a_1x1'); b1 = net.conv(b1, c1b, 3, 1, 'SAME', 'Conv2d_0b_3x3')
        net.scope.pop()
        net.scope.append('Branch_2')
        b2 = net.conv(x, c2a, 1, 1, 'SAME', 'Conv2d_0a_1x1')
        b2 = net.conv(b2, c2b, 3, 1, 'SAME', 'Conv2d_0a_3x3' if quirk else 'Conv2d_0b_3x3')   # :240 scope quirk in Mixed_5b
        net.scope.pop()
        net.scope.append('Branch_3')
        b3 = net.max_pool(x, 3, 1, 'SAME'); b3 = net.conv(b3, c3, 1, 1, 'SAME', 'Conv2d_0b_1x1'); net.scope.pop()
        net.scope.pop()
        y = net.concat([b0, b1, b2, b3])
        end_points[name] = y
        return y
    x = block(x, 'Mixed_3b', 64, 96, 128, 16, 32, 32)
    x = block(x, 'Mixed_3c', 128, 128, 192, 32, 96, 64)
    x = net.max_pool(x, 3, 2, 'SAME'); end_points['MaxPool_4a_3x3'] = x
    x = block(x, 'Mixed_4b', 192, 96, 208, 16, 48, 64)
    x = block(x, 'Mixed_4c', 160, 112, 224, 24, 64, 64)
    x = block(x, 'Mixed_4d', 128, 128, 256, 24, 64, 64)
    x = block(x, 'Mixed_4e', 112, 144, 288, 32, 64, 64)
    x = block(x, 'Mixed_4f', 256, 160, 320, 32, 128, 128)
    x = net.max_pool(x, 2, 2, 'SAME'); end_points['MaxPool_5a_2x2'] = x
    x = block(x, 'Mixed_5b', 256, 160, 320, 32, 128, 128, quirk=True)
    x = block(x, 'Mixed_5c', 384, 192, 384, 48, 128, 128)
    net.scope.pop()
    return x


def _run_v1(net: _Net, images):
    end_points = OrderedDict()
    x = _inception_v1_base(net, np.asarray(images, np.float32), end_points)
    pooled = net.avg_pool(x, (7, 7), 1, 'VALID')        # inception_v1.py:326 (fixed 7x7 kernel)
    end_points['AvgPool_0a_7x7'] = pooled
    return pooled, end_points


def init_params_v1(seed=0, image_size=224):
    net = _Net(None, np.random.default_rng(seed), run=False)
    _run_v1(net, np.zeros((1, image_size, image_size, 3), np.float32))
    return net.params


def inception_v1(params, images, act_dtype='f32'):
    """-> (net [B,1,1,1024], end_points): get_network_fn('inception_v1', num_classes=None,
    is_training=False) -- the reference's default backbone (train.py:56,65: Mixed_4f feature map)."""
    return _run_v1(_Net(params, None, act_dtype=act_dtype, run=True), images)


def inception_v1_grads(params, images, d_net, d_fm, fm_name='Mixed_4f', act_dtype='f32', override=None):
    n = _Net(params, None, act_dtype=act_dtype, run=True, tape=True, override=override)
    net, ep = _run_v1(n, images)
    seeds = []
    if d_net is not None:
        seeds.append((net, np.asarray(d_net, np.float32).reshape(net.shape)))
    if d_fm is not None:
        seeds.append((ep[fm_name], np.asarray(d_fm, np.float32).reshape(ep[fm_name].shape)))
    return n.backward(seeds), net, ep


def _run(net: _Net, images):
    end_points = OrderedDict()
    x = _inception_v3_base(net, np.asarray(images, np.float32), end_points)
    # head: inception_v3.py:520-532 -- kernel = min(8, H_f), VALID, num_classes=None
    k = (min(x.shape[1], 8), min(x.shape[2], 8))
    pooled = net.avg_pool(x, k, 1, 'VALID')
    end_points['AvgPool_1a'] = pooled
    return pooled, end_points


def init_params(seed=0, image_size=224):
    """Variance-scaling conv weights, BN beta 0 / mean 0 / var 1."""
    net = _Net(None, np.random.default_rng(seed), run=False)
    _run(net, np.zeros((1, image_size, image_size, 3), np.float32))
    return net.params


def randomize_bn(params, seed=1):
    """Give BN statistics non-trivial values so parity tests exercise them."""
    rng = np.random.default_rng(seed)
    for k in params:
        n = params[k].shape[0]
        if k.endswith('BatchNorm/beta'):
            params[k] = (0.1 * rng.standard_normal(n)).astype(np.float32)
        elif k.endswith('moving_mean'):
            params[k] = (0.1 * rng.standard_normal(n)).astype(np.float32)
        elif k.endswith('moving_variance'):
            params[k] = rng.uniform(0.5, 1.5, n).astype(np.float32)
    return params


def inception_v3(params, images, act_dtype='f32'):
    """-> (net [B,1,1,2048], end_points).  nets_factory.get_network_fn('inception_v3',
    num_classes=None, is_training=False) (nets/nets_factory.py:116-159)."""
    net = _Net(params, None, act_dtype=act_dtype, run=True)
    return _run(net, images)


def inception_v3_grads(params, images, d_net, d_fm, fm_name='Mixed_7c', act_dtype='f32', override=None):
    """Gradients of the CNN variables given d(net [B,1,1,C]) and d(end_points[fm_name]) (the
    two tensors ModelBase._encoder hands to the decoder, model_base.py:93-104).
    -> (grads {name: array}, net, end_points)."""
    # act_dtype='bf16': the taped forward emulates the product's bf16 storage (so the ReLU masks and
    # pool arg-maxima are those of a bf16 forward); the reverse pass itself stays fp32
    n = _Net(params, None, act_dtype=act_dtype, run=True, tape=True, override=override)
    net, ep = _run(n, images)
    seeds = []
    if d_net is not None:
        seeds.append((net, np.asarray(d_net, np.float32).reshape(net.shape)))
    if d_fm is not None:
        seeds.append((ep[fm_name], np.asarray(d_fm, np.float32).reshape(ep[fm_name].shape)))
    return n.backward(seeds), net, ep


def describe(image_size=224):
    """Layer table: list of conv records and MACs/image."""
    net = _Net(None, np.random.default_rng(0), run=False)
    _run(net, np.zeros((1, image_size, image_size, 3), np.float32))
    return net.conv_log, net.macs, net.params


def encoder(params, images, fm_name='Mixed_7c', act_dtype='f32'):
    """ModelBase._encoder (src/model_base.py:56-104), non-legacy:
    im_embed = squeeze(net) [B,C_g]; cnn_fmaps = reshape(end_points[fm], [B,H*W,C])."""
    net, ep = inception_v3(params, images, act_dtype)
    fm = ep[fm_name]
    B, H, W, C = fm.shape
    return net.reshape(B, -1), fm.reshape(B, H * W, C)
