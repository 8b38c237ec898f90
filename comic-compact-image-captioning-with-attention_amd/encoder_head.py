"""Legacy image-embedding head of `ModelBase._encoder` (reference src/model_base.py:80-91, `--legacy`):

    net      = ops.layer_norm_activate('LN_tanh', squeeze(net), tanh)      common/ops.py:241-275
    im_embed = ops.linear('im_embed', net, 1024, bias_init=None)           common/ops.py:200-238

Variables `Model/encoder/LN_tanh/{beta,gamma}` and `Model/encoder/im_embed/weight` sit outside the frozen scope
`Model/encoder/cnn`, so decoder-mode training updates them (model_base.py:834-849); the gradient stops at the CNN
output (the reference refuses cnn_finetune / scst with --legacy, train.py:241-249).
"""
from __future__ import annotations

import math

import numpy as np

from . import _lib as L
from .decoder import FlatParams

LN_EPS = 1e-12          # tf.contrib.layers.layer_norm variance_epsilon [TF-1.9]
HEAD_DIM = 1024
TF_NAMES = {'ln_beta': 'Model/encoder/LN_tanh/beta', 'ln_gamma': 'Model/encoder/LN_tanh/gamma',
            'W': 'Model/encoder/im_embed/weight'}


def init_params(c_in, seed=0, out_dim=HEAD_DIM):
    """LayerNorm beta 0 / gamma 1; `weight` from the scope's Xavier-uniform initialiser (model.py:41-42)."""
    rng = np.random.default_rng(seed + 977)
    lim = math.sqrt(6.0 / (c_in + out_dim))
    return dict(ln_beta=np.zeros(c_in, np.float32), ln_gamma=np.ones(c_in, np.float32),
                W=rng.uniform(-lim, lim, (c_in, out_dim)).astype(np.float32))


class LegacyEncoderHead:
    def __init__(self, c_in, params=None, device='cuda:0', seed=0, out_dim=HEAD_DIM):
        import torch
        self.torch, self.lib, self.device = torch, L.load(), device
        self.c_in, self.out_dim = int(c_in), int(out_dim)
        self.params = FlatParams(dict(ln_beta=(c_in,), ln_gamma=(c_in,), W=(c_in, out_dim)), device)
        self.params.load(params if params is not None else init_params(c_in, seed, out_dim))
        self.grads = self.params.like()
        self._buf = {}

    def _bufs(self, B):
        if B not in self._buf:
            t = self.torch
            f32 = dict(dtype=t.float32, device=self.device)
            self._buf[B] = dict(z=t.empty((B, self.c_in), **f32), xhat=t.empty((B, self.c_in), **f32),
                                out=t.empty((B, self.out_dim), **f32), dz=t.empty((B, self.c_in), **f32),
                                pg=t.empty((B, self.c_in), **f32), pb=t.empty((B, self.c_in), **f32))
        return self._buf[B]

    def forward(self, net):
        """net [B, C_in] fp32 (the squeezed pooled CNN output) -> im_embed [B, 1024]."""
        B = int(net.shape[0])
        assert net.shape == (B, self.c_in) and net.dtype == self.torch.float32 and net.is_contiguous()
        b, p, st = self._bufs(B), self.params, L.stream_ptr()
        L.check(self.lib.comic_ln_tanh_fwd(net.data_ptr(), p.view('ln_gamma').data_ptr(), p.view('ln_beta').data_ptr(),
                                           b['z'].data_ptr(), b['xhat'].data_ptr(), B, self.c_in, LN_EPS, st), 'ln_tanh_fwd')
        L.check(self.lib.comic_gemm_f32(b['z'].data_ptr(), p.view('W').data_ptr(), b['out'].data_ptr(), None, B,
                                        self.out_dim, self.c_in, self.c_in, self.out_dim, self.out_dim, 0, 0, 1.0, 0.0, st),
                'head gemm')
        self._last_B = B
        return b['out']

    def backward(self, d_im_embed):
        """d_im_embed [B, 1024] (gradient of `forward`'s output of the same batch) -> fills self.grads."""
        B = self._last_B
        b, p, g, st = self._bufs(B), self.params, self.grads, L.stream_ptr()
        d = d_im_embed.contiguous()
        C_, N = self.c_in, self.out_dim
        # dW = z^T d ; dz = d W^T
        L.check(self.lib.comic_gemm_f32(b['z'].data_ptr(), d.data_ptr(), g.view('W').data_ptr(), None, C_, N, B, C_, N, N,
                                        1, 0, 1.0, 0.0, st), 'head dW')
        L.check(self.lib.comic_gemm_f32(d.data_ptr(), p.view('W').data_ptr(), b['dz'].data_ptr(), None, B, C_, N, N, N, C_,
                                        0, 1, 1.0, 0.0, st), 'head dz')
        L.check(self.lib.comic_ln_tanh_bwd_rows(b['dz'].data_ptr(), b['z'].data_ptr(), b['xhat'].data_ptr(),
                                                b['pg'].data_ptr(), b['pb'].data_ptr(), B, C_, st), 'ln_tanh_bwd_rows')
        L.check(self.lib.comic_colsum(b['pg'].data_ptr(), g.view('ln_gamma').data_ptr(), B, C_, 0.0, st), 'd gamma')
        L.check(self.lib.comic_colsum(b['pb'].data_ptr(), g.view('ln_beta').data_ptr(), B, C_, 0.0, st), 'd beta')
        return g

    def export_params(self):
        """{TF variable name: array}"""
        return {TF_NAMES[k]: v for k, v in self.params.to_numpy().items()}

    def load_named(self, arrays):
        self.params.load({k: arrays[n] for k, n in TF_NAMES.items()})
