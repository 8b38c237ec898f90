"""Optimiser and learning-rate schedule of the reference, on the flat parameter buffer.

`tf.train.AdamOptimizer` / `tf.train.MomentumOptimizer` through `slim.learning.create_train_op` (reference
src/model_base.py:387-401, :852-883) with the TF-1.9 ApplyAdam formula (eps outside the
bias correction), L2 regularisation over every trainable variable
(model_base.py:408-417, common/ops.py:184-190) folded into the update as grad += decay*w,
and the cosine schedule of `_create_cosine_lr` (model_base.py:809-820).
"""
from __future__ import annotations

import math

import numpy as np

from . import _lib as L


def cosine_lr(step, max_step, lr_start, lr_end):
    s = np.float32(step / max_step)
    s = np.float32(1.0) + np.cos(np.minimum(np.float32(1.0), s) * np.float32(math.pi), dtype=np.float32)
    return float(np.float32(np.float32(lr_start - lr_end) * s / np.float32(2) + np.float32(lr_end)))


def legacy_lr_reduce(lr, epoch, lr_end, every_n_epochs):
    """train_fn._lr_reduce_check (src/train_fn.py:307-317)."""
    if lr > lr_end and epoch % every_n_epochs == 0:
        lr = max(lr / 2, lr_end)
    return lr


class GradClip:
    """`--clip_gradient_norm c` (train.py:137 -> model_base.py:394-401): slim.learning.create_train_op clips EVERY variable's
    gradient by that tensor's own norm (clip_gradient_norms -> tf.clip_by_norm [TF-1.9 slim]), after the gradient
    multipliers; the L2 term is part of the loss and therefore of the clipped gradient.  One chunk table per flat
    parameter buffer (comic_clip_by_norm, include/comic_hip.h)."""
    CHUNK = 8192

    def __init__(self, params, clip_norm):
        import torch
        self.lib = L.load()
        self.clip_norm = float(clip_norm)
        rows = []
        for seg, k in enumerate(params.shapes):
            n = int(np.prod(params.shapes[k])) if len(params.shapes[k]) else 1
            first, count = len(rows), (n + self.CHUNK - 1) // self.CHUNK
            for c in range(count):
                rows.append((seg, params.offsets[k] + c * self.CHUNK, min(self.CHUNK, n - c * self.CHUNK), first, count))
        self.n_chunks = len(rows)
        self.chunks = torch.from_numpy(np.asarray(rows, np.int64).reshape(-1, 5)).to(params.device)
        self.partial = torch.zeros(max(1, self.n_chunks), dtype=torch.float32, device=params.device)

    def apply(self, params, grads, l2, grad_scale, skip=None):
        st = skip if skip is not None else getattr(grads, 'status', None)
        L.check(self.lib.comic_clip_by_norm(grads.data.data_ptr(), params.data.data_ptr(), self.chunks.data_ptr(),
                                            self.n_chunks, l2, grad_scale, self.clip_norm, self.partial.data_ptr(),
                                            st.data_ptr() if st is not None else None, L.stream_ptr()), 'clip_by_norm')


class AdamTF:
    def __init__(self, params, beta1=0.9, beta2=0.999, epsilon=1e-2, l2_decay=1e-5, clip_norm=0.0):
        self.lib = L.load()
        self.params = params
        self.m = params.like()
        self.v = params.like()
        self.beta1, self.beta2, self.eps, self.l2 = beta1, beta2, epsilon, l2_decay
        self.t = 0                      # number of applied updates (== global_step)
        self.clip = GradClip(params, clip_norm) if clip_norm and clip_norm > 0 else None

    def step(self, grads, lr, grad_scale=1.0, skip=None, ranges=None, before_range=None):
        """skip: the status word of ANOTHER gradient buffer that gates this update too (the CNN optimisers of cnn_finetune
        follow the decoder's verdict on the step: model.py).
        ranges: [(lo, hi), ...] element ranges of the flat buffer (together: all of it) updated one launch each, in that
        order, `before_range(i)` called in front of launch i -- the chunked gradient exchange (trainer.DataParallel.
        exchange_and_step) orders the stream behind chunk i's all-reduce there, so the update of a chunk runs beside the
        exchange of the next.  The update is element-wise: any partition gives the bits of the one-launch update."""
        self.t += 1
        lr_t = lr * math.sqrt(1.0 - self.beta2 ** self.t) / (1.0 - self.beta1 ** self.t)
        # a gradient buffer with a status word (decoder.FlatParams(status_tail=True)): the update is skipped on the device
        # when comic_decoder_train_step voided the step (on any rank: the word is part of the all-reduced buffer).  The
        # host-side step count `t` advances regardless -- one bias-correction step of drift per voided step, and the run
        # stops at the next log point anyway (train_fn._check_loss reads the sticky count).
        st = skip if skip is not None else getattr(grads, 'status', None)
        if self.clip is not None:
            assert ranges is None, 'per-variable clipping reads whole variables: one launch over the flat buffer'
            self.clip.apply(self.params, grads, self.l2, grad_scale, skip=st)
        for i, (lo, hi) in enumerate(ranges or [(0, self.params.numel)]):
            if before_range is not None:
                before_range(i)
            L.check(self.lib.comic_adam_tf_gated(self.params.data.data_ptr() + 4 * lo, grads.data.data_ptr() + 4 * lo,
                                                 self.m.data.data_ptr() + 4 * lo, self.v.data.data_ptr() + 4 * lo, hi - lo,
                                                 lr_t, self.beta1, self.beta2, self.eps, self.l2, grad_scale,
                                                 st.data_ptr() if st is not None else None, L.stream_ptr()), 'adam_tf')

    def state_dict(self):
        return dict(t=self.t, m=self.m.data.clone(), v=self.v.data.clone())

    def load_state_dict(self, sd):
        self.t = int(sd['t'])
        self.m.data.copy_(sd['m'])
        self.v.data.copy_(sd['v'])


class MomentumTF:
    """tf.train.MomentumOptimizer(lr, momentum=0.9, use_nesterov=False) (reference src/model_base.py:867-880;
    `--optimiser sgd`): accum = momentum*accum + g, w -= lr*accum, with the same L2 fold as AdamTF.  Same interface
    (`t`, `step`, `m` = the accumulator; `v` stays zero so checkpoints keep one layout)."""

    def __init__(self, params, momentum=0.9, l2_decay=1e-5, clip_norm=0.0, **_):
        self.lib = L.load()
        self.params = params
        self.m = params.like()
        self.v = params.like()
        self.momentum, self.l2 = momentum, l2_decay
        self.beta1, self.beta2, self.eps = momentum, 0.0, 0.0
        self.t = 0
        self.clip = GradClip(params, clip_norm) if clip_norm and clip_norm > 0 else None

    def step(self, grads, lr, grad_scale=1.0, skip=None, ranges=None, before_range=None):
        """ranges / before_range: as AdamTF.step."""
        self.t += 1
        st = skip if skip is not None else getattr(grads, 'status', None)
        if self.clip is not None:
            assert ranges is None, 'per-variable clipping reads whole variables: one launch over the flat buffer'
            self.clip.apply(self.params, grads, self.l2, grad_scale, skip=st)
        for i, (lo, hi) in enumerate(ranges or [(0, self.params.numel)]):
            if before_range is not None:
                before_range(i)
            L.check(self.lib.comic_momentum_tf_gated(self.params.data.data_ptr() + 4 * lo, grads.data.data_ptr() + 4 * lo,
                                                     self.m.data.data_ptr() + 4 * lo, hi - lo, lr, self.momentum, self.l2,
                                                     grad_scale, st.data_ptr() if st is not None else None, L.stream_ptr()),
                    'momentum_tf')

    state_dict = AdamTF.state_dict
    load_state_dict = AdamTF.load_state_dict


def make_optimiser(name, params, epsilon=1e-2, l2_decay=1e-5, clip_norm=0.0):
    """`_get_optimiser` (model_base.py:852-883) + create_train_op's clip_gradient_norm (model_base.py:394-401)."""
    if name == 'adam':
        return AdamTF(params, epsilon=epsilon, l2_decay=l2_decay, clip_norm=clip_norm)
    if name == 'sgd':
        return MomentumTF(params, l2_decay=l2_decay, clip_norm=clip_norm)
    raise ValueError('Unknown optimiser.')
