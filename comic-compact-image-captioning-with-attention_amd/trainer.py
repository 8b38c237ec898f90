"""One process per GPU training harness for the hot path (decoder-mode XE, SCST step).

Counterpart of the per-step body of `train_fn.train_fn` / `train_fn_scst`
(reference src/train_fn.py:91-144, :218-256): CNN forward (frozen, BN in inference mode,
model_base.py:72-77) -> decoder forward/backward -> [RCCL all-reduce of the flat gradient]
-> fused TF-Adam.  The reference has no multi-GPU path; data parallelism here shards the
batch by rows: one all-reduce of the flat fp32 gradient buffer per step
(22.8 MB for COMIC-256 on InceptionV3), none inside the forward/backward or the SCST rollout.
"""
from __future__ import annotations

import numpy as np

from . import decoder as cdec, nets, optim


class DataParallel:
    """Thin wrapper over torch.distributed (backend 'nccl' == RCCL on ROCm, 'gloo' on CPU)."""

    def __init__(self, dist=None):
        self.dist = dist if (dist is not None and dist.is_available() and dist.is_initialized()) else None
        self.world = self.dist.get_world_size() if self.dist else 1
        self.rank = self.dist.get_rank() if self.dist else 0

    def shard(self, n_rows):
        """Row range of this rank in a global batch of n_rows (contiguous, deterministic)."""
        per = n_rows // self.world
        assert per * self.world == n_rows, 'global batch must divide by the world size'
        return self.rank * per, (self.rank + 1) * per

    def global_tokens(self, local_tokens, device):
        if not self.dist:
            return float(local_tokens)
        import torch
        t = torch.tensor([float(local_tokens)], dtype=torch.float64, device=device)
        self.dist.all_reduce(t)
        return float(t.item())

    def average_(self, flat):
        """In-place rank-mean of a flat gradient tensor (sum all-reduce, then 1/W in the
        optimiser's grad_scale to save a pass)."""
        if self.dist:
            self.dist.all_reduce(flat)
        return 1.0 / self.world


class CaptionTrainer:
    def __init__(self, cnn_params, dec_spec, dec_params=None, batch=64, image_size=(224, 224), cnn_dtype='bf16',
                 device='cuda:0', lr_start=1e-2, lr_end=1e-5, max_step=100000, adam_epsilon=1e-2, dp=None, seed=0,
                 plan=None):
        self.plan = plan or nets.CnnPlan('inception_v3', image_size)
        self.encoder = nets.CnnEncoder(self.plan, cnn_params, batch, cnn_dtype, device)
        self.decoder = cdec.Decoder(dec_spec, dec_params, device, seed)
        self.opt = optim.AdamTF(self.decoder.params, epsilon=adam_epsilon, l2_decay=dec_spec.l2_decay)
        self.dp = dp or DataParallel(None)
        self.lr_start, self.lr_end, self.max_step = lr_start, lr_end, max_step
        self.device = device
        self.batch = batch
        self.use_graph = True       # hipGraph replay of the CNN plan
        self.use_graph_decoder = False   # eager decoder launches measured faster next to the side stream
        # decoder-mode pipelining: the CNN is frozen, so the encoder forward of the NEXT batch
        # does not depend on this step's update and runs on a second stream under the decoder
        import torch
        self._torch = torch
        self._side = torch.cuda.Stream(device=device)
        self._ev_cnn = torch.cuda.Event()
        self._ev_used = torch.cuda.Event()
        self._pending = None

    @property
    def global_step(self):
        return self.opt.t

    def lr(self):
        return optim.cosine_lr(self.global_step, self.max_step, self.lr_start, self.lr_end)

    def xe_step(self, images, captions, masks=None, training=True):
        """images [B,H,W,3] fp32 device tensor, captions [B,L] int (PAD -1) -> dict(loss, map_loss)."""
        im_embed, fm = self.encoder.forward(images, use_graph=self.use_graph)
        cap = np.asarray(captions)
        local_tokens = float((cap[:, 1:] >= 0).sum())
        denom = None
        if self.dp.world > 1:
            denom = self.dp.global_tokens(local_tokens, self.device) / self.dp.world + 1e-12
        res = self.decoder.train_step(fm, im_embed, cap, masks=masks, training=training, xe_denom=denom,
                                      use_graph=self.use_graph_decoder)
        scale = self.dp.average_(self.decoder.grads.data)
        self.opt.step(self.decoder.grads, self.lr(), grad_scale=scale)
        return res

    def enable_cnn_finetune(self, cnn_grad_multiplier=1.0):
        """train_mode cnn_finetune (train.py:241-249): the CNN variables join the trainable set."""
        mult = float(cnn_grad_multiplier)
        l2 = self.opt.l2 * mult
        self.opt_cnn = (optim.AdamTF(self.encoder.w_master, epsilon=self.opt.eps, l2_decay=l2),
                        optim.AdamTF(self.encoder.beta, epsilon=self.opt.eps, l2_decay=l2), mult)
        self.encoder.enable_training()

    def finetune_step(self, images, captions, masks=None, training=True):
        """One cnn_finetune update: CNN forward -> decoder forward/backward (with input gradients)
        -> CNN backward -> [all-reduce] -> TF-Adam on decoder and CNN variables -> weight refresh."""
        assert getattr(self, 'opt_cnn', None), 'enable_cnn_finetune() first'
        im_embed, fm = self.encoder.forward(images, use_graph=self.use_graph)
        cap = np.asarray(captions)
        denom = None
        if self.dp.world > 1:
            denom = self.dp.global_tokens(float((cap[:, 1:] >= 0).sum()), self.device) / self.dp.world + 1e-12
        res = self.decoder.train_step(fm, im_embed, cap, masks=masks, training=training, xe_denom=denom,
                                      use_graph=False, want_input_grads=True)
        t = self.encoder.backward(res['dfm'], res['dim_embed'])
        lr = self.lr()
        ow, ob, mult = self.opt_cnn
        scale = self.dp.average_(self.decoder.grads.data)
        self.dp.average_(t.dw.data)
        self.dp.average_(t.dbeta.data)
        self.opt.step(self.decoder.grads, lr, grad_scale=scale)
        ow.t = ob.t = self.opt.t - 1
        ow.step(t.dw, lr, grad_scale=scale * mult)
        ob.step(t.dbeta, lr, grad_scale=scale * mult)
        self.encoder.refresh_weights()
        return res

    def enable_overlap(self, polite_lds_kb=84):
        """Prepare the encoder for running under the decoder step (submit_images / xe_step_pending): its
        conv workgroups take a whole CU's LDS each, so the decoder's kernels always find wave slots."""
        if self.encoder.polite_lds_kb != polite_lds_kb:
            self.encoder.polite_lds_kb = polite_lds_kb
            self.encoder._graph, self.encoder._calls = None, 0      # re-capture with the new launch parameters

    def submit_images(self, images):
        """Start the encoder forward of a future step on the side stream (frozen-CNN modes)."""
        torch = self._torch
        main = torch.cuda.current_stream()
        self._side.wait_stream(main) if self._pending is None else self._side.wait_event(self._ev_used)
        with torch.cuda.stream(self._side):
            self._pending = self.encoder.forward(images, use_graph=self.use_graph)
            self._ev_cnn.record(self._side)

    def xe_step_pending(self, captions, next_images=None, masks=None, training=True):
        """XE step on the batch submitted earlier; `next_images` (if given) are submitted as soon
        as the decoder has taken its copy of the encoder outputs, overlapping with this step."""
        assert self._pending is not None, 'submit_images() first'
        torch = self._torch
        main = torch.cuda.current_stream()
        main.wait_event(self._ev_cnn)
        im_embed, fm = self._pending
        cap = np.asarray(captions)
        denom = None
        if self.dp.world > 1:
            denom = self.dp.global_tokens(float((cap[:, 1:] >= 0).sum()), self.device) / self.dp.world + 1e-12

        def consumed():
            self._ev_used.record(main)
            if next_images is not None:
                self.submit_images(next_images)
        res = self.decoder.train_step(fm, im_embed, cap, masks=masks, training=training, xe_denom=denom,
                                      use_graph=self.use_graph_decoder, on_inputs_consumed=consumed)
        if next_images is None:
            self._pending = None
        scale = self.dp.average_(self.decoder.grads.data)
        self.opt.step(self.decoder.grads, self.lr(), grad_scale=scale)
        return res

    def scst_step(self, images, hypo_ids, rewards, masks=None, training=True, tile=1):
        """train_fn_scst's train run (train_fn.py:251-256).  images: tiled by the beam size (tile=1) or
        the untiled batch with tile=beam (encoder once, outputs tiled: identical values)."""
        im_embed, fm = self.encoder.forward(images, use_graph=self.use_graph)
        if tile > 1:
            im_embed, fm = im_embed.repeat(tile, 1), fm.repeat(tile, 1, 1)
        res = self.decoder.train_step(fm, im_embed, hypo_ids, masks=masks, rewards=rewards, training=training,
                                      use_graph=self.use_graph_decoder)
        scale = self.dp.average_(self.decoder.grads.data)
        self.opt.step(self.decoder.grads, self.lr(), grad_scale=scale)
        return res
