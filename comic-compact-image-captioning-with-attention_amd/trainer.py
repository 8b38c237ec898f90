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

from . import decoder as cdec, nets, optim, streams


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

    def global_xe_denominator(self, wmask_dev):
        """Device tensor (1 element) = sum of the XE weights over ALL ranks / world + 1e-12: the sequence_loss
        normaliser (model_base.py:337-340) a rank uses so that the rank-mean of the gradients equals the gradient of
        the global batch.  The all-reduce is stream-ordered; nothing is read back to the host."""
        t = wmask_dev.sum().reshape(1)
        if self.dist:
            self.dist.all_reduce(t)
        return t / self.world + 1e-12

    # ---- bucketed gradient exchange: buckets are reduced on a side stream while the backward pass goes on ----------
    def max_scalar(self, value):
        """max over the ranks of a host number (a collective: every rank calls it at the same point)."""
        if self.world == 1:
            return value
        import torch
        dev = 'cpu' if self.dist.get_backend() == 'gloo' else 'cuda'
        x = torch.tensor([float(value)], dtype=torch.float64, device=dev)
        self.dist.all_reduce(x, op=self.dist.ReduceOp.MAX)
        return float(x.item())

    def reduce_async(self, flat_slice):
        """Start the sum all-reduce of `flat_slice` (a contiguous view of a flat gradient buffer whose producers have
        been enqueued on the current stream).  CUDA tensors: the collective is issued from a communication stream
        that first waits for the current stream, so later kernels of the caller are not ordered behind it; finish
        with `wait_all()`.  CPU tensors (gloo tests): reduced at once."""
        if not self.dist:
            return
        if not flat_slice.is_cuda:
            self.dist.all_reduce(flat_slice)
            return
        import torch
        if getattr(self, '_comm', None) is None:
            self._comm = streams.lane(torch, flat_slice.device, 'comm')
            self._pending = []
        self._comm.wait_stream(torch.cuda.current_stream())
        with torch.cuda.stream(self._comm):
            self._pending.append(self.dist.all_reduce(flat_slice, async_op=True))

    def wait_all(self):
        """Order the current stream behind every collective started with reduce_async (no host wait).
        Returns the 1/world factor for the optimiser's grad_scale."""
        for w in getattr(self, '_pending', None) or []:
            w.wait()                      # ProcessGroupNCCL: makes the CURRENT stream wait for the collective
        if getattr(self, '_pending', None):
            self._pending = []
        return 1.0 / self.world

    @staticmethod
    def chunk_bounds(params, n_chunks=4):
        """Cut a FlatParams buffer into <= n_chunks contiguous element ranges on variable boundaries, of about equal size
        (a variable is never split: COMIC-256's 22.8 MB are W_init 6.3, K 10.5, W_m 4.2 MB and 1.8 MB of small ones).
        -> [(lo, hi), ...] ascending, together [0, numel)."""
        cuts = sorted(params.offsets.values())[1:]
        target = params.numel / float(max(1, n_chunks))
        bounds, lo = [], 0
        for c in cuts:
            if c - lo >= target and len(bounds) < n_chunks - 1:
                bounds.append((lo, c))
                lo = c
        bounds.append((lo, params.numel))
        return bounds

    def exchange_and_step(self, opt, grads, lr, n_chunks=4):
        """Rank-mean of the flat gradient + optimiser step.  One process: the plain step.  Several ranks, device tensors:
        the buffer travels in `n_chunks` pieces (chunk_bounds) issued back to back on the communication stream, the LAST
        range first -- it carries the status word behind the variables (a step voided on any rank gates every chunk's
        update, so that word must arrive before the first update) -- and the optimiser updates a range as soon as ITS
        all-reduce is done: the update of chunk i runs beside the exchange of chunk i + 1 instead of behind one 22.8 MB
        all-reduce.  Same sums, same element-wise update: bit for bit the flat exchange + one-launch step
        (tests/test_dp_gloo.py, tests/test_gpu_dp.py).  Unmeasured on a multi-GPU node."""
        if self.world == 1:
            opt.step(grads, lr, grad_scale=1.0)
            return
        if getattr(opt, 'clip', None) is not None or n_chunks <= 1 or not grads.data.is_cuda:
            opt.step(grads, lr, grad_scale=self.average_(grads.data))
            return
        key = (id(grads), n_chunks)
        if getattr(self, '_chunk_key', None) != key:
            self._chunk_key, self._chunks = key, self.chunk_bounds(grads, n_chunks)[::-1]
        self.wait_all()                                   # (nothing of an earlier exchange is left pending)
        tail = grads.data.numel() - grads.numel
        for j, (lo, hi) in enumerate(self._chunks):
            self.reduce_async(grads.data[lo:hi + (tail if j == 0 else 0)])
        pending, self._pending = self._pending, []
        opt.step(grads, lr, grad_scale=1.0 / self.world, ranges=self._chunks, before_range=lambda i: pending[i].wait())

    def average_(self, flat):
        """In-place rank-mean of a flat gradient tensor (sum all-reduce, then 1/W in the
        optimiser's grad_scale to save a pass)."""
        if self.dist:
            self.dist.all_reduce(flat)
        return 1.0 / self.world


def side_stream(torch, device):
    """The stream the frozen encoder's forward runs on beside the decoder.  Priorities on this runtime are 0 (default,
    lowest) and -1; raising either stream above the other measured slower (COMIC_SIDE_PRIORITY to experiment)."""
    import os
    return streams.lane(torch, device, 'encoder', priority=int(os.environ.get('COMIC_SIDE_PRIORITY', '0')))


class EncoderPipeline:
    """Frozen-CNN pipelining: ONE encoder forward covers the image batches of the next `group` training steps
    (batch group*B) and runs on a second stream under the decoder steps of the current group; its outputs go to one
    of two staging copies from which the steps take their B rows.  Used by CaptionTrainer (bench.py) and by
    CaptionModel.run_train_step (the reference-API path)."""

    def __init__(self, encoder, batch, group, device, use_graph=True):
        import torch
        self._torch = torch
        self.encoder, self.batch, self.group, self.use_graph = encoder, int(batch), int(group), use_graph
        assert self.group >= 1 and encoder.batch == self.batch * self.group
        self.side = side_stream(torch, device)
        self._ev_ready = torch.cuda.Event()
        self._ev_done = [torch.cuda.Event(), torch.cuda.Event()]
        self._ev_free = [torch.cuda.Event(), torch.cuda.Event()]
        self._stage = None
        self._n_sub = self._n_taken = 0

    @property
    def steps_ready(self):
        """training steps whose features are submitted and not yet taken"""
        return self._n_sub * self.group - self._n_taken

    def submit(self, images, events=None):
        """Start the forward of the next group (images [group*B,H,W,3] fp32 on the device) on the side stream.
        events: an optional (start, end) pair of timing events recorded on the side stream around the forward."""
        torch = self._torch
        main = torch.cuda.current_stream()
        par = self._n_sub % 2
        self._ev_ready.record(main)                         # the images (and everything before) are ready
        self.side.wait_event(self._ev_ready)
        if self._n_sub >= 2:
            self.side.wait_event(self._ev_free[par])        # every step of the group two back has taken its rows
        with torch.cuda.stream(self.side):
            if events: events[0].record(self.side)
            im, fm = self.encoder.forward(images, use_graph=self.use_graph)
            if events: events[1].record(self.side)
            if self._stage is None:
                self._stage = [(torch.empty_like(im), torch.empty_like(fm)) for _ in range(2)]
            self._stage[par][0].copy_(im)
            self._stage[par][1].copy_(fm)
            self._ev_done[par].record(self.side)
        self._n_sub += 1

    def take(self):
        """(im_embed, fm, release) of the next step out of the submitted groups; call release() once the decoder
        holds its copy (it returns True when this was the FIRST step of its group: the moment to submit the
        next group, which then has the whole group's decoder steps to run under)."""
        assert self.steps_ready > 0, 'submit() first'
        main = self._torch.cuda.current_stream()
        g, j = divmod(self._n_taken, self.group)
        par = g % 2
        if j == 0:
            main.wait_event(self._ev_done[par])
        self._n_taken += 1
        B = self.batch
        im, fm = self._stage[par]

        def release():
            if j == self.group - 1:
                self._ev_free[par].record(main)
            return j == 0
        return im[j * B:(j + 1) * B], fm[j * B:(j + 1) * B], release


class CaptionTrainer:
    def __init__(self, cnn_params, dec_spec, dec_params=None, batch=64, image_size=(224, 224), cnn_dtype='bf16',
                 device='cuda:0', lr_start=1e-2, lr_end=1e-5, max_step=100000, adam_epsilon=1e-2, dp=None, seed=0,
                 plan=None, encoder_group=1):
        self.plan = plan or nets.CnnPlan('inception_v3', image_size)
        # encoder_group S > 1 (frozen-CNN pipelining only): ONE encoder forward covers the image batches of S
        # consecutive steps (batch S*B: fuller tiles, S times fewer launches per image); the decoder steps take
        # their B rows from a double-buffered copy of its outputs (submit_images / xe_step_pending)
        self.group = int(encoder_group)
        assert self.group >= 1
        self.encoder = nets.CnnEncoder(self.plan, cnn_params, batch * self.group, cnn_dtype, device)
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
        self._side = side_stream(torch, device)
        self._ev_cnn = torch.cuda.Event()
        self._ev_used = torch.cuda.Event()
        self._pending = None
        self._pipe = EncoderPipeline(self.encoder, batch, self.group, device) if self.group > 1 else None

    @property
    def global_step(self):
        return self.opt.t

    def lr(self):
        return optim.cosine_lr(self.global_step, self.max_step, self.lr_start, self.lr_end)

    def xe_step(self, images, captions, masks=None, training=True):
        """images [B,H,W,3] fp32 device tensor, captions [B,L] int (PAD -1) -> dict(loss, map_loss)."""
        im_embed, fm = self.encoder.forward(images, use_graph=self.use_graph)
        cap = np.asarray(captions)
        res = self.decoder.train_step(fm, im_embed, cap, masks=masks, training=training, dp=self.dp,
                                      use_graph=self.use_graph_decoder)
        self.dp.exchange_and_step(self.opt, self.decoder.grads, self.lr())
        return res

    def enable_cnn_finetune(self, cnn_grad_multiplier=1.0, autotune_backward=False, tune_cache=None):
        """train_mode cnn_finetune (train.py:241-249): the CNN variables join the trainable set.
        autotune_backward: time the kernel variants of every backward-data convolution once (CnnEncoder.autotune_backward)."""
        mult = float(cnn_grad_multiplier)
        l2 = self.opt.l2 * mult
        self.opt_cnn = (optim.AdamTF(self.encoder.w_master, epsilon=self.opt.eps, l2_decay=l2),
                        optim.AdamTF(self.encoder.beta, epsilon=self.opt.eps, l2_decay=l2), mult)
        self.encoder.enable_training()
        if autotune_backward:
            self.encoder.autotune_backward(cache=tune_cache)

    def finetune_step(self, images, captions, masks=None, training=True):
        """One cnn_finetune update: CNN forward -> decoder forward/backward (with input gradients)
        -> CNN backward -> [all-reduce] -> TF-Adam on decoder and CNN variables -> weight refresh."""
        assert getattr(self, 'opt_cnn', None), 'enable_cnn_finetune() first'
        im_embed, fm = self.encoder.forward(images, use_graph=self.use_graph)
        cap = np.asarray(captions)
        res = self.decoder.train_step(fm, im_embed, cap, masks=masks, training=training, dp=self.dp,
                                      use_graph=False, want_input_grads=True)
        lr = self.lr()
        ow, ob, mult = self.opt_cnn
        if self.dp.world > 1:
            # decoder gradient first, then each finished bucket of the CNN gradient, reduced on the communication
            # stream under the backward of the earlier blocks
            self.dp.reduce_async(self.decoder.grads.data)
            if getattr(self, '_buckets', None) is None:
                self._buckets = self.encoder.grad_buckets(6)

            def exchange(t, bk):
                self.dp.reduce_async(t.dw.data[bk[2][0]:bk[2][1]])
                self.dp.reduce_async(t.dbeta.data[bk[3][0]:bk[3][1]])
            t = self.encoder.backward(res['dfm'], res['dim_embed'], self._buckets, exchange)
            scale = self.dp.wait_all()
        else:
            t = self.encoder.backward(res['dfm'], res['dim_embed'])
            scale = 1.0
        self.opt.step(self.decoder.grads, lr, grad_scale=scale)
        ow.t = ob.t = self.opt.t - 1
        void = getattr(self.decoder.grads, 'status', None)     # a step voided on the device skips the CNN update too
        ow.step(t.dw, lr, grad_scale=scale * mult, skip=void)
        ob.step(t.dbeta, lr, grad_scale=scale * mult, skip=void)
        self.encoder.refresh_weights()
        self.encoder.clear_grads_async()
        return res

    def enable_overlap(self, polite_lds_kb=84):
        """Prepare the encoder for running under the decoder step (submit_images / xe_step_pending): its
        conv workgroups take a whole CU's LDS each, so the decoder's kernels always find wave slots."""
        if self.encoder.polite_lds_kb != polite_lds_kb:
            self.encoder.polite_lds_kb = polite_lds_kb
            self.encoder._drop_graphs()                             # re-capture with the new launch parameters

    def submit_images(self, images, events=None):
        """Start the encoder forward of a future step (group > 1: of the next `group` steps, images
        [group*B,H,W,3]) on the side stream (frozen-CNN modes).  events: an optional (start, end) pair of timing
        events recorded on the side stream around the forward."""
        torch = self._torch
        main = torch.cuda.current_stream()
        if self.group == 1:
            self._side.wait_stream(main) if self._pending is None else self._side.wait_event(self._ev_used)
            with torch.cuda.stream(self._side):
                if events: events[0].record(self._side)
                self._pending = self.encoder.forward(images, use_graph=self.use_graph)
                if events: events[1].record(self._side)
                self._ev_cnn.record(self._side)
            return
        self._pipe.use_graph = self.use_graph
        self._pipe.submit(images, events)

    def take_features(self):
        """group > 1: (im_embed, fm, release) of the next step (EncoderPipeline.take)."""
        return self._pipe.take()

    def xe_step_pending(self, captions, next_images=None, masks=None, training=True):
        """XE step on the batch submitted earlier; `next_images` (if given) are submitted as soon
        as the decoder has taken its copy of the encoder outputs, overlapping with this step."""
        torch = self._torch
        main = torch.cuda.current_stream()
        if self.group > 1:
            return self._xe_step_grouped(captions, next_images, masks, training)
        assert self._pending is not None, 'submit_images() first'
        main.wait_event(self._ev_cnn)
        im_embed, fm = self._pending
        cap = np.asarray(captions)

        def consumed():
            self._ev_used.record(main)
            if next_images is not None:
                self.submit_images(next_images)
        res = self.decoder.train_step(fm, im_embed, cap, masks=masks, training=training, dp=self.dp,
                                      use_graph=self.use_graph_decoder, on_inputs_consumed=consumed)
        if next_images is None:
            self._pending = None
        self.dp.exchange_and_step(self.opt, self.decoder.grads, self.lr())
        return res

    def _xe_step_grouped(self, captions, next_images, masks, training):
        """xe_step_pending with encoder_group > 1: `next_images` ([group*B,H,W,3], the batches of the NEXT group
        of steps) are taken at the first step of a group and ignored at the others."""
        im_embed, fm, release = self.take_features()
        cap = np.asarray(captions)

        def consumed():
            if release() and next_images is not None:
                self.submit_images(next_images)
        res = self.decoder.train_step(fm, im_embed, cap, masks=masks, training=training, dp=self.dp,
                                      use_graph=self.use_graph_decoder, on_inputs_consumed=consumed)
        self.dp.exchange_and_step(self.opt, self.decoder.grads, self.lr())
        return res

    def scst_step(self, images, hypo_ids, rewards, masks=None, training=True, tile=1):
        """train_fn_scst's train run (train_fn.py:251-256).  images: tiled by the beam size (tile=1) or
        the untiled batch with tile=beam (encoder once, outputs tiled: identical values)."""
        im_embed, fm = self.encoder.forward(images, use_graph=self.use_graph)
        if tile > 1:
            im_embed, fm = im_embed.repeat(tile, 1), fm.repeat(tile, 1, 1)
        res = self.decoder.train_step(fm, im_embed, hypo_ids, masks=masks, rewards=rewards, training=training,
                                      use_graph=self.use_graph_decoder)
        self.dp.exchange_and_step(self.opt, self.decoder.grads, self.lr())
        return res
