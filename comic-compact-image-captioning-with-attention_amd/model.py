"""`CaptionModel` / `CaptionModel_SCST` with the reference's constructor signatures and the
attributes its drivers use (src/model.py:21-141, src/model_base.py).

The reference objects are TF graph builders whose tensors are fetched with `sess.run`; here
the same names are METHODS/attributes over the native executors:

  reference                               this framework
  --------------------------------------  ----------------------------------------------
  sess.run(m.dec_log_ppl)  (train mode)    m.run_train_step(batch) -> dec_log_ppl (one update)
  sess.run(m.dec_log_ppl)  (eval mode)     m.run_eval_step(batch)  -> dec_log_ppl
  sess.run(m.infer_output)                 m.infer(images) -> [dec_preds, attention_maps]
  m.global_step / m.lr / m.update_lr       same names (Python ints / floats)
  m.restore_model(sess, saver, lr)         m.restore_model(lr)
  m_sample.dec_preds_beam / _greedy        m.sample(images) -> (beam ids (W,B,T), greedy ids (B,T))
  sess.run(m.train_scst, feed)             m.run_train_scst(imgs, captions, rewards)
Models built with reuse=True share encoder/decoder/optimiser objects with the first model
(AUTO_REUSE variable sharing, model.py:40-42).
"""
from __future__ import annotations

import os

import numpy as np

from . import checkpoint as ckpt
from . import decoder as cdec
from . import encoder_head, nets, optim, streams
from .trainer import DataParallel

_SHARED = {}          # variable store of the current "graph" (tf.variable_scope('Model', AUTO_REUSE))


def reset_default_graph():
    _SHARED.clear()


class ModelBase(object):
    """Shared construction / restore / decode logic (src/model_base.py)."""

    def __init__(self, config):
        self._config = c = config
        assert c.token_type in ['radix', 'word', 'char']
        self._softmax_size = c.radix_base + 2 if c.token_type == 'radix' else len(c.itow)

    # ---- graph assembly --------------------------------------------------------------
    def _build(self, batch_size, device, dp=None):
        c = self._config
        import torch
        self.torch, self.device = torch, device
        share = _SHARED if self.reuse or _SHARED else _SHARED
        if 'plan' not in share:
            # frozen CNN (every mode but cnn_finetune): forward-only plan with the pool branches rewritten
            frozen = bool(getattr(c, 'freeze_scopes', 'Model/encoder/cnn'))
            # --cnn_dtype bf16x3: hi / lo split activations and filters on the bf16 kernels (nets.CnnPlan(x3=True)),
            # the fast plan at the fp32 parity bar, every train mode (cnn_finetune: csrc/conv.hip conv_backward_x3)
            x3 = getattr(c, 'cnn_dtype', 'bf16') == 'bf16x3'
            plan = nets.get_network_fn(c.cnn_name, num_classes=None, is_training=False)(
                tuple(c.cnn_input_size), c.cnn_fm_attention, pool_after_projection=frozen, fuse_pools=frozen, x3=x3)
            share['plan'] = plan
            share['cnn_params'] = plan.init_params(seed=c.rand_seed % (2 ** 31))
            fm = plan.fm_dims()
            c_net = plan.buffers[plan.pooled][2]
            share['head'] = None
            if getattr(c, 'legacy', False):
                # model_base.py:80-91: LN_tanh + linear(1024) between the pooled CNN output and the decoder
                share['head'] = encoder_head.LegacyEncoderHead(c_net, None, device, seed=c.rand_seed % (2 ** 31))
                c_net = encoder_head.HEAD_DIM
            spec = cdec.DecoderSpec.from_config(c, (fm[0] * fm[1], fm[2]), c_net)
            share['spec'] = spec
            share['decoder'] = cdec.Decoder(spec, None, device, seed=c.rand_seed % (2 ** 31))
            share['encoders'] = {}
        self.plan, self.spec, self.decoder = share['plan'], share['spec'], share['decoder']
        self.head = share['head']
        self._share = share
        self._encoder_for(batch_size)
        self.dp = dp or DataParallel(None)
        if self.dp.world > 1:
            self.decoder.set_dropout_stream(c.rand_seed, self.dp.rank)

    def _encoder_for(self, batch_size):
        """One encoder (activation buffers, hipGraph) per batch size over ONE set of CNN variables,
        like the reference's reuse=True train / eval / infer graphs (train_fn.py:60-66)."""
        encs = self._share['encoders']
        if batch_size not in encs:
            dtype = getattr(self._config, 'cnn_dtype', 'bf16')
            first = next(iter(encs.values()), None)
            enc = encs[batch_size] = nets.CnnEncoder(self.plan, self._share['cnn_params'], batch_size, dtype, self.device,
                                                     weights_from=first)
            # kernel variants per conv launch, timed once on this GPU at this batch size (a few seconds; cached in the
            # run directory).  Every variant gives the same bits.  Skipped for toy problem sizes and when
            # config.cnn_autotune is False; the trainable CNN keeps the heuristic choice (its plan differs).
            H, W = self.plan.buffers[self.plan.input][:2]
            c = self._config
            if (getattr(c, 'cnn_autotune', True) and dtype in ('bf16', 'bf16x3') and str(self.device).startswith('cuda')
                    and batch_size * H * W >= 16 * 224 * 224 and getattr(c, 'freeze_scopes', 'Model/encoder/cnn')):
                log_path = getattr(c, 'log_path', None)
                cache = os.path.join(log_path, 'conv_variants.json') if log_path and os.path.isdir(log_path) else None
                enc.autotune(cache=cache)
        return encs[batch_size]

    @property
    def cnn_trainable(self):
        """train_mode cnn_finetune sets freeze_scopes = '' (train.py:241-249), which puts the CNN
        variables into `_get_trainable_vars` (model_base.py:834-849)."""
        return self.is_training() and not getattr(self._config, 'freeze_scopes', 'Model/encoder/cnn')

    def _encode(self, images):
        images = np.asarray(images, np.float32) if not self.torch.is_tensor(images) else images
        enc = self._encoder_for(int(images.shape[0]))
        if not self.torch.is_tensor(images):
            images = self.torch.from_numpy(np.ascontiguousarray(images)).to(self.device)
        im_embed, fm = enc.forward(images, use_graph=True)
        return self._embed(im_embed), fm

    def _encode_copy(self, images):
        """_encode, the two outputs as copies of their own (the encoder's output buffers belong to its next forward; a decode
        that is still in flight on another stream must not read them)."""
        im_embed, fm = self._encode(images)
        return im_embed.clone(), fm.clone()

    def _embed(self, net):
        """`self.im_embed` of ModelBase._encoder: the squeezed pooled output, or its legacy LN_tanh + linear head."""
        return net if self.head is None else self.head.forward(net.contiguous())

    def is_training(self):
        return self.mode == 'train'

    def _cnn_update(self, res, lr):
        """cnn_finetune: encoder backward from the decoder's input gradients, rank-mean of the CNN
        gradients, TF-Adam on the fp32 masters, refresh of what the forward reads."""
        enc = self._encoder_for(self._batch_size)
        ow, ob, mult = self._share['opt_cnn']
        if self.dp.world > 1:
            # bucketed exchange: the decoder gradient and every finished bucket of the CNN gradient are all-reduced on
            # the communication stream while the backward of the earlier blocks runs
            self.dp.reduce_async(self.decoder.grads.data)
            self._dec_reduced = True
            buckets = self.__dict__.setdefault('_buckets', enc.grad_buckets(int(getattr(self._config, 'grad_buckets', 6))))

            def exchange(t, bk):
                self.dp.reduce_async(t.dw.data[bk[2][0]:bk[2][1]])
                self.dp.reduce_async(t.dbeta.data[bk[3][0]:bk[3][1]])
            t = enc.backward(res['dfm'], res['dim_embed'], buckets, exchange)
            s1 = self.dp.wait_all()
        else:
            t = enc.backward(res['dfm'], res['dim_embed'])
            s1 = 1.0
        ow.t = ob.t = self.opt.t            # one global step for every variable
        # a step the decoder voided on the device (status word of its gradient buffer, all-reduced with it under data
        # parallelism) is a no-op for the CNN variables as well
        void = getattr(self.decoder.grads, 'status', None)
        ow.step(t.dw, lr, grad_scale=s1 * mult, skip=void)
        ob.step(t.dbeta, lr, grad_scale=s1 * mult, skip=void)
        enc.refresh_weights()
        enc.clear_grads_async()

    # ---- optimiser / LR (model_base.py:775-883) -----------------------------------------
    def _create_optimiser(self):
        c, share = self._config, self._share
        clip = float(getattr(c, 'clip_gradient_norm', 0) or 0)      # per-variable tf.clip_by_norm (model_base.py:394-401)
        if 'opt' not in share:
            share['opt'] = optim.make_optimiser(c.optimiser, self.decoder.params, epsilon=c.adam_epsilon,
                                                l2_decay=getattr(c, 'l2_decay', 1e-5), clip_norm=clip)
            share['legacy_lr'] = c.lr_start
        self.opt = share['opt']
        if self.head is not None and 'opt_head' not in share:
            share['opt_head'] = optim.make_optimiser(c.optimiser, self.head.params, epsilon=c.adam_epsilon,
                                                     l2_decay=getattr(c, 'l2_decay', 1e-5), clip_norm=clip)
        if self.cnn_trainable and 'opt_cnn' not in share:
            # gradient_multipliers scale the whole CNN gradient, L2 term included (model_base.py:388-401)
            enc = self._encoder_for(self._batch_size)
            mult = float(getattr(c, 'cnn_grad_multiplier', 1.0))
            l2 = getattr(c, 'l2_decay', 1e-5) * mult
            share['opt_cnn'] = (optim.make_optimiser(c.optimiser, enc.w_master, epsilon=c.adam_epsilon, l2_decay=l2,
                                                     clip_norm=clip),
                                optim.make_optimiser(c.optimiser, enc.beta, epsilon=c.adam_epsilon, l2_decay=l2,
                                                     clip_norm=clip), mult)

    @property
    def global_step(self):
        return self._share['opt'].t if 'opt' in self._share else 0

    @property
    def lr(self):
        c = self._config
        if getattr(c, 'legacy', False):
            return self._share['legacy_lr']
        return optim.cosine_lr(self.global_step, c.max_step, c.lr_start, c.lr_end)

    def update_lr(self, lr_value):
        self._share['legacy_lr'] = lr_value

    # ---- restore (model_base.py:422-490) ------------------------------------------------
    def restore_model(self, lr=None):
        c = self._config
        if not c.checkpoint_path:
            print('INFO: Training entire model from scratch.')
            return (self.lr if lr is None else lr) if self.is_training() else None
        path = c.checkpoint_path
        if os.path.isdir(path):
            path = ckpt.latest_checkpoint(path, 'model') or ckpt.latest_checkpoint(path, 'model_compact')
        elif not os.path.isfile(path) and os.path.isfile(path + '.npz'):
            path = path + '.npz'
        if path is None or not (os.path.isfile(path) or os.path.isfile(path + '.index')):
            raise ValueError('checkpoint not found: %s' % c.checkpoint_path)
        cnn_names = list(self.plan.param_shapes())
        head_names = list(encoder_head.TF_NAMES.values()) if self.head is not None else None
        cnn, dec, extra, head = ckpt.restore(path, cnn_names, self.spec, getattr(c, 'resume_training', False),
                                             getattr(c, 'checkpoint_exclude_scopes', ''), head_names=head_names)
        if head:
            self.head.load_named(head)
        if cnn:
            self._share['cnn_params'].update(cnn)
            for enc in list(self._share['encoders'].values())[:1]:
                enc.load_params(self._share['cnn_params'])      # in place: shared by every encoder / optimiser
        if dec is not None:
            cur = self.decoder.params.to_numpy()
            cur.update(dec)
            self.decoder.params.load(cur)
            print('INFO: Restored `Model` from checkpoint: {}'.format(path))
        else:
            print('INFO: Restored CNN model from checkpoint {}'.format(path))
        if extra and 'opt' in self._share:
            o = self._share['opt']
            o.t = int(extra.get('global_step', 0))
            if 'optimise/caption/adam_m' in extra:
                o.m.flat.copy_(self.torch.from_numpy(extra['optimise/caption/adam_m'])[:o.m.numel])
                o.v.flat.copy_(self.torch.from_numpy(extra['optimise/caption/adam_v'])[:o.v.numel])
            else:
                slots = ckpt.adam_from_tf(self.spec, extra)        # TF bundle: per-variable Adam / Adam_1
                accum = ckpt.momentum_from_tf(self.spec, extra)    # ... or MomentumOptimizer's single `Momentum` slot
                if slots and not isinstance(o, optim.MomentumTF):
                    o.m.load(slots[0])
                    o.v.load(slots[1])
                elif accum and isinstance(o, optim.MomentumTF):
                    o.m.load(accum)
            if 'optimise/caption/head_adam_m' in extra and 'opt_head' in self._share:
                oh = self._share['opt_head']
                oh.m.data.copy_(self.torch.from_numpy(extra['optimise/caption/head_adam_m']))
                oh.v.data.copy_(self.torch.from_numpy(extra['optimise/caption/head_adam_v']))
            elif 'opt_head' in self._share:            # TF bundle: per-variable slots of the legacy head
                from .encoder_head import TF_NAMES
                oh = self._share['opt_head']
                key = lambda k, slot: ckpt.ADAM_SCOPE + TF_NAMES[k] + slot
                if isinstance(oh, optim.MomentumTF) and all(key(k, '/Momentum') in extra for k in TF_NAMES):
                    oh.m.load({k: extra[key(k, '/Momentum')] for k in TF_NAMES})
                elif not isinstance(oh, optim.MomentumTF) and all(key(k, '/Adam') in extra and key(k, '/Adam_1') in extra
                                                                   for k in TF_NAMES):
                    oh.m.load({k: extra[key(k, '/Adam')] for k in TF_NAMES})
                    oh.v.load({k: extra[key(k, '/Adam_1')] for k in TF_NAMES})
            if 'optimise/caption/cnn_w_adam_m' in extra and 'opt_cnn' in self._share:
                ow, ob, _ = self._share['opt_cnn']
                ow.m.data.copy_(self.torch.from_numpy(extra['optimise/caption/cnn_w_adam_m']))
                ow.v.data.copy_(self.torch.from_numpy(extra['optimise/caption/cnn_w_adam_v']))
                ob.m.data.copy_(self.torch.from_numpy(extra['optimise/caption/cnn_b_adam_m']))
                ob.v.data.copy_(self.torch.from_numpy(extra['optimise/caption/cnn_b_adam_v']))
            elif 'opt_cnn' in self._share:             # TF bundle: per-variable slots of the CNN variables
                ow, ob, _ = self._share['opt_cnn']
                enc = next(iter(self._share['encoders'].values()))
                sc = ckpt.ADAM_SCOPE + 'Model/encoder/cnn/'
                if isinstance(ow, optim.MomentumTF):
                    enc.import_slots(ow.m, ob.m, extra, 'Momentum', sc)
                elif enc.import_slots(ow.m, ob.m, extra, 'Adam', sc):
                    enc.import_slots(ow.v, ob.v, extra, 'Adam_1', sc)
            print('INFO: Resume training from checkpoint: {}'.format(path))
        if not self.is_training():
            return None
        if lr is not None and getattr(c, 'legacy', False):
            self.update_lr(lr)
        return self.lr

    def voided_steps(self):
        """Training steps the device voided so far (Decoder.voided_steps; synchronises)."""
        return self.decoder.voided_steps()

    def sync_parameters(self):
        """Data parallel: broadcast every variable and optimiser slot from rank 0 (parameters initialised or restored
        per rank would otherwise differ wherever a checkpoint does not cover them, and the replicas would never agree).
        The reference is single-GPU; this is part of the DP extension."""
        dp = self.dp
        if dp.world <= 1 or dp.dist is None:
            return
        bufs = [self.decoder.params.data]
        if self.head is not None:
            bufs.append(self.head.params.data)
        for enc in list(self._share['encoders'].values())[:1]:
            bufs += [enc.w_master.data, enc.beta.data, enc.mean.data, enc.scale.data, enc.shift.data]
        for key in ('opt', 'opt_head'):
            if key in self._share:
                bufs += [self._share[key].m.data, self._share[key].v.data]
        if 'opt_cnn' in self._share:
            for o in self._share['opt_cnn'][:2]:
                bufs += [o.m.data, o.v.data]
        for b in bufs:
            dp.dist.broadcast(b, 0)
        step = self.torch.tensor([self.global_step], dtype=self.torch.int64, device=self.device)
        dp.dist.broadcast(step, 0)
        for key in ('opt', 'opt_head'):
            if key in self._share:
                self._share[key].t = int(step.item())
        for enc in list(self._share['encoders'].values())[:1]:
            enc.refresh_weights()

    def save(self, save_path, compact=True, max_to_keep=None):
        extra = {}
        fmt = getattr(self._config, 'checkpoint_format', 'npz')
        if not compact and 'opt' in self._share:
            o = self._share['opt']
            if fmt == 'tf' and isinstance(o, optim.MomentumTF):     # `--optimiser sgd`: <var>/Momentum, no beta powers
                extra = ckpt.momentum_to_tf(self.spec, o.m.to_numpy())
            elif fmt == 'tf':
                extra = ckpt.adam_to_tf(self.spec, o.m.to_numpy(), o.v.to_numpy(), o.t, o.beta1, o.beta2)
            else:
                extra = {'optimise/caption/adam_m': o.m.flat.cpu().numpy(),
                         'optimise/caption/adam_v': o.v.flat.cpu().numpy()}
        if self.head is not None:
            extra.update(self.head.export_params())
            if not compact and 'opt_head' in self._share:
                oh = self._share['opt_head']
                if fmt == 'tf':         # per-variable slots under the optimiser's scope, like the decoder's
                    from .encoder_head import TF_NAMES
                    hm = oh.m.to_numpy()
                    if isinstance(oh, optim.MomentumTF):
                        extra.update({ckpt.ADAM_SCOPE + TF_NAMES[k] + '/Momentum': hm[k] for k in TF_NAMES})
                    else:
                        hv = oh.v.to_numpy()
                        extra.update({ckpt.ADAM_SCOPE + TF_NAMES[k] + '/Adam': hm[k] for k in TF_NAMES})
                        extra.update({ckpt.ADAM_SCOPE + TF_NAMES[k] + '/Adam_1': hv[k] for k in TF_NAMES})
                else:
                    extra.update({'optimise/caption/head_adam_m': oh.m.data.cpu().numpy(),
                                  'optimise/caption/head_adam_v': oh.v.data.cpu().numpy()})
        if 'opt_cnn' in self._share:      # fine-tuned CNN variables back into the checkpoint layout
            self._share['cnn_params'].update(next(iter(self._share['encoders'].values())).export_params())
            if not compact:
                ow, ob, _ = self._share['opt_cnn']
                if fmt == 'tf':       # per-variable slots under the optimiser's scope: <scope>/Model/encoder/cnn/<var>/Adam[_1]
                    enc = next(iter(self._share['encoders'].values()))
                    sc = ckpt.ADAM_SCOPE + 'Model/encoder/cnn/'
                    if isinstance(ow, optim.MomentumTF):
                        extra.update({sc + k: v for k, v in enc.export_slots(ow.m, ob.m, 'Momentum').items()})
                    else:
                        extra.update({sc + k: v for k, v in enc.export_slots(ow.m, ob.m, 'Adam').items()})
                        extra.update({sc + k: v for k, v in enc.export_slots(ow.v, ob.v, 'Adam_1').items()})
                else:
                    extra.update({'optimise/caption/cnn_w_adam_m': ow.m.data.cpu().numpy(),
                                  'optimise/caption/cnn_w_adam_v': ow.v.data.cpu().numpy(),
                                  'optimise/caption/cnn_b_adam_m': ob.m.data.cpu().numpy(),
                                  'optimise/caption/cnn_b_adam_v': ob.v.data.cpu().numpy()})
        return ckpt.save(save_path, self.global_step, self._share['cnn_params'], self.spec,
                         self.decoder.params.to_numpy(), extra, max_to_keep, fmt=fmt)

    # ---- decode (model_base.py:692-757, :272-314) ----------------------------------------
    def _decode(self, images, beam_size, max_length, top_beam=True, want_attention=True, length_penalty_weight=0.0):
        im_embed, fm = self._encode(images)
        return self._decode_features(im_embed, fm, beam_size, max_length, top_beam, want_attention, length_penalty_weight)

    def _decode_features(self, im_embed, fm, beam_size, max_length, top_beam=True, want_attention=True,
                         length_penalty_weight=0.0):
        c = self._config
        iters = self.decoder.max_iterations(max_length, len(c.wtoi))
        if beam_size > 1:
            if not want_attention and not length_penalty_weight:
                # captions alone: the ids come back through pinned memory behind one event (no stream drain per array)
                r = {'predicted_ids': self.decoder.beam_search_ids(fm, im_embed, beam_size, iters)()}
            else:
                r = self.decoder.beam_search(fm, im_embed, beam_size, iters, want_attention=want_attention,
                                             length_penalty_weight=length_penalty_weight)
            pred = r['predicted_ids']                                  # (T, B, W)
            T = pred.shape[0]
            attn = None
            if want_attention:
                hist = r['attn_hist'].reshape(T, -1, beam_size, self.spec.H, self.spec.M)[:, :, 0]
                attn = hist.transpose(1, 2, 0, 3)                      # (B, H, T, M) of beam 0
            if top_beam:
                return pred[:, :, 0].T.copy(), attn
            return pred.transpose(2, 1, 0).copy(), attn               # (W, B, T)
        if beam_size == 0:
            # _decoder_rnn_scst(0): the sampled rollout (model_base.py:716-726 sample=True -> SampleEmbeddingHelper,
            # ops_rnn.py:158-166; commented out of the reference's SCST graph, model.py:127-129, kept callable here)
            n = self._share['sample_draws'] = self._share.get('sample_draws', 0) + 1
            ids, amap, _ = self.decoder.sample(fm, im_embed, iters, seed=int(getattr(c, 'rand_seed', 0)) + n)
            return ids, amap.cpu().numpy()
        ids, amap, _ = self.decoder.greedy(fm, im_embed, iters)
        return ids, amap.cpu().numpy()


class CaptionModel(ModelBase):
    def __init__(self, config, mode, batch_ops=None, reuse=False, name=None, device='cuda:0', dp=None):
        assert mode in ['train', 'eval', 'infer']
        print('INFO: Building graph for: {}'.format(name))
        super(CaptionModel, self).__init__(config)
        self.mode, self.batch_ops, self.reuse, self.name = mode, batch_ops, reuse, name
        c = self._config
        bs = {'train': getattr(c, 'batch_size_train', 32), 'eval': getattr(c, 'batch_size_eval', 61),
              'infer': getattr(c, 'batch_size_infer', 25)}[mode]
        self._batch_size = bs
        self._build(bs, device, dp)
        if self.is_training():
            self._create_optimiser()
        self.dec_log_ppl = None
        print('INFO: Model `{}` initialisation complete.'.format(mode))

    # ---- frozen-CNN pipelining (decoder mode): trainer.EncoderPipeline behind the reference API -------------
    def _pipelined(self, batch):
        c = self._config
        return (batch is None and not self.cnn_trainable and getattr(c, 'pipeline_encoder', True)
                and str(self.device).startswith('cuda'))

    def _submit_group(self):
        """Pull the batches of the next `encoder_group` steps from the input pipeline and start their ONE encoder
        forward on the side stream; the captions wait in a queue for their steps."""
        torch, pipe = self.torch, self._pipe
        ahead = self.__dict__.setdefault('_ahead', [])
        batches = [ahead.pop(0) if ahead else next(self.batch_ops) for _ in range(pipe.group)]
        imgs = [b[0] if torch.is_tensor(b[0]) else torch.from_numpy(np.ascontiguousarray(b[0], np.float32)).to(self.device)
                for b in batches]
        pipe.submit(imgs[0] if len(imgs) == 1 else torch.cat(imgs))
        self._cap_queue.extend(b[1] for b in batches)

    def _next_features(self):
        """-> (im_embed, fm, captions, consumed) of the next training step."""
        if getattr(self, '_pipe', None) is None:
            from .trainer import EncoderPipeline
            import collections
            group = int(getattr(self._config, 'encoder_group', 1) or 0)
            if group <= 0:                 # --encoder_group 0: auto
                # a bf16x3 plan holds three channel regions per bf16 buffer: a third of the images reaches the same buffer sizes
                # (and keeps the largest one, 109x109x192x3 at 224, under the kernels' 2^31-element limit)
                x3 = str(getattr(self._config, 'cnn_dtype', 'bf16')) == 'bf16x3'
                side = max(getattr(self._config, 'cnn_input_size', None) or [224, 224])
                per_fwd = (1280 // 3 if x3 else 1280) * 224 * 224 // max(224 * 224, side * side)
                group = auto_encoder_group(self._batch_size, images_per_forward=max(per_fwd, self._batch_size))
                group = max(1, min(group, int(getattr(self._config, 'max_step', group)) - int(self.global_step)))
            enc = self._encoder_for(self._batch_size * group)
            enc.polite_lds_kb = int(getattr(self._config, 'encoder_polite_lds_kb', 84))   # see CaptionTrainer.enable_overlap
            self._pipe = EncoderPipeline(enc, self._batch_size, group, self.device)
            self._cap_queue = collections.deque()
            self._submit_group()
        im_embed, fm, release = self._pipe.take()
        # one batch of the NEXT group per step: the input pipeline keeps working under the steps of this group instead of
        # delivering group-size batches in a burst at the group's end (a prefetch queue of 4 against groups of 20 left the
        # device idle for the 16 batches the loader still had to make: 3.3 ms per step from files, 1.7 ms of work)
        ahead = self.__dict__.setdefault('_ahead', [])
        if self._pipe.group > 1 and len(ahead) < self._pipe.group and not getattr(self, '_ahead_end', False):
            try:
                ahead.append(next(self.batch_ops))
            except StopIteration:            # a finite iterator: the group that needs the missing batch reports it
                self._ahead_end = True

        def consumed():
            if release():
                self._submit_group()
        return im_embed, fm, self._cap_queue.popleft(), consumed

    def run_train_step(self, batch=None):
        """== sess.run(m_train.dec_log_ppl): one XE update (train_fn.py:120-121).  With a frozen CNN and batches
        drawn from the input pipeline the encoder forward of the NEXT step(s) runs on a second stream under this
        step's decoder (config.pipeline_encoder, default on; config.encoder_group steps per forward, default 1)."""
        consumed = None
        if self._pipelined(batch):
            im_embed, fm, captions, consumed = self._next_features()
            im_embed = self._embed(im_embed)
        else:
            images, captions = batch if batch is not None else next(self.batch_ops)
            im_embed, fm = self._encode(images)
        cap = np.asarray(captions)
        lr = self.lr
        ft = self.cnn_trainable
        want_in = ft or self.head is not None
        res = self.decoder.train_step(fm, im_embed, cap, training=True, dp=self.dp, use_graph=not want_in,
                                      want_input_grads=want_in, on_inputs_consumed=consumed,
                                      copy_inputs=consumed is None)
        self._dec_reduced = False
        if ft:
            self._cnn_update(res, lr)
        if not ft and self.head is None:   # decoder mode: the flat gradient in chunks, each updated behind its own all-reduce
            self.dp.exchange_and_step(self.opt, self.decoder.grads, lr)
            self.dec_log_ppl = res['loss']
            self.last = res
            return res['loss']
        scale = 1.0 / self.dp.world if self._dec_reduced else self.dp.average_(self.decoder.grads.data)
        if self.head is not None:          # legacy head: its variables train with the decoder (model_base.py:834-849)
            hg = self.head.backward(res['dim_embed'])
            self.dp.average_(hg.data)
            oh = self._share['opt_head']
            oh.t = self.opt.t
            oh.step(hg, lr, grad_scale=scale)
        self.opt.step(self.decoder.grads, lr, grad_scale=scale)
        self.dec_log_ppl = res['loss']
        self.last = res
        return res['loss']

    def run_eval_step(self, batch=None):
        """== sess.run(m_valid.dec_log_ppl) (train_fn.py:328-330): loss only, dropout off.
        The fused step also produces gradients; they are simply not applied."""
        images, captions = batch if batch is not None else next(self.batch_ops)
        im_embed, fm = self._encode(images)
        res = self.decoder.train_step(fm, im_embed, np.asarray(captions), training=False, use_graph=True)
        return res['loss']

    def infer(self, batch=None):
        """== sess.run(m_infer.infer_output) -> [dec_preds (B,T), attention_maps (B,H,T,M)].
        Batches drawn from the input pipeline are decoded with the encoder forward of the NEXT group of batches already
        running on a second stream (the decode steps are small launches that leave most of the GPU idle;
        config.pipeline_encoder, default on; config.pipeline_encoder_group batches per forward, 0 = auto).  Same
        arithmetic per image; a forward over more images may pick other conv tiles, i.e. another fp32 summation order."""
        c = self._config
        if batch is not None:
            images = batch[0] if isinstance(batch, (tuple, list)) else batch
            im_embed, fm = self._encode(images)
        else:
            feats = self._next_infer_features()
            if feats is None:
                raise StopIteration('infer(): the input pipeline is exhausted')
            im_embed, fm = feats
        ids, attn = self._decode_features(im_embed, fm, c.infer_beam_size, c.infer_max_length, top_beam=True,
                                          length_penalty_weight=getattr(c, 'infer_length_penalty_weight', 0.0))
        self.infer_output = [ids, attn]
        return self.infer_output

    def infer_pipelined(self, want_attention=True):
        """Generator over the batches of the input pipeline -> [dec_preds (B,T), attention_maps or None] in input order, with
        the decode loops of THREE batches in flight (COMIC_INFER_IN_FLIGHT; one stream and one buffer set of the decoder each:
        Decoder.beam_search_ids(slot=)).
        A beam-search step is five dependent launches of 50-230 workgroups that leave most of the chip idle between them; the
        kernels of a second, independent batch fill those holes: 2.29 -> 1.64 ms per batch of 50 at beam 3 on the word
        baseline (tools/beam_time.py TWO=1), the same ids; three in flight +6 % over two (24.5k against 23.1k captions/s on one
        box; four and five lose: 23.9k, 19.3k -- each loop streams the 52 MB vocabulary projection per step).  Only the captions-only beam search runs this way (no attention
        maps, no length penalty: what `infer.py` writes unless --save_attention_maps); everything else yields infer()."""
        c, torch = self._config, self.torch
        lp = getattr(c, 'infer_length_penalty_weight', 0.0)
        NL = max(1, min(5, int(os.environ.get('COMIC_INFER_IN_FLIGHT', '3'))))      # decode loops in flight (3: measured best of 1-5)
        if want_attention or lp or c.infer_beam_size <= 1 or not str(self.device).startswith('cuda'):
            NL = 1
        if NL == 1:
            while True:
                feats = self._next_infer_features()
                if feats is None:
                    return
                ids, attn = self._decode_features(feats[0], feats[1], c.infer_beam_size, c.infer_max_length, top_beam=True,
                                                  want_attention=want_attention, length_penalty_weight=lp)
                yield [ids, attn]
        iters = self.decoder.max_iterations(c.infer_max_length, len(c.wtoi))
        lanes = [streams.lane(torch, self.device, 'infer%d' % k) for k in range(NL)]
        pending = [None] * NL
        n = 0
        while True:
            feats = self._next_infer_features()
            k = n % NL
            if pending[k] is not None:            # the batch decoded on this lane NL batches ago: next in input order
                yield [pending[k]()[:, :, 0].T.copy(), None]
                pending[k] = None
            if feats is None:
                break
            im_embed, fm = feats
            lanes[k].wait_stream(torch.cuda.current_stream())
            with torch.cuda.stream(lanes[k]):
                pending[k] = self.decoder.beam_search_ids(fm, im_embed, c.infer_beam_size, iters, slot=k)
            im_embed.record_stream(lanes[k])
            fm.record_stream(lanes[k])
            n += 1
        for j in range(NL):                       # the lanes still hold the last batches, oldest first
            k = (n + j) % NL
            if pending[k] is not None:
                yield [pending[k]()[:, :, 0].T.copy(), None]
                pending[k] = None

    def _next_infer_features(self):
        """(im_embed, fm) of the next batch of the input pipeline, None at its end.  ONE encoder forward covers the next G
        batches (config.pipeline_encoder_group, 0 = auto: about 200 images -- the forward at 200 images runs at 1.6x the MFMA
        rate of 50) and runs on a second stream under the decode steps of the group in flight; batches that do not fill a
        group (the end of the input, a ragged last batch) are encoded one by one, in input order."""
        c, torch = self._config, self.torch

        def next_images():
            try:
                b = next(self.batch_ops)
            except StopIteration:
                return None
            im = b[0] if isinstance(b, (tuple, list)) else b
            return im if torch.is_tensor(im) else torch.from_numpy(np.ascontiguousarray(im, np.float32)).to(self.device)

        if not (getattr(c, 'pipeline_encoder', True) and str(self.device).startswith('cuda')
                and os.environ.get('COMIC_PIPELINE_INFER', '1') == '1'):
            images = next_images()
            return None if images is None else self._encode_copy(images)

        def next_group(G, B):
            """up to G batches of B images in input order; stops at the end of the input or behind a ragged batch"""
            out = []
            while len(out) < G:
                im = next_images()
                if im is None:
                    break
                out.append(im)
                if int(im.shape[0]) != B:
                    break
            return out

        tail = self.__dict__.setdefault('_infer_tail', [])
        pipe = getattr(self, '_ipipe', None)
        if pipe is None or pipe.steps_ready == 0:
            if tail:
                return self._encode_copy(tail.pop(0))
            first = next_images() if pipe is None else None
            if pipe is None:
                if first is None:
                    return None
                B = int(first.shape[0])
                G = int(getattr(c, 'pipeline_encoder_group', 0)) or max(1, min(8, 200 // max(1, B)))
                grp = [first] + next_group(G - 1, B)
            else:
                B, G = pipe.batch, pipe.group
                grp = next_group(G, B)
            if not grp:
                return None
            if len(grp) < G or any(int(g.shape[0]) != B for g in grp):
                tail.extend(grp)
                return self._encode_copy(tail.pop(0))
            if pipe is None:
                from .trainer import EncoderPipeline
                pipe = self._ipipe = EncoderPipeline(self._encoder_for(G * B), B, G, self.device)
            pipe.submit(torch.cat(grp, 0) if G > 1 else grp[0])
        im_embed, fm, release = pipe.take()
        im_embed, fm = self._embed(im_embed).clone(), fm.clone()     # the staging copy goes back to the pipeline at once
        if release():                                                 # first batch of its group: the next group's forward
            grp = next_group(pipe.group, pipe.batch)                  # runs under this group's decode steps
            if len(grp) == pipe.group and all(int(g.shape[0]) == pipe.batch for g in grp):
                pipe.submit(torch.cat(grp, 0) if pipe.group > 1 else grp[0])
            else:
                tail.extend(grp)                                      # served once the group in flight is consumed
        return im_embed, fm


def auto_encoder_group(batch_size, images_per_forward=1280, cap=64):
    """Steps per encoder forward when `--encoder_group 0` (auto): enough images per forward to fill the conv tiles
    (the InceptionV3 forward runs at twice the MFMA rate at 1280 images than at 64 -- DESIGN.md section 5), capped so
    that the activation buffers of one forward stay around 16 GB.  bench.py's pick_encoder_group applies the same size
    to its timed step count."""
    return max(1, min(cap, images_per_forward // max(1, int(batch_size))))


class CaptionModel_SCST(ModelBase):
    def __init__(self, config, scst_mode, reuse=False, device='cuda:0', dp=None):
        assert scst_mode in ['train', 'sample']
        print('INFO: Building graph for: {}'.format(scst_mode))
        super(CaptionModel_SCST, self).__init__(config)
        self.mode = scst_mode if scst_mode == 'train' else 'infer'
        self.reuse, self.name = reuse, scst_mode
        c = self._config
        bs = c.batch_size_train            # the encoder sees the untiled batch in both SCST graphs (run_train_scst tiles its outputs)
        self._batch_size = bs
        self._build(bs, device, dp)
        if self.is_training():
            self._create_optimiser()
        print('INFO: Model `{}` initialisation complete.'.format(scst_mode))

    def sample(self, imgs, defer_greedy=False):
        """-> (dec_preds_beam (beam,B,T), dec_preds_greedy (B,T)); beam search with
        infer_max_length=20, length penalty 0 (model_base.py:208-215).
        defer_greedy: the beam rollouts are enqueued first and fetched without draining the stream, the greedy rollout is
        enqueued behind them and the second return value is a function that fetches it: the caller turns the beam ids into
        text meanwhile (same parameters, same features: the order of the two rollouts changes nothing)."""
        c = self._config
        # (the caches below hold the image object itself: `is` on a live object cannot be fooled by a recycled id())
        pf = self._share.pop('scst_prefetch', None)
        grp = self._share.get('scst_group') or []
        hit = next((k for k, e in enumerate(grp) if e[0] is imgs), None)
        if hit is not None:
            _, im_embed, fm = grp.pop(hit)         # one of the batches of a grouped forward (prefetch_group)
        elif pf is not None and pf[0] is imgs:
            im_embed, fm = pf[1], pf[2]            # forward enqueued during the previous step's reward computation
        else:
            im_embed, fm = self._encode(imgs)      # ONE encoder forward serves both rollouts ...
        # ... and the training step on the same images that follows (train_fn_scst: the CNN is frozen in SCST mode, so
        # run_train_scst takes these features instead of a second forward)
        self._share['scst_features'] = (imgs, im_embed.clone(), fm.clone())
        if defer_greedy and c.scst_beam_size > 1:
            iters = self.decoder.max_iterations(20, len(c.wtoi))
            fetch_beam = self.decoder.beam_search_ids(fm, im_embed, c.scst_beam_size, iters)
            fetch_greedy = self.decoder.greedy(fm, im_embed, iters, defer=True)
            beam = fetch_beam().transpose(2, 1, 0).copy()                 # (W, B, T)
            return beam, (lambda: fetch_greedy()[0])
        greedy, _ = self._decode_features(im_embed, fm, 1, 20, want_attention=False)
        beam, _ = self._decode_features(im_embed, fm, c.scst_beam_size, 20, top_beam=False, want_attention=False)
        return beam, ((lambda: greedy) if defer_greedy else greedy)

    def prefetch_features(self, imgs):
        """Enqueue the encoder forward of the NEXT step's images now: the device is idle while the host scores this
        step's rollouts (text, CIDEr-D / BLEU, ids), and the frozen CNN does not depend on the update in between.
        sample(imgs) on the same object picks the features up."""
        if 'opt_cnn' in self._share:
            return
        im_embed, fm = self._encode(imgs)
        self._share['scst_prefetch'] = (imgs, im_embed.clone(), fm.clone())

    def prefetch_group(self, batches):
        """The frozen CNN of SCST mode does not depend on the updates in between: ONE encoder forward for the images of the
        next len(batches) steps (config.encoder_group; a batch-32 forward runs at a fifth of the per-image speed of a
        batch-256 one), enqueued while the host scores the current step.  sample(imgs) on the same image objects picks its
        rows up.  Falls back to one forward per batch when the batches differ in shape or the CNN is trainable."""
        if 'opt_cnn' in self._share or not batches:
            return
        torch = self.torch
        imgs = [b if torch.is_tensor(b) else torch.from_numpy(np.ascontiguousarray(b, np.float32)).to(self.device) for b in batches]
        if len({tuple(t.shape) for t in imgs}) != 1 or len(imgs) == 1:
            for src, t in zip(batches, imgs):
                im_embed, fm = self._encode(t)
                self._share.setdefault('scst_group', []).append((src, im_embed.clone(), fm.clone()))
            return
        B = int(imgs[0].shape[0])
        im_embed, fm = self._encode(torch.cat(imgs, 0))
        im_embed, fm = im_embed.clone(), fm.clone()
        for k, src in enumerate(batches):
            self._share.setdefault('scst_group', []).append((src, im_embed[k * B:(k + 1) * B], fm[k * B:(k + 1) * B]))

    def run_train_scst(self, imgs, captions, rewards, tile=1):
        """One reward-weighted update on `tile` hypotheses per image.  imgs: the batch tiled `tile`
        times (tile=1, the reference's feed) or the untiled batch with tile=beam: the frozen encoder
        then runs once and (im_embed, fm) are tiled -- same values, 1/tile of the CNN work."""
        kept = self._share.pop('scst_features', None)
        if kept is not None and kept[0] is imgs and 'opt_cnn' not in self._share:
            im_embed, fm = kept[1], kept[2]          # the sampling pass's features of these very images
        else:
            im_embed, fm = self._encode(imgs)
        if tile > 1:
            im_embed, fm = im_embed.repeat(tile, 1), fm.repeat(tile, 1, 1)
        if rewards is None:           # begin_train_scst: the forward is enqueued, the rewards come with finish_train_scst
            self.decoder.train_step(fm, im_embed, np.asarray(captions), training=True, use_graph=True, phase='fwd')
            self._share['scst_open'] = np.asarray(captions)
            return None
        lr = self.lr
        res = self.decoder.train_step(fm, im_embed, np.asarray(captions), rewards=np.asarray(rewards, np.float32),
                                      training=True, use_graph=True)
        self.dp.exchange_and_step(self.opt, self.decoder.grads, lr)
        return res['loss']

    def begin_train_scst(self, imgs, captions, tile=1):
        """The part of run_train_scst that needs no reward -- encoder features, teacher-forced forward to the logits -- enqueued
        on the device; the caller computes the rewards meanwhile and hands them to finish_train_scst."""
        return self.run_train_scst(imgs, captions, None, tile=tile)

    def finish_train_scst(self, rewards):
        """Loss, backward and update of the step begin_train_scst started (same kernels in the same order as run_train_scst)."""
        captions = self._share.pop('scst_open')
        lr = self.lr
        res = self.decoder.train_step(None, None, captions, rewards=np.asarray(rewards, np.float32), training=True,
                                      use_graph=True, phase='bwd')
        self.dp.exchange_and_step(self.opt, self.decoder.grads, lr)
        return res['loss']
