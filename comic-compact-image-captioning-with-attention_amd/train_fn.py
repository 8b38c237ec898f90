"""Training drivers with the reference's entry points (src/train_fn.py): `train_fn`,
`train_fn_scst`, `_run_eval_loop`, `_lr_reduce_check`, `try_to_train` -- same step/epoch
accounting, save cadence, log lines and error-file behaviour; the per-step body runs on the
native executors instead of `sess.run`."""
from __future__ import annotations

import os
import sys
import time
import traceback as tb

import numpy as np

from . import configuration as conf
from . import inputs, model as mdl
from .ops import id_to_caption, radix_ids_to_captions_and_ids
from .scst.scorers import captionScorer

pjoin = os.path.join


def _manager(config):
    if config.token_type == 'radix':
        return inputs.InputManager_Radix(config)
    if config.token_type == 'char':
        return inputs.InputManager_Char(config)
    return inputs.InputManager(config)


def _model_size_report(m, log_path):
    n = m.decoder.params.n_params()
    msg = 'INFO: Scope `Model/decoder/rnn_decoder` contains {:,d} trainable parameters.'.format(n)
    print('\n{}\n'.format(msg))
    with open(pjoin(log_path, 'model_size.txt'), 'a') as f:
        f.write('\r\n{}\r\n\r\n'.format(msg))
        for k, shp in m.spec.param_shapes().items():
            f.write('{}\r\n{}\r\n\r\n'.format(k, list(shp)))
    return n


def train_fn(config, device='cuda:0', dp=None):
    """Main training function. To be called by `try_to_train()`."""
    print('INFO: Logging to `{}`.'.format(config.log_path))
    mdl.reset_default_graph()
    inputs_man = _manager(config)
    try:
        return _train_loop(inputs_man, device, dp)
    finally:
        inputs_man.close()                             # stops the prefetch threads of this stage


def _train_loop(inputs_man, device, dp):
    inputs_man.enable_device_preprocess(device)        # host: JPEG decode only
    c = inputs_man.config
    num_batches = int(c.split_sizes['train'] / (c.batch_size_train * max(1, int(getattr(c, 'dp_world', 1) or 1))))   # global batches
    lr = c.lr_start
    n_steps_log = int(num_batches / c.num_logs_per_epoch)
    m_train = mdl.CaptionModel(c, mode='train', batch_ops=inputs_man.batch_train, reuse=False, name='train',
                               device=device, dp=dp)
    m_train.dset_size = c.split_sizes['train']
    m_valid = None
    if inputs_man.batch_eval is not None:
        m_valid = mdl.CaptionModel(c, mode='eval', batch_ops=inputs_man.batch_eval, reuse=True, name='valid',
                                   device=device, dp=dp)
        m_valid.dset_size = c.split_sizes['valid']
    lr = m_train.restore_model(lr)
    m_train.sync_parameters()                        # data parallel: every rank starts from rank 0's variables
    _model_size_report(m_train, c.log_path)
    start_step = m_train.global_step
    n_steps_log = max(1, int(n_steps_log / 5))
    print('INFO: Graph constructed. Training begins now.')
    start_epoch = time.time()
    for step in range(start_step, c.max_step):
        epoch = int(step / num_batches) + 1
        ppl = m_train.run_train_step()
        global_step = m_train.global_step
        if (step + 1) % (n_steps_log * 5) == 0:
            _check_loss(ppl, global_step, m_train, dp)
            t = time.time() - start_epoch
            speed = (step + 1 - start_step) * c.batch_size_train / t
            print('   Training speed: {:7.2f} examples/sec.'.format(speed))
        elif (step + 1) % n_steps_log == 0:
            _check_loss(ppl, global_step, m_train, dp)
            logstr = 'Epoch {:2d} ~~ {:6.2f} %  ~  '.format(epoch, ((step % num_batches) + 1) / num_batches * 100)
            logstr += 'Perplexity {:8.4f} ~ LR {:5.3e} ~ '.format(float(np.exp(float(ppl))), m_train.lr)
            logstr += 'Step {}'.format(global_step)
            print('   ' + logstr)
        if num_batches > 5000:
            save = (step + 1) % int(num_batches / 2) == 0
        else:
            save = (step + 1) % num_batches == 0
        save = save and (step + 100) < c.max_step
        if save or (step + 1) == c.max_step:
            if dp is None or dp.rank == 0:
                m_train.save(c.save_path + '_compact', compact=True, max_to_keep=c.max_saves)
                m_train.save(c.save_path, compact=False, max_to_keep=2)
            if m_valid is not None:
                _run_eval_loop(c, m_valid, global_step)
        if (step + 1) % num_batches == 0:
            if getattr(c, 'legacy', False):
                lr = _lr_reduce_check(c, epoch, lr)
                m_train.update_lr(lr)
            t = time.time() - start_epoch
            print('\n\n>>> Epoch {:3d} complete'.format(epoch))
            print('>>> Time taken: {:10.2f} minutes\n\n'.format(t / 60))
            start_epoch = time.time()
            start_step = step + 1
    print('\n\nINFO: Training completed.')


def train_fn_scst(config, idx_ngram=False, device='cuda:0', dp=None):
    """SCST training function. To be called by `try_to_train()`."""
    print('INFO: Logging to `{}`.'.format(config.log_path))
    mdl.reset_default_graph()
    inputs_man = inputs.InputManager_SCST(config)
    try:
        return _scst_loop(inputs_man, idx_ngram, device, dp)
    finally:
        inputs_man.close()


def _scst_loop(inputs_man, idx_ngram, device, dp):
    inputs_man.enable_device_preprocess(device)
    c = inputs_man.config
    num_batches = int(c.split_sizes['train'] / (c.batch_size_train * max(1, int(getattr(c, 'dp_world', 1) or 1))))   # global batches
    lr = c.lr_start
    n_steps_log = int(num_batches / c.num_logs_per_epoch)
    m_train = mdl.CaptionModel_SCST(c, scst_mode='train', reuse=False, device=device, dp=dp)
    m_sample = mdl.CaptionModel_SCST(c, scst_mode='sample', reuse=True, device=device, dp=dp)
    idf_fname = c.dataset_file_pattern.format('scst-idxs' if idx_ngram else 'scst-words') + '.p'
    idf_fp = pjoin(c.dataset_dir, 'captions', idf_fname)
    if not os.path.isfile(idf_fp):
        raise ValueError('File not found: `{}`'.format(idf_fp))
    scorer = captionScorer(path_to_cached_tokens=idf_fp,
                           metric_weights=dict(ciderD=c.scst_weight_ciderD, bleu=c.scst_weight_bleu))
    lr = m_train.restore_model(lr)
    m_train.sync_parameters()                        # data parallel: every rank starts from rank 0's variables
    _model_size_report(m_train, c.log_path)
    start_step = m_train.global_step
    n_steps_log = max(1, int(n_steps_log / 5))
    print('INFO: Graph constructed. SCST training begins now.')
    start_epoch = time.time()
    greedy_high_sc = 0
    ahead = None
    # --encoder_group G > 1 (frozen CNN): the images of the next G steps go through ONE encoder forward, enqueued while the
    # host scores the current step (CaptionModel_SCST.prefetch_group); 1 = one forward per step, prefetched one step ahead
    group = int(getattr(c, 'encoder_group', 1) or 0)
    if group <= 0:                              # --encoder_group 0 (auto): about 256 images per forward, at most 8 steps ahead
        group = mdl.auto_encoder_group(c.batch_size_train, images_per_forward=256, cap=8)
    queue = []
    for step in range(start_step, c.max_step):
        epoch = int(step / num_batches) + 1
        if group > 1:
            if not queue:                       # (the first step, or a group that could not be prefetched)
                queue = [next(inputs_man.batch_train) for _ in range(min(group, c.max_step - step))]
                m_sample.prefetch_group([b[0] for b in queue])
            imgs, refs = queue.pop(0)
        else:
            imgs, refs = ahead if ahead is not None else next(inputs_man.batch_train)
        # `cap_beam` is (beam_size, batch_size, time) -> (beam_size * batch_size, time):
        # [[im0_hypo0], ..., [imN_hypo0], [im0_hypo1], ..., [imN_hypo1]]   (train_fn.py:226-238)
        # (the greedy rollout runs on the device while the host turns the beam rollouts into text and ids)
        cap_beam, fetch_greedy = m_sample.sample(imgs, defer_greedy=True)
        cap_beam = np.reshape(cap_beam, [-1, cap_beam.shape[-1]])
        # every sampled hypothesis is trained on (get_hypo_scores returns `sample` itself), so the update's forward pass --
        # which no reward enters -- is enqueued BEFORE the host scores the rollouts and runs on the device meanwhile.
        # The reference feeds the images tiled by the beam size (train_fn.py:251-253); the CNN is frozen and
        # deterministic, so the encoder runs once and its two outputs are tiled instead
        if c.token_type == 'radix' and getattr(inputs_man, 'radix_wtoi', None) is not None:
            # ids -> text -> target ids in one pass (equal to the two calls; the host work between the rollouts and the
            # update's forward pass is what the device waits for)
            caps, hypos_idx = radix_ids_to_captions_and_ids(cap_beam, c, inputs_man.radix_wtoi)
            cap_beam = [[s] for s in caps]
        else:
            cap_beam = [[s] for s in id_to_caption(cap_beam, c)]
            hypos_idx = inputs_man.captions_to_batched_ids(cap_beam)
        # the update's forward pass goes to the device BEFORE the host looks at the greedy rollout (which is still running:
        # its ids come back through an event of their own), so the device runs rollouts and update back to back
        m_train.begin_train_scst(imgs, hypos_idx, tile=c.scst_beam_size)
        cap_greedy = [[s] for s in id_to_caption(fetch_greedy(), c)]
        # the next batch's encoder forward joins the update's forward pass on the device while the host scores
        if group > 1:
            if not queue and step + 1 < c.max_step:
                queue = [next(inputs_man.batch_train) for _ in range(min(group, c.max_step - step - 1))]
                m_sample.prefetch_group([b[0] for b in queue])
        else:
            ahead = next(inputs_man.batch_train) if step + 1 < c.max_step else None
            if ahead is not None:
                m_sample.prefetch_features(ahead[0])
        hypos, sc_sample, sc_greedy = scorer.get_hypo_scores(refs, cap_beam, cap_greedy)
        rewards = sc_sample - sc_greedy
        greedy_high_sc = max(greedy_high_sc, np.amax(sc_greedy))
        assert hypos is cap_beam and hypos_idx.shape[0] == sc_sample.shape[0]
        ppl = m_train.finish_train_scst(rewards)
        global_step = m_train.global_step
        if (step + 1) % (n_steps_log * 5) == 0:
            t = time.time() - start_epoch
            speed = (step + 1 - start_step) * c.batch_size_train / t
            logstr = '\n   Training speed: {:7.2f} examples/sec.'.format(speed)
            logstr += '\n   mean reward: \t{:8.4f}'.format(np.mean(rewards))
            logstr += '\n   greedy high score: \t{:8.4f}'.format(greedy_high_sc)
            logstr += '\n   greedy: \t\t`{}`'.format(cap_greedy[0][0])
            logstr += '\n   top beam: \t\t`{}`\n'.format(hypos[0][0])
            print(logstr)
        elif (step + 1) % n_steps_log == 0:
            _check_loss(ppl, global_step, m_train, dp)
            logstr = '   Epoch {:2d} ~~ {:6.2f} %  ~  '.format(epoch, ((step % num_batches) + 1) / num_batches * 100)
            logstr += 'Greedy score {:8.4f} ~ Loss {:8.4f} ~ LR {:5.3e} ~ Step {}'.format(
                np.mean(sc_greedy), float(ppl), m_train.lr, global_step)
            print(logstr)
        if num_batches > 5000:
            save = (step + 1) % int(num_batches / 2) == 0
        else:
            save = (step + 1) % num_batches == 0
        save = save and (step + 100) < c.max_step
        if (save or (step + 1) == c.max_step) and (dp is None or dp.rank == 0):
            m_train.save(c.save_path + '_compact', compact=True, max_to_keep=c.max_saves)
            m_train.save(c.save_path, compact=False, max_to_keep=2)
        if (step + 1) % num_batches == 0:
            t = time.time() - start_epoch
            print('\n\n>>> Epoch {:3d} complete'.format(epoch))
            print('>>> Time taken: {:10.2f} minutes\n\n'.format(t / 60))
            start_epoch = time.time()
            start_step = step + 1
    print('\n\nINFO: Training completed.')


def _lr_reduce_check(config, epoch, learning_rate):
    """ Helper to reduce learning rate every n epochs."""
    if learning_rate > config.lr_end and epoch % config.lr_reduce_every_n_epochs == 0:
        learning_rate /= 2
        if learning_rate < config.lr_end:
            learning_rate = config.lr_end
    return learning_rate


def _check_loss(loss, step, model=None, dp=None):
    """Log points are where the host looks at the device (one sync per `num_logs_per_epoch`-th of an epoch).  Two things
    stop the run here: a non-finite loss of THIS step, and a non-zero count of steps the device voided since the start --
    comic_decoder_train_step turns a step's losses into NaN, raises the gradient buffer's status word (the gated optimiser
    then skips the update on the device) and counts it in the parameter buffer's sticky word when a bounded wait of a
    persistent time loop expired (include/comic_hip.h), so a voided step BETWEEN two log points is seen too.  Under data
    parallelism the count is max-reduced, so every rank raises at the same log point instead of leaving the others in
    the next collective.  try_to_train writes the error file."""
    v = float(loss)
    voided = int(model.voided_steps()) if model is not None and hasattr(model, 'voided_steps') else 0
    if dp is not None and getattr(dp, 'world', 1) > 1:
        voided = int(dp.max_scalar(voided))
        any_bad = dp.max_scalar(0 if np.isfinite(v) else 1)       # (every rank takes part, whatever its own loss)
        v = float('nan') if any_bad else v
    if voided:
        raise RuntimeError('step {}: the device voided {} training step(s) -- a persistent decoder loop timed out (retry with '
                           'COMIC_PERSIST=0 to run the per-step kernels)'.format(step, voided))
    if not np.isfinite(v):
        raise RuntimeError('step {}: the training loss is {} (on this or another rank) -- the run diverged'.format(step, v))


def _run_eval_loop(c, m, global_step):
    """Validation loop; returns the average perplexity per word."""
    assert m.dset_size % c.batch_size_eval == 0
    num_batches = int(m.dset_size / c.batch_size_eval)
    print('\nEvaluating model...\n')
    ppl_list = [float(m.run_eval_step()) for _ in range(num_batches)]
    avg_ppl = float(np.exp(np.mean(ppl_list)))
    print('>>> {} perplexity per word: {:8.4f}\n'.format(m.name, avg_ppl))
    return avg_ppl


def _dp_barrier():
    try:
        import torch.distributed as dist
        if dist.is_available() and dist.is_initialized():
            dist.barrier()
    except ImportError:
        pass


def _dp_any_failed(failed_here):
    """True on every rank when any rank reports a failure (a collective; a no-op without torch.distributed)."""
    try:
        import torch
        import torch.distributed as dist
    except ImportError:
        return bool(failed_here)
    if not (dist.is_available() and dist.is_initialized()):
        return bool(failed_here)
    dev = 'cpu' if dist.get_backend() == 'gloo' else 'cuda'
    x = torch.tensor([1.0 if failed_here else 0.0], device=dev)
    dist.all_reduce(x, op=dist.ReduceOp.MAX)
    return bool(x.item() > 0)


def try_to_train(train_fn, try_block=True, overwrite=False, **kargs):
    """Wrapper for the main training function."""
    config = conf.Config(**kargs)
    # data parallel: rank 0 alone checks / creates the run directory, and every rank learns its verdict -- a rank 0 that
    # left here on its own (SystemExit has exit code 0: torchrun does not tear the job down) would leave the others in
    # the barrier until the collective times out
    verdict = None
    if int(kargs.get('dp_rank', 0) or 0) == 0:
        try:
            config.overwrite_safety_check(overwrite)
        except BaseException as e:                     # incl. SystemExit
            verdict = e
    if _dp_any_failed(verdict is not None):
        if verdict is not None:
            raise verdict
        raise SystemExit('rank 0 refused the run directory %s' % config.log_path)
    if config.resume_training:
        print('INFO: Resuming training from checkpoint.')
        config = conf.load_config(pjoin(config.log_path, 'config.pkl'))
        config.resume_training = True
        config.checkpoint_path = kargs.pop('log_path')
        config.lr_end = kargs.pop('lr_end')
        config.max_epoch = kargs.pop('max_epoch')
        for k in ('dp_world', 'dp_rank'):              # of THIS launch, not of the run that wrote config.pkl
            if k in kargs:
                setattr(config, k, kargs[k])
    else:
        if int(kargs.get('dp_rank', 0) or 0) == 0:     # one writer; the other ranks wait for the files
            config.save_config_to_file()
        _dp_barrier()
    if not try_block:
        return train_fn(config)
    try:
        train_fn(config)
    except KeyboardInterrupt:
        raise
    except BaseException:
        error_log = sys.exc_info()
        if not os.path.exists(config.log_path):
            os.makedirs(config.log_path)
        err_msg = 'Error occured:\r\n\r\n%s\r\n' % str(error_log[0])
        err_msg += '%s\r\n%s\r\n\r\n' % (str(error_log[1]), str(error_log[2]))
        err_msg += '\r\n\r\nTraceback stack:\r\n\r\n'
        for entry in tb.format_list(tb.extract_tb(error_log[2])):
            err_msg += '%s\r\n' % str(entry)
        name = 'error__' + os.path.split(config.log_path)[1] + '.txt'
        with open(pjoin(os.path.dirname(config.log_path), name), 'w') as f:
            f.write(err_msg)
        print('\nWARNING: An error has occurred.\n')
        print(err_msg)
        if int(getattr(config, 'dp_world', 1) or 1) > 1:
            # a rank-local failure under data parallelism: the other ranks sit in (or are about to enter) a collective.
            # Leaving with a non-zero exit code makes the launcher end the whole job instead of letting them wait for
            # the collective's timeout.
            sys.stdout.flush()
            os._exit(1)
