"""CPU restatement (numpy) of greedy and beam-search decoding.

TEST INFRASTRUCTURE -- see oracle/__init__.py.  PARITY UNPINNED: restates
tf.contrib.seq2seq (tensorflow==1.9.0, un-vendored) as used by the reference:

  rnn_decoder_search ......... common/ops_rnn.py:115-180  (GreedyEmbeddingHelper,
                               impute_finished=False: ids after EOS are NOT masked)
  rnn_decoder_beam_search .... common/ops_rnn.py:49-112   (BeamSearchDecoder,
                               reorder_tensor_arrays=True, length_penalty_weight 0)
  BeamSearchDecoderMultiHead . common/ops_rnn.py:807-845  (alpha history re-ordered with
                               gather_tree_from_array)
  start/end ids, max iters ... src/model_base.py:692-757
  tile_batch before keys ..... src/model_base.py:127-131
  post-process ............... src/model_base.py:272-314
  [TF-1.9] beam step / gather_tree semantics: SURVEY Appendix A.10.
"""
from __future__ import annotations

import numpy as np

from . import decoder_ref as dr

F32_MIN = np.finfo(np.float32).min


def max_iterations(cfg, infer_max_length, vocab_len):
    """model_base.py:708-714."""
    it = infer_max_length
    if cfg.token_type == 'radix':
        from .text_ref import number_to_base
        it *= len(number_to_base(vocab_len, cfg.radix_base))
    elif cfg.token_type == 'char':
        it *= 5
    return it


def greedy_decode(p, cfg, fm, im_embed, max_iters, gumbel=None):
    """-> ids [B,T_exec] int32, logits [B,T_exec,V], attn_maps [B,H,T_exec,M].
    gumbel [max_iters,B,V]: SampleEmbeddingHelper instead of GreedyEmbeddingHelper (ops_rnn.py:158-166; [TF-1.9]
    sample_ids = Categorical(logits).sample()): the draw is argmax(logits + Gumbel noise), the noise being the caller's."""
    B, M, _ = fm.shape
    keys, values = dr.memory_projections(p, cfg, fm)
    c, h, _ = dr.rnn_init(p, cfg, im_embed, None)
    att = np.zeros((B, cfg.attn_size), fm.dtype)
    ids = np.full(B, cfg.start_id, np.int64)
    finished = np.zeros(B, bool)
    out_ids, out_logits, out_alpha = [], [], []
    for t in range(max_iters):
        x = dr.embed(p['emb'], ids)
        y, c, h, att, alpha, _ = dr.decoder_step(p, cfg, keys, values, x, c, h, att, None)
        lg = y @ p['W_o'] + p['b_o']
        ids = (lg if gumbel is None else lg + gumbel[t]).argmax(axis=1)   # lowest index wins ties (A.8)
        out_ids.append(ids.astype(np.int32)); out_logits.append(lg); out_alpha.append(alpha)
        finished |= (ids == cfg.end_id)
        if finished.all():
            break
    return (np.stack(out_ids, 1), np.stack(out_logits, 1),
            np.stack(out_alpha, 0).transpose(1, 2, 0, 3))


def gather_tree(step_ids, parent_ids, max_sequence_lengths, end_token):
    """[TF-1.9] beam_search_ops.gather_tree (C++ op).  All inputs [T,B,W] int32."""
    T, B, W = parent_ids.shape
    beams = np.full((T, B, W), end_token, np.int32)
    for b in range(B):
        L = min(T, int(max_sequence_lengths[b]))
        if L <= 0:
            continue
        for w in range(W):
            beams[L - 1, b, w] = step_ids[L - 1, b, w]
            parent = parent_ids[L - 1, b, w]
            for level in range(L - 2, -1, -1):
                if parent < 0 or parent > W:
                    raise ValueError('Saw invalid parent id %d' % parent)
                beams[level, b, w] = step_ids[level, b, parent]
                parent = parent_ids[level, b, parent]
            fin = False
            for t in range(L):
                if fin:
                    beams[t, b, w] = end_token
                elif beams[t, b, w] == end_token:
                    fin = True
    return beams


def gather_tree_from_array(t, parent_ids, sequence_length):
    """[TF-1.9] beam_search_decoder.gather_tree_from_array.
    t: [T, B*W, S] or [T,B,W,S]; parent_ids [T,B,W]; sequence_length [B,W]."""
    T, B, W = parent_ids.shape
    beam_ids = np.tile(np.arange(W, dtype=np.int32)[None, None, :], (T, B, 1))
    mask = (np.arange(T)[None, None, :] < np.asarray(sequence_length)[:, :, None]).astype(np.int32)
    mask = mask.transpose(2, 0, 1)                                        # [T,B,W]
    masked = beam_ids * mask + (1 - mask) * (W + 1)
    max_len = np.asarray(sequence_length).max(axis=1).astype(np.int32)
    sorted_ids = gather_tree(masked, parent_ids, max_len, W + 1)
    sorted_ids = np.where(mask.astype(bool), sorted_ids, beam_ids)
    src = np.asarray(t).reshape(T, B, W, -1)
    ti = np.arange(T)[:, None, None]
    bi = np.arange(B)[None, :, None]
    return src[ti, bi, sorted_ids].reshape(np.asarray(t).shape)


def beam_search_decode(p, cfg, fm, im_embed, beam, max_iters, return_debug=False, length_penalty_weight=0.0):
    """-> predicted_ids [T,B,W] (after gather_tree), scores [T,B,W], attn history
    [T, B*W, H*M] beam-sorted, plus raw step/parent ids when return_debug.
    length_penalty_weight (ops_rnn.py:96, infer.py:65) [TF-1.9 _beam_search_step / _get_scores]: candidates are ranked by
    total / ((5 + length) / 6)^w, length = the beam's + 1 unless the beam is finished or the candidate is EOS
    (lengths_to_add = one_hot(EOS, on 0, off 1) * not finished); the beam state keeps the unpenalised totals,
    `scores` are the penalised ones."""
    B, M, C = fm.shape
    W, V, H = beam, cfg.softmax_size, cfg.attn_num_heads
    dt = fm.dtype
    # tile_batch: each row repeated W times consecutively
    fm_t = np.repeat(fm, W, axis=0)
    im_t = np.repeat(im_embed, W, axis=0)
    keys, values = dr.memory_projections(p, cfg, fm_t)
    c, h, _ = dr.rnn_init(p, cfg, im_t, None)
    att = np.zeros((B * W, cfg.attn_size), dt)
    log_probs = np.full((B, W), -np.inf, dt); log_probs[:, 0] = 0
    finished = np.ones((B, W), bool); finished[:, 0] = False
    lengths = np.zeros((B, W), np.int64)
    ids = np.full(B * W, cfg.start_id, np.int64)
    step_ids, parent_ids, scores_all, alphas = [], [], [], []
    bidx = np.arange(B)[:, None]
    for t in range(max_iters):
        x = dr.embed(p['emb'], ids)
        y, c, h, att, alpha, _ = dr.decoder_step(p, cfg, keys, values, x, c, h, att, None)
        alphas.append(alpha.reshape(B * W, H * M))
        logits = (y @ p['W_o'] + p['b_o']).reshape(B, W, V)
        step_lp = dr.log_softmax(logits, axis=-1)
        fin_row = np.full(V, F32_MIN, dt); fin_row[cfg.end_id] = 0
        step_lp = np.where(finished[:, :, None], fin_row[None, None, :], step_lp)
        total = log_probs[:, :, None] + step_lp
        flat_total = total.reshape(B, W * V)
        if length_penalty_weight != 0:
            add = np.ones(V, np.int64); add[cfg.end_id] = 0
            new_len = lengths[:, :, None] + add[None, None, :] * (~finished)[:, :, None]
            pen = (((np.float32(5.) + new_len.astype(dt)) / np.float32(6.)) ** np.float32(length_penalty_weight)).astype(dt)
            flat = (total / pen).reshape(B, W * V)
        else:
            flat = flat_total
        # top_k: descending value, lower flat index first among equals
        order = np.argsort(-flat, axis=1, kind='stable')[:, :W]
        next_scores = np.take_along_axis(flat, order, axis=1)
        next_totals = np.take_along_axis(flat_total, order, axis=1)
        word = (order % V).astype(np.int32)
        parent = (order // V).astype(np.int32)
        prev_fin = finished[bidx, parent]
        next_fin = prev_fin | (word == cfg.end_id)
        lengths = lengths[bidx, parent] + (~prev_fin).astype(np.int64)
        log_probs = next_totals.astype(dt)
        finished = next_fin
        gidx = (bidx * W + parent).reshape(-1)
        c, h, att = c[gidx], h[gidx], att[gidx]
        step_ids.append(word); parent_ids.append(parent); scores_all.append(next_scores)
        ids = word.reshape(-1).astype(np.int64)
        if finished.all():
            break
    step_ids = np.stack(step_ids); parent_ids = np.stack(parent_ids); scores_all = np.stack(scores_all)
    max_len = lengths.max(axis=1).astype(np.int32)
    predicted = gather_tree(step_ids, parent_ids, max_len, cfg.end_id)
    hist = gather_tree_from_array(np.stack(alphas), parent_ids, lengths)
    if return_debug:
        return predicted, scores_all, hist, dict(step_ids=step_ids, parent_ids=parent_ids,
                                                 lengths=lengths)
    return predicted, scores_all, hist


def post_process_beam(predicted, scores, hist, cfg, beam, top_beam=True):
    """model_base.py:272-314 for beam outputs."""
    T = predicted.shape[0]
    if top_beam:
        ids = predicted[:, :, 0].T                      # [B,T]
        sc = scores[:, :, 0].T
    else:
        ids = predicted.transpose(2, 1, 0)              # [W,B,T]
        sc = scores.transpose(2, 1, 0)
    H = cfg.attn_num_heads
    m = hist.reshape(T, -1, beam, hist.shape[-1])[:, :, 0, :]      # beam 0
    m = m.reshape(T, m.shape[1], H, -1).transpose(1, 2, 0, 3)      # [B,H,T,M]
    return ids, sc, m
