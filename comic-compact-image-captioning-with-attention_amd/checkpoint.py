"""Checkpoint I/O of the hot path.

The reference saves TensorFlow V2 checkpoints `model_compact-N` (variables under `Model/`)
and `model-N` (everything, incl. global_step and the Adam slots) with two Savers
(src/train_fn.py:67-70,131-132) and restores them with the three-way logic of
`ModelBase.restore_model` (src/model_base.py:422-490).  This module keeps the file NAMES,
the variable NAMES and the restore logic; the container is `.npz`
(one array per TF variable name).  Reading/writing the TF tensor-bundle container itself is
the next row of SURVEY §8f (no real checkpoint is available offline to verify against).
"""
from __future__ import annotations

import os
import re

import numpy as np

from .decoder import TF_NAMES

DEC_SCOPE = 'Model/decoder/rnn_decoder/'
CNN_SCOPE = 'Model/encoder/cnn/'


def decoder_var_names(spec):
    out = {}
    for k in spec.param_shapes():
        n = TF_NAMES[k]
        if isinstance(n, dict):
            n = n[spec.init_method]
        if spec.method == 'dot':
            n = n.replace('multi_add_attention/', 'MultiHeadDot/')
        out[k] = DEC_SCOPE + n
    return out


def save(path_prefix, global_step, cnn_params, dec_spec, dec_params, extra=None, max_to_keep=None):
    """Write `<path_prefix>-<global_step>.npz`; returns the path."""
    arrays = {}
    for k, v in cnn_params.items():
        arrays[CNN_SCOPE + k] = np.asarray(v)
    names = decoder_var_names(dec_spec)
    for k, v in dec_params.items():
        arrays[names[k]] = np.asarray(v)
    for k, v in (extra or {}).items():
        arrays[k] = np.asarray(v)
    arrays['global_step'] = np.asarray(global_step, np.int32)
    path = '%s-%d.npz' % (path_prefix, int(global_step))
    os.makedirs(os.path.dirname(path) or '.', exist_ok=True)
    np.savez(path, **arrays)
    if max_to_keep:
        d, base = os.path.split(path_prefix)
        pat = re.compile(r'^%s-(\d+)\.npz$' % re.escape(base))
        found = sorted((int(m.group(1)), f) for f in os.listdir(d or '.') for m in [pat.match(f)] if m)
        for _, f in found[:-max_to_keep]:
            os.remove(os.path.join(d or '.', f))
    return path


def latest_checkpoint(directory, prefix='model'):
    """tf.train.latest_checkpoint counterpart for `<prefix>-N.npz` files."""
    pat = re.compile(r'^%s-(\d+)\.npz$' % re.escape(prefix))
    best = None
    for f in os.listdir(directory):
        m = pat.match(f)
        if m and (best is None or int(m.group(1)) > best[0]):
            best = (int(m.group(1)), os.path.join(directory, f))
    return best[1] if best else None


def load(path):
    with np.load(path, allow_pickle=False) as z:
        return {k: z[k] for k in z.files}


def restore(path, cnn_param_names, dec_spec, resume_training=False, exclude_scopes=None):
    """ModelBase.restore_model logic: returns (cnn_params | None, dec_params | None, extra).
    * every model variable present -> whole `Model/` (and, when resuming, step + Adam slots)
    * otherwise -> CNN only, names with the `Model/encoder/cnn/` prefix stripped (slim ckpt)."""
    arrays = load(path)
    exc = [s.strip() for s in (exclude_scopes or '').split(',') if s.strip()]

    def excluded(name):
        return any(re.search(e, name) for e in exc)
    names = decoder_var_names(dec_spec)
    model_vars = [CNN_SCOPE + n for n in cnn_param_names] + list(names.values())
    if all(v in arrays for v in model_vars):
        cnn = {n: arrays[CNN_SCOPE + n] for n in cnn_param_names if not excluded(CNN_SCOPE + n)}
        dec = {k: arrays[v] for k, v in names.items() if not excluded(v)}
        extra = {k: v for k, v in arrays.items() if not k.startswith('Model/')} if resume_training else {}
        return cnn, dec, extra
    cnn = {}
    for n in cnn_param_names:
        if excluded(CNN_SCOPE + n):
            continue
        if n in arrays:
            cnn[n] = arrays[n]
        elif CNN_SCOPE + n in arrays:
            cnn[n] = arrays[CNN_SCOPE + n]
        else:
            raise KeyError('checkpoint %s has no variable %s' % (path, n))
    return cnn, None, {}
