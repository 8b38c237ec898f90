"""Checkpoint I/O of the hot path.

The reference saves TensorFlow V2 checkpoints `model_compact-N` (variables under `Model/`)
and `model-N` (everything, incl. global_step and the Adam slots) with two Savers
(src/train_fn.py:67-70,131-132) and restores them with the three-way logic of
`ModelBase.restore_model` (src/model_base.py:422-490).  This module keeps the file NAMES,
the variable NAMES and the restore logic.  Two containers:
  * `.npz`  (default for saving: one array per TF variable name)
  * the TF checkpoint-V2 tensor bundle `<prefix>.index` + `<prefix>.data-00000-of-00001`
    (`tf_bundle.py`; `save(..., fmt='tf')`, restored transparently), so slim CNN checkpoints and
    the reference's own `model_compact-N` / `model-N` files load unchanged.  The bundle code is
    written from the published formats and pinned by round trips only (no TensorFlow here).
In the bundle the Adam slots use tf.train.AdamOptimizer's names under the train-op scope
(`optimise/caption/<variable>/Adam`, `/Adam_1`, `optimise/caption/beta{1,2}_power`,
model_base.py:387-401); the `.npz` container keeps them as flat arrays.
"""
from __future__ import annotations

import os
import re

import numpy as np

from . import tf_bundle
from .decoder import TF_NAMES, CELL_SCOPES, CELL_VARS, STEP_SCOPE, STEP_VARS

DEC_SCOPE = 'Model/decoder/rnn_decoder/'
CNN_SCOPE = 'Model/encoder/cnn/'


def decoder_var_names(spec, legacy_flat=False):
    """{parameter key: TensorFlow variable name} of the decoder (see decoder.TF_NAMES).
    legacy_flat: the names rounds 1-3 of this package wrote (every variable directly under rnn_decoder/, cell variables
    always under rnn_init_input/): still accepted on restore."""
    out = {}
    first_input = getattr(spec, 'init_method', 'first_input') == 'first_input'
    for k in spec.param_shapes():
        n = TF_NAMES[k]
        if isinstance(n, dict):
            n = n[spec.init_method]
        if spec.method == 'dot':
            n = n.replace('multi_add_attention/', 'MultiHeadDot/')
        rnn = getattr(spec, 'rnn_name', 'LSTM')
        if k in ('K', 'b') and rnn != 'LSTM':
            scope, kn, bn = CELL_SCOPES[rnn]
            n = '%s/%s' % (scope, kn if k == 'K' else bn)
        if k in CELL_VARS:
            n = ('rnn_init_input/' if (first_input or legacy_flat) else STEP_SCOPE) + n
        elif k in STEP_VARS and not legacy_flat:
            n = STEP_SCOPE + n
        out[k] = DEC_SCOPE + n
    return out


def _rename_legacy(arrays, dec_spec):
    """Checkpoints written before the names were derived from the reference's scopes: map them to the current names."""
    cur, old = decoder_var_names(dec_spec), decoder_var_names(dec_spec, legacy_flat=True)
    ren = {old[k]: cur[k] for k in cur if old[k] != cur[k]}
    if not ren or not any(o in arrays for o in ren) or any(c in arrays for c in ren.values()):
        return arrays
    out = {}
    for name, v in arrays.items():
        hit = next((o for o in ren if name == o or name == ADAM_SCOPE + o + '/Adam' or name == ADAM_SCOPE + o + '/Adam_1' or
                    name == ADAM_SCOPE + o + '/Momentum'), None)
        out[name.replace(hit, ren[hit]) if hit else name] = v
    return out


def save(path_prefix, global_step, cnn_params, dec_spec, dec_params, extra=None, max_to_keep=None, fmt='npz'):
    """Write `<path_prefix>-<global_step>.npz` (or the TF bundle pair for fmt='tf'); returns the path."""
    arrays = {}
    for k, v in cnn_params.items():
        arrays[CNN_SCOPE + k] = np.asarray(v)
    names = decoder_var_names(dec_spec)
    for k, v in dec_params.items():
        arrays[names[k]] = np.asarray(v)
    for k, v in (extra or {}).items():
        arrays[k] = np.asarray(v)
    if fmt == 'tf':
        arrays['global_step'] = np.asarray(global_step, np.int64)      # tf.train.get_or_create_global_step
        prefix = '%s-%d' % (path_prefix, int(global_step))
        tf_bundle.write_bundle(prefix, arrays)
        d, base = os.path.split(prefix)
        kept = tf_bundle.update_checkpoint_state(d or '.', base)
        if max_to_keep:
            mine = [p for p in kept if p.rsplit('-', 1)[0] == os.path.basename(path_prefix)]
            for p in mine[:-max_to_keep]:
                for f in (p + '.index', tf_bundle.data_path(p)):
                    if os.path.isfile(os.path.join(d or '.', f)):
                        os.remove(os.path.join(d or '.', f))
            tf_bundle.prune_checkpoint_state(d or '.', set(mine[:-max_to_keep]))
        return prefix
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
    """tf.train.latest_checkpoint counterpart: the `<prefix>-N.npz` file or TF bundle `<prefix>-N.index` (returned as its
    prefix path) with the highest step N."""
    best = None
    for ext in (r'\.npz', r'\.index'):
        pat = re.compile(r'^%s-(\d+)%s$' % (re.escape(prefix), ext))
        for f in os.listdir(directory):
            m = pat.match(f)
            if m and (best is None or int(m.group(1)) > best[0]):       # the highest step wins, whatever its container
                best = (int(m.group(1)), os.path.join(directory, f))
    if best:
        return best[1] if best[1].endswith('.npz') else best[1][:-len('.index')]
    return None


def is_tf_bundle(path):
    return os.path.isfile(path + '.index') or path.endswith('.index') or path.endswith('.ckpt')


def load(path):
    """{variable name: array} from an `.npz` file or a TF tensor bundle (prefix, `.index` path or
    a slim `*.ckpt` file in the older single-file naming that still is a bundle prefix)."""
    if path.endswith('.index'):
        path = path[:-len('.index')]
    if os.path.isfile(path + '.index'):
        return tf_bundle.read_bundle(path)
    with np.load(path, allow_pickle=False) as z:
        return {k: z[k] for k in z.files}


ADAM_SCOPE = 'optimise/caption/'


def adam_to_tf(dec_spec, m, v, t, beta1=0.9, beta2=0.999):
    """Flat-buffer Adam state -> tf.train.AdamOptimizer's variables ({name: array}; m, v keyed like
    the decoder parameters).  After t applied updates TF holds beta^(t+1) in the power accumulators."""
    names = decoder_var_names(dec_spec)
    out = {}
    for k, n in names.items():
        out[ADAM_SCOPE + n + '/Adam'] = np.asarray(m[k], np.float32)
        out[ADAM_SCOPE + n + '/Adam_1'] = np.asarray(v[k], np.float32)
    out[ADAM_SCOPE + 'beta1_power'] = np.asarray(beta1 ** (t + 1), np.float32)
    out[ADAM_SCOPE + 'beta2_power'] = np.asarray(beta2 ** (t + 1), np.float32)
    return out


def momentum_to_tf(dec_spec, accum):
    """tf.train.MomentumOptimizer keeps ONE slot per variable, named `<scope>/<var>/Momentum`, and no power accumulators."""
    names = decoder_var_names(dec_spec)
    return {ADAM_SCOPE + n + '/Momentum': np.asarray(accum[k], np.float32) for k, n in names.items()}


def momentum_from_tf(dec_spec, arrays):
    """Inverse of momentum_to_tf: -> accum dict keyed like the decoder parameters, or None."""
    names = decoder_var_names(dec_spec)
    if not all(ADAM_SCOPE + n + '/Momentum' in arrays for n in names.values()):
        return None
    return {k: arrays[ADAM_SCOPE + n + '/Momentum'] for k, n in names.items()}


def adam_from_tf(dec_spec, arrays):
    """Inverse of adam_to_tf: -> (m, v) dicts keyed like the decoder parameters, or None."""
    names = decoder_var_names(dec_spec)
    if not all(ADAM_SCOPE + n + '/Adam' in arrays and ADAM_SCOPE + n + '/Adam_1' in arrays for n in names.values()):
        return None
    return ({k: arrays[ADAM_SCOPE + n + '/Adam'] for k, n in names.items()},
            {k: arrays[ADAM_SCOPE + n + '/Adam_1'] for k, n in names.items()})


def restore(path, cnn_param_names, dec_spec, resume_training=False, exclude_scopes=None, head_names='unset'):
    """ModelBase.restore_model logic: returns (cnn_params | None, dec_params | None, extra).
    * every model variable present -> whole `Model/` (and, when resuming, step + Adam slots)
    * otherwise -> CNN only, names with the `Model/encoder/cnn/` prefix stripped (slim ckpt).
    head_names (a list of further `Model/` variable names, or None: the legacy encoder head, model_base.py:80-91):
    when given, they count as model variables and a 4-tuple (cnn, dec, extra, head arrays | None) is returned."""
    if head_names != 'unset':
        hn = list(head_names or [])
        arrays = _rename_legacy(load(path), dec_spec)
        r = _restore(arrays, path, cnn_param_names, dec_spec, resume_training, exclude_scopes, hn)
        head = {n: arrays[n] for n in hn} if (r[1] is not None and hn) else None
        return r + (head,)
    return _restore(_rename_legacy(load(path), dec_spec), path, cnn_param_names, dec_spec, resume_training, exclude_scopes, [])


def _restore(arrays, path, cnn_param_names, dec_spec, resume_training, exclude_scopes, more_model_vars):
    exc = [s.strip() for s in (exclude_scopes or '').split(',') if s.strip()]

    def excluded(name):
        return any(re.search(e, name) for e in exc)
    names = decoder_var_names(dec_spec)
    model_vars = [CNN_SCOPE + n for n in cnn_param_names] + list(names.values()) + list(more_model_vars)
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
