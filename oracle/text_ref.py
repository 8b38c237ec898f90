"""CPU restatement of the host-side token helpers on the hot path.

TEST INFRASTRUCTURE -- see oracle/__init__.py.  PINNED: checked against the
reference's own functions (ast-extracted and exec'd by oracle/make_golden.py)
through tests/golden/text_golden.json.

  number_to_base ............. common/ops.py:25-40
  _baseN_arr_to_dec .......... src/infer_fn.py:36-43
  id_to_caption .............. src/infer_fn.py:46-75
  radix table ................ common/inputs/manager_image_caption.py:240-254
  captions_to_batched_ids .... common/inputs/manager_image_caption.py:477-509
"""
from __future__ import annotations

import numpy as np


def number_to_base(n, base):
    if base < 2:
        raise ValueError('Base cannot be less than 2.')
    if n == 0:
        return [0]
    sign = -1 if n < 0 else 1
    n = abs(n)
    digits = []
    while n:
        digits.append(sign * int(n % base))
        n //= base
    return digits[::-1]


def base_n_to_dec(arr, base):
    r = 0
    for d in arr:
        r = r * base + int(d)
    return r


def id_to_caption(ids, token_type, itow, wtoi, radix_base=256):
    ids = np.asarray(ids)
    caps = []
    if token_type == 'radix':
        vocab = len(itow)
        wl = len(number_to_base(vocab, radix_base))
        for i in range(ids.shape[0]):
            row = [int(w) for w in ids[i] if 0 <= w < radix_base]
            if len(row) % wl != 0:
                row = row[:-1]                      # reference drops ONE trailing id only
            sent = []
            for j in range(0, len(row), wl):
                wid = base_n_to_dec(row[j:j + wl], radix_base)
                if wid < vocab:
                    sent.append(itow[str(wid)])
            caps.append(' '.join(sent))
    else:
        eos = wtoi['<EOS>']
        sep = ' ' if token_type == 'word' else ''
        for i in range(ids.shape[0]):
            caps.append(sep.join(itow[str(int(w))] for w in ids[i] if w >= 0 and w != eos))
    return caps


def build_radix_wtoi(wtoi, radix_base):
    wl = len(number_to_base(len(wtoi), radix_base))
    assert wtoi['<PAD>'] == -1
    table = {}
    for k, v in wtoi.items():
        if k == '<GO>':
            table[k] = [radix_base]
        elif k == '<EOS>':
            table[k] = [radix_base + 1]
        elif k == '<PAD>':
            table[k] = [-1]
        else:
            d = number_to_base(v, radix_base)
            table[k] = [0] * (wl - len(d)) + d
    return table


def captions_to_batched_ids(hypos, token_type, wtoi, radix_wtoi=None):
    rows = []
    for h in hypos:
        if token_type == 'radix':
            toks = ['<GO>'] + h[0].split() + ['<EOS>']
            r = np.concatenate([radix_wtoi.get(w, radix_wtoi['<UNK>']) for w in toks])
        elif token_type == 'word':
            toks = ['<GO>'] + h[0].split() + ['<EOS>']
            r = np.array([wtoi.get(w, wtoi['<UNK>']) for w in toks])
        else:
            r = np.array([wtoi['<GO>']] + [wtoi[ch] for ch in h[0]] + [wtoi['<EOS>']])
        rows.append(r)
    L = max(len(r) for r in rows)
    assert L > 1
    out = np.full((len(rows), L), wtoi['<PAD>'], np.int64)
    for i, r in enumerate(rows):
        out[i, :len(r)] = r
    return out
