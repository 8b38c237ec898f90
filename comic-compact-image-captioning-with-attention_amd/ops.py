"""Host helpers mirroring the reference's `common/ops.py` for the hot path.

Only the pieces the captioning path calls are provided: `number_to_base`
(common/ops.py:25-40) and the token/caption conversions that live next to it in the
reference (`infer_fn.id_to_caption`, src/infer_fn.py:36-75; the radix table and
`captions_to_batched_ids`, common/inputs/manager_image_caption.py:240-254, :477-509).
"""
from __future__ import annotations

import numpy as np


def number_to_base(n, base):
    """Function to convert any base-10 integer to base-N (digits, most significant first)."""
    if base < 2:
        raise ValueError('Base cannot be less than 2.')
    if n == 0:
        return [0]
    sign = 1
    if n < 0:
        sign, n = -1, -n
    digits = []
    while n:
        n, r = divmod(n, base)
        digits.append(sign * int(r))
    digits.reverse()
    return digits


def base_n_to_dec(digits, base):
    """Inverse of `number_to_base` (infer_fn._baseN_arr_to_dec)."""
    value = 0
    for d in digits:
        value = value * base + int(d)
    return value


def _word_array(config):
    """itow as an object array indexed by word id (built once per table; None where the table has no such id)."""
    cache = getattr(config, '_itow_list', None)
    if cache is None or cache[0] is not config.itow:
        lst = [config.itow.get(str(i)) for i in range(len(config.itow))]                  # ids run to len - 2: '-1' is <PAD>
        cache = (config.itow, lst, np.array(lst, dtype=object))
        try:
            config._itow_list = cache
        except AttributeError:
            pass
    return cache[2]


def id_to_caption(ids, config):
    """ids [N,T] -> list of N strings.  Radix: keep ids in [0, base), drop one trailing id
    when the count is not a multiple of the word length, decode base-N groups, skip word ids
    >= vocab; word/char: drop negatives and <EOS>."""
    ids = np.asarray(ids)
    captions = []
    if config.token_type == 'radix':
        base = config.radix_base
        vocab_size = len(config.itow)
        word_len = len(number_to_base(vocab_size, base))
        # word list indexed by id, built once per itow table (the SCST loop decodes (1 + beam) * batch rows per step)
        cache = getattr(config, '_itow_list', None)
        if cache is None or cache[0] is not config.itow:
            lst = [config.itow.get(str(i)) for i in range(vocab_size)]                    # ids run to len - 2: '-1' is <PAD>
            cache = (config.itow, lst, np.array(lst, dtype=object))
            try:
                config._itow_list = cache
            except AttributeError:
                pass
        words_of = cache[1]
        def lookup(wids):
            words = [words_of[w] for w in wids]
            if None in words:                             # the dict lookup of the reference raises here
                raise KeyError(str(wids[words.index(None)]))
            return ' '.join(words)
        if word_len <= 2 and ids.ndim == 2 and ids.shape[1] > 0:
            # all rows at once (the SCST loop decodes (1 + beam) * batch rows per step): compact the digits of every row
            # to the left in order, drop one trailing digit of odd rows, combine digit pairs
            valid = (ids >= 0) & (ids < base)
            order = np.argsort(~valid, axis=1, kind='stable')
            comp = np.take_along_axis(np.where(valid, ids, 0).astype(np.int64), order, axis=1)
            n_words = valid.sum(axis=1) // word_len
            if word_len == 2:
                if comp.shape[1] % 2:
                    comp = np.concatenate([comp, np.zeros((comp.shape[0], 1), np.int64)], axis=1)
                wid = comp[:, 0::2] * base + comp[:, 1::2]
            else:
                wid = comp
            keep = (np.arange(wid.shape[1])[None, :] < n_words[:, None]) & (wid < vocab_size)
            flat_ids = wid[keep]                          # row-major: the words of row 0, then row 1, ...
            flat = cache[2][flat_ids].tolist()
            if None in flat:                              # the dict lookup of the reference raises here
                raise KeyError(str(int(flat_ids[flat.index(None)])))
            ends = np.cumsum(keep.sum(axis=1)).tolist()
            return [' '.join(flat[a:b]) for a, b in zip([0] + ends[:-1], ends)]
        weights = base ** np.arange(word_len - 1, -1, -1, dtype=np.int64)       # most significant digit first
        for row in ids:
            keep = row[(row >= 0) & (row < base)]
            if len(keep) % word_len:                     # the reference pops ONE id, whatever the remainder
                keep = keep[:-1]
            n = len(keep) // word_len
            wid = keep[:n * word_len].reshape(n, word_len).astype(np.int64) @ weights
            if len(keep) > n * word_len:                 # word_len > 2: a short last group decodes as it stands
                wid = np.append(wid, base_n_to_dec(keep[n * word_len:].tolist(), base))
            captions.append(lookup(wid[wid < vocab_size].tolist()))
        return captions
    eos = config.wtoi['<EOS>']
    joiner = ' ' if config.token_type == 'word' else ''
    for row in ids:
        captions.append(joiner.join(config.itow[str(int(w))] for w in row if w >= 0 and w != eos))
    return captions


def radix_ids_to_captions_and_ids(ids, config, radix_wtoi):
    """The SCST loop's round trip `captions_to_batched_ids([[s] for s in id_to_caption(ids, config)], ...)` in one pass
    (src/train_fn.py:226-253: sampled ids -> text for the scorer -> target ids for the update): -> (captions, ids matrix),
    equal to the two calls.  The target ids are assembled from the decoded WORD ids through a per-word digit table
    (`radix_wtoi[itow[w]]`, `<UNK>` for a word the table lacks) instead of splitting the joined strings again; vocabularies
    with a word that is empty or contains white space (where join + split is not the identity) take the two calls."""
    ids = np.asarray(ids)
    base = config.radix_base
    vocab_size = len(config.itow)
    word_len = len(number_to_base(vocab_size, base))
    cache = getattr(config, '_radix_roundtrip', None)
    if cache is None or cache[0] is not config.itow or cache[1] is not radix_wtoi:
        unk = radix_wtoi['<UNK>']
        dig = np.full((vocab_size, max(word_len, 1)), 0, np.int64)
        dlen = np.zeros(vocab_size, np.int64)
        simple = True
        for i in range(vocab_size):
            w = config.itow.get(str(i))
            if w is None:
                continue
            simple = simple and bool(w) and w.split() == [w]
            d = radix_wtoi.get(w, unk)
            simple = simple and len(d) <= dig.shape[1]
            if simple:
                dig[i, :len(d)] = d
                dlen[i] = len(d)
        cache = (config.itow, radix_wtoi, simple, dig, dlen)
        try:
            config._radix_roundtrip = cache
        except AttributeError:
            pass
    simple, dig, dlen = cache[2:]
    if not (simple and config.token_type == 'radix' and word_len <= 2 and ids.ndim == 2 and ids.shape[1] > 0):
        caps = id_to_caption(ids, config)
        return caps, captions_to_batched_ids([[c] for c in caps], config, radix_wtoi)
    # the vectorised decode of id_to_caption, keeping the word ids
    valid = (ids >= 0) & (ids < base)
    order = np.argsort(~valid, axis=1, kind='stable')
    comp = np.take_along_axis(np.where(valid, ids, 0).astype(np.int64), order, axis=1)
    n_words = valid.sum(axis=1) // word_len
    if word_len == 2:
        if comp.shape[1] % 2:
            comp = np.concatenate([comp, np.zeros((comp.shape[0], 1), np.int64)], axis=1)
        wid = comp[:, 0::2] * base + comp[:, 1::2]
    else:
        wid = comp
    keep = (np.arange(wid.shape[1])[None, :] < n_words[:, None]) & (wid < vocab_size)
    flat_ids = wid[keep]
    words = _word_array(config)[flat_ids].tolist()
    if None in words:
        raise KeyError(str(int(flat_ids[words.index(None)])))
    counts = keep.sum(axis=1)
    ends = np.cumsum(counts)
    starts = ends - counts
    captions = [' '.join(words[a:b]) for a, b in zip(starts.tolist(), ends.tolist())]
    # target rows: <GO> | digits of every word | <EOS>, PAD behind
    N = ids.shape[0]
    dl = dlen[flat_ids]
    csum = np.cumsum(dl)
    nz = counts > 0
    before = np.zeros(N, np.int64)                 # digits of all words in front of the row's first one
    before[nz] = csum[starts[nz]] - dl[starts[nz]]
    row_digits = np.zeros(N, np.int64)
    row_digits[nz] = csum[ends[nz] - 1] - before[nz]
    row_of = np.repeat(np.arange(N), counts)
    first = csum - dl - np.repeat(before, counts)  # position of a word's first digit among its row's digits
    go, eos, pad = radix_wtoi['<GO>'], radix_wtoi['<EOS>'], config.wtoi['<PAD>']
    assert len(go) == 1 and len(eos) == 1
    width = int(row_digits.max()) + 2 if N else 2
    out = np.full((N, width), pad, np.int64)
    out[:, 0] = go[0]
    for k in range(dig.shape[1]):
        m = dl > k
        out[row_of[m], 1 + first[m] + k] = dig[flat_ids[m], k]
    out[np.arange(N), 1 + row_digits] = eos[0]
    return captions, out


def build_radix_wtoi(wtoi, radix_base):
    """word -> zero-left-padded base-N digits; <GO> -> [base], <EOS> -> [base+1], <PAD> -> [-1]."""
    assert wtoi['<PAD>'] == -1
    max_word_len = len(number_to_base(len(wtoi), radix_base))
    special = {'<GO>': [radix_base], '<EOS>': [radix_base + 1], '<PAD>': [-1]}
    table = {}
    for word, idx in wtoi.items():
        if word in special:
            table[word] = special[word]
        else:
            digits = number_to_base(idx, radix_base)
            table[word] = [0] * (max_word_len - len(digits)) + digits
    return table


def captions_to_batched_ids(hypos, config, radix_wtoi=None):
    """List of [string] hypotheses -> padded int matrix used as SCST targets."""
    token_type = config.token_type
    assert token_type in ('radix', 'word', 'char')
    rows = []
    for h in hypos:
        if token_type == 'char':
            r = [config.wtoi['<GO>']] + [config.wtoi[ch] for ch in h[0]] + [config.wtoi['<EOS>']]
        else:
            toks = ['<GO>'] + h[0].split() + ['<EOS>']
            if token_type == 'radix':
                r = [d for w in toks for d in radix_wtoi.get(w, radix_wtoi['<UNK>'])]
            else:
                r = [config.wtoi.get(w, config.wtoi['<UNK>']) for w in toks]
        rows.append(r)
    width = max(len(r) for r in rows)
    assert width > 1
    out = np.full((len(rows), width), config.wtoi['<PAD>'], np.int64)
    for i, r in enumerate(rows):
        out[i, :len(r)] = r
    return out
