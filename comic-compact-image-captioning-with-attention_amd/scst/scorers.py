"""SCST reward scorer with the reference's `captionScorer` interface
(common/scst/scorers.py:29-171), computed by the multi-threaded C++ scorer in
libcomic_hip.so (`comic_scorer_*`).  The df statistics come from the reference's
`{pattern}scst-words.p` pickle unchanged (`{'document_frequency', 'ref_len'}`,
common/scst/prepro_ngrams.py:149-151)."""
from __future__ import annotations

import ctypes as C
import os
import pickle

import numpy as np

from .. import _lib as L


def load_df_pickle(path):
    with open(path, 'rb') as f:
        return pickle.load(f, encoding='latin1')


class captionScorer(object):
    """`metric_weights` = dict(ciderD=float, bleu=[w1, w2, w3, w4]) (train_fn.py:200)."""

    def __init__(self, path_to_cached_tokens, metric_weights, n_threads=None):
        self.lib = L.load()
        df = path_to_cached_tokens if isinstance(path_to_cached_tokens, dict) else load_df_pickle(path_to_cached_tokens)
        keys, counts = [], []
        for ng, cnt in df['document_frequency'].items():
            keys.append((' '.join(ng) if isinstance(ng, tuple) else ng).encode('utf-8'))
            counts.append(float(cnt))
        blob = b'\0'.join(keys) + b'\0'
        carr = (C.c_double * len(counts))(*counts)
        self._h = self.lib.comic_scorer_create(blob, carr, len(counts), float(df['ref_len']))
        if not self._h:
            raise L.ComicHipError('comic_scorer_create failed: %s' % self.lib.comic_last_error())
        self.weights = metric_weights
        self.n_threads = n_threads or min(16, os.cpu_count() or 1)     # 16 = the CPU share of one GPU on the boxes measured

    def __del__(self):
        try:
            if getattr(self, '_h', None):
                self.lib.comic_scorer_destroy(self._h)
                self._h = None
        except Exception:
            pass

    def _score(self, hypos, refs_per_hypo):
        n = len(hypos)
        hy = (C.c_char_p * n)(*[h.encode('utf-8') for h in hypos])
        enc = {}                                     # the hypotheses of an image share its references: encode each once
        flat = [enc.get(r) or enc.setdefault(r, r.encode('utf-8')) for rl in refs_per_hypo for r in rl]
        rf = (C.c_char_p * len(flat))(*flat)
        per = (C.c_int32 * n)(*[len(rl) for rl in refs_per_hypo])
        cider = (C.c_double * n)()
        bleu = (C.c_double * (4 * n))()
        L.check(self.lib.comic_scorer_score(self._h, hy, n, rf, per, cider, bleu, self.n_threads), 'scorer_score')
        return np.frombuffer(cider, np.float64).copy(), np.frombuffer(bleu, np.float64).reshape(n, 4).copy()

    def get_hypo_scores(self, refs, sample, greedy, best_hypo_only=False):
        """Same contract as the reference: `sample` = [[im0_h0],...,[imN_h0],[im0_h1],...];
        returns (final_hypo, sc_sample, sc_greedy) with the greedy baseline tiled to len(sample)."""
        assert isinstance(refs, list) and isinstance(sample, list) and isinstance(greedy, list)
        assert isinstance(refs[0], list) and isinstance(sample[0], list) and isinstance(greedy[0], list)
        assert len(refs) == len(greedy) and len(sample) % len(greedy) == 0
        ng, ns = len(greedy), len(sample)
        mult = ns // ng
        hypos = [g[0] for g in greedy] + [s[0] for s in sample]
        rlist = [list(refs[i]) for i in range(ng)] + [list(refs[i % ng]) for i in range(ns)]
        cider, bleu = self._score(hypos, rlist)
        total = np.zeros(ng + ns)
        w = self.weights
        if 'ciderD' in w and np.amax(w['ciderD']) > 0:
            total = total + cider * w['ciderD']
        if 'bleu' in w and np.amax(w['bleu']) > 0:
            for i, wi in enumerate(w['bleu']):
                total = total + bleu[:, i] * wi
        sc_greedy, sc_sample = total[:ng], total[ng:]
        if ns > ng and best_hypo_only:
            sc = sc_sample.reshape(mult, ng)
            best = np.argmax(sc, axis=0)
            final = [sample[i + ng * best[i]] for i in range(ng)]
            return final, np.amax(sc, axis=0), sc_greedy
        if ns > ng:
            sc_greedy = np.concatenate([sc_greedy] * mult)
        return sample, sc_sample, sc_greedy
