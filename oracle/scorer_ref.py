"""CPU restatement (pure Python, float64) of the SCST reward scorer.

TEST INFRASTRUCTURE -- see oracle/__init__.py.  PINNED: checked against outputs of
the reference's own Python through tests/golden/scorer_golden.json (CIDEr-D imported
as-is; BLEU / captionScorer / prepro_ngrams after a lib2to3 pass on a temp copy; see
oracle/make_golden.py).

  captionScorer.get_hypo_scores  common/scst/scorers.py:43-171
  BleuSilent ................... common/scst/scorers.py:174-197
  CiderD.compute_score ......... common/scst/cider_ruotianluo/pyciderevalcap/ciderD/ciderD.py:30-56
  CiderScorer .................. .../ciderD/ciderD_scorer.py:52-222
  BleuScorer ................... common/coco_caption/pycocoevalcap/bleu/bleu_scorer.py:23-263
  document frequency ........... common/scst/prepro_ngrams.py:61-73, :122-151
"""
from __future__ import annotations

import math
from collections import defaultdict

import numpy as np


def precook(s, n=4):
    words = s.split()
    counts = defaultdict(int)
    for k in range(1, n + 1):
        for i in range(len(words) - k + 1):
            counts[tuple(words[i:i + k])] += 1
    return len(words), counts


# ------------------------------------------------------------------ CIDEr-D --
def compute_doc_freq(ref_lists, n=4):
    """prepro_ngrams.compute_doc_freq: one count per image containing the n-gram."""
    df = defaultdict(float)
    for refs in ref_lists:
        seen = set()
        for r in refs:
            seen.update(precook(r, n)[1].keys())
        for ng in seen:
            df[ng] += 1
    return df


def build_df_from_refs(ref_lists, n=4):
    """prepro_ngrams.py:122-151: refs get ' <EOS>' appended for the df statistics."""
    refs = [[r + ' <EOS>' for r in rl] for rl in ref_lists]
    return dict(document_frequency=compute_doc_freq(refs, n), ref_len=len(refs))


class CiderD:
    def __init__(self, document_frequency, ref_len, n=4, sigma=6.0):
        self.n, self.sigma = n, sigma
        self.df = document_frequency
        self.log_ref_len = math.log(float(ref_len))

    def _vec(self, cnts):
        vec = [dict() for _ in range(self.n)]
        norm = [0.0] * self.n
        length = 0
        for ng, tf in cnts.items():
            df = math.log(max(1.0, self.df.get(ng, 0.0)))
            k = len(ng) - 1
            vec[k][ng] = float(tf) * (self.log_ref_len - df)
            norm[k] += vec[k][ng] ** 2
            if k == 1:                       # reference quirk: length = number of bigrams
                length += tf
        return vec, [math.sqrt(x) for x in norm], length

    def _sim(self, vh, vr, nh, nr, lh, lr):
        delta = float(lh - lr)
        val = [0.0] * self.n
        for k in range(self.n):
            for ng, w in vh[k].items():
                r = vr[k].get(ng, 0.0)
                val[k] += min(w, r) * r
            if nh[k] != 0 and nr[k] != 0:
                val[k] /= (nh[k] * nr[k])
            val[k] *= math.e ** (-(delta ** 2) / (2 * self.sigma ** 2))
        return val

    def score_one(self, hypo, refs):
        vh, nh, lh = self._vec(precook(hypo, self.n)[1])
        score = np.zeros(self.n)
        for r in refs:
            vr, nr, lr = self._vec(precook(r, self.n)[1])
            score += np.array(self._sim(vh, vr, nh, nr, lh, lr))
        return float(np.mean(score) / len(refs) * 10.0)

    def compute_score(self, gts, res):
        scores = np.array([self.score_one(res[k][0], gts[k]) for k in gts])
        return float(scores.mean()), scores


# --------------------------------------------------------------------- BLEU --
def bleu_sentence_scores(hypo, refs, n=4):
    """Per-sentence BLEU-1..n as appended to `bleu_list` (bleu_scorer.py:215-243),
    effective reference length option 'closest'."""
    small, tiny = 1e-9, 1e-15
    reflens, maxcounts = [], {}
    for r in refs:
        rl, c = precook(r, n)
        reflens.append(rl)
        for ng, cnt in c.items():
            maxcounts[ng] = max(maxcounts.get(ng, 0), cnt)
    testlen, counts = precook(hypo, n)
    reflen = min((abs(l - testlen), l) for l in reflens)[1]
    guess = [max(0, testlen - k + 1) for k in range(1, n + 1)]
    correct = [0] * n
    for ng, cnt in counts.items():
        correct[len(ng) - 1] += min(maxcounts.get(ng, 0), cnt)
    out = []
    bleu = 1.0
    for k in range(n):
        bleu *= (float(correct[k]) + tiny) / (float(guess[k]) + small)
        out.append(bleu ** (1.0 / (k + 1)))
    ratio = (testlen + tiny) / (reflen + small)
    if ratio < 1:
        out = [b * math.exp(1 - 1 / ratio) for b in out]
    return out


# ------------------------------------------------------------ captionScorer --
class CaptionScorer:
    """scorers.captionScorer restated.  `weights` = dict(ciderD=float, bleu=[w1..w4])."""

    def __init__(self, df_dict, weights):
        self.cider = CiderD(df_dict['document_frequency'], df_dict['ref_len'])
        self.weights = weights

    def get_hypo_scores(self, refs, sample, greedy):
        ng, ns = len(greedy), len(sample)
        assert len(refs) == ng and ns % ng == 0
        mult = ns // ng
        # key order: greedy 0..ng-1, then sample ng..ng+ns-1; sample i uses refs[i % ng]
        hyp = [g[0] for g in greedy] + [s[0] for s in sample]
        gts = [refs[i] for i in range(ng)] + [refs[i % ng] for i in range(ns)]
        total = np.zeros(ng + ns)
        w = self.weights
        if 'ciderD' in w and np.amax(w['ciderD']) > 0:
            total += np.array([self.cider.score_one(h, r) for h, r in zip(hyp, gts)]) * w['ciderD']
        if 'bleu' in w and np.amax(w['bleu']) > 0:
            b = np.array([bleu_sentence_scores(h, r) for h, r in zip(hyp, gts)])   # [N,4]
            for i, wi in enumerate(w['bleu']):
                total += b[:, i] * wi
        sc_greedy, sc_sample = total[:ng], total[ng:]
        if ns > ng:
            sc_greedy = np.concatenate([sc_greedy] * mult)
        return sample, sc_sample, sc_greedy
