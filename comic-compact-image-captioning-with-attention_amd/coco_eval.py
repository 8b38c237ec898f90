"""Native corpus-level caption metrics for quick validation: BLEU-1..4, ROUGE-L and CIDEr.

Counterpart of the Python scorers behind `COCOEvalCap.evaluate` (reference
common/coco_caption/pycocoevalcap/eval.py:18-62; called from src/infer_fn.py:295-315):
  Bleu(4) ...... pycocoevalcap/bleu/bleu.py:19-43 over bleu_scorer.py:198-263 (closest reference length,
                 corpus statistics summed over the images)
  Rouge() ...... pycocoevalcap/rouge/rouge.py:13-105 (LCS F-score, beta = 1.2, best precision and best recall
                 over the references)
  Cider() ...... pycocoevalcap/cider/cider.py:16-53 over cider_scorer.py:53-191 (tf-idf with the document
                 frequencies of the EVALUATED reference set, clipped cosine, Gaussian length penalty sigma 6)
The reference tokenises with the Stanford PTB tokenizer and also runs METEOR and SPICE (Java: not available
here); `tokenize` below is the pure-Python stand-in (lower-case, punctuation split off and dropped as
ptbtokenizer.py:21-22,62-65 drops it), so absolute numbers can differ slightly from the Java pipeline on captions
with clitics or unusual punctuation.  On pre-tokenised strings the three scorers are pinned against the reference's
own Python by tests/golden/cocoeval_golden.json (oracle/make_golden.py cocoeval).

`evaluate_captions(annotation_file, result_json)` has the signature `infer_fn.evaluate_model` expects.
"""
from __future__ import annotations

import json
import math
import re
from collections import defaultdict

import numpy as np

PUNCTUATIONS = {"''", "'", "``", "`", "-LRB-", "-RRB-", "-LCB-", "-RCB-", ".", "?", "!", ",", ":", "-", "--", "...",
                ";"}
_TOKEN = re.compile(r"\.\.\.|--|[A-Za-z0-9]+(?:'[A-Za-z]+)?|[^\sA-Za-z0-9]")


def tokenize(caption):
    """Lower-cased tokens of one caption without the punctuation tokens the reference drops."""
    toks = _TOKEN.findall(caption.replace('\n', ' ').lower())
    return ' '.join(t for t in toks if t not in PUNCTUATIONS)


def _ngrams(s, n=4):
    words = s.split()
    counts = defaultdict(int)
    for k in range(1, n + 1):
        for i in range(len(words) - k + 1):
            counts[tuple(words[i:i + k])] += 1
    return len(words), counts


class Bleu(object):
    def __init__(self, n=4):
        self._n = n

    def compute_score(self, gts, res):
        """-> ([BLEU-1..n of the corpus], [[per-image BLEU-k] for k])."""
        assert list(gts.keys()) == list(res.keys())
        n, small, tiny = self._n, 1e-9, 1e-15
        tot_guess, tot_correct = [0] * n, [0] * n
        tot_test = tot_ref = 0
        per_image = [[] for _ in range(n)]
        for k in gts:
            hypo, refs = res[k], gts[k]
            assert isinstance(hypo, list) and len(hypo) == 1 and len(refs) >= 1
            reflens, maxcounts = [], {}
            for r in refs:
                rl, c = _ngrams(r, n)
                reflens.append(rl)
                for ng, cnt in c.items():
                    maxcounts[ng] = max(maxcounts.get(ng, 0), cnt)
            testlen, counts = _ngrams(hypo[0], n)
            reflen = min((abs(l - testlen), l) for l in reflens)[1]            # 'closest', ties -> shorter
            guess = [max(0, testlen - j) for j in range(n)]
            correct = [0] * n
            for ng, cnt in counts.items():
                correct[len(ng) - 1] += min(maxcounts.get(ng, 0), cnt)
            tot_test += testlen
            tot_ref += reflen
            bleu = 1.0
            ratio = (testlen + tiny) / (reflen + small)
            for j in range(n):
                tot_guess[j] += guess[j]
                tot_correct[j] += correct[j]
                bleu *= (float(correct[j]) + tiny) / (float(guess[j]) + small)
                b = bleu ** (1.0 / (j + 1))
                per_image[j].append(b * math.exp(1 - 1 / ratio) if ratio < 1 else b)
        scores = []
        bleu = 1.0
        ratio = (tot_test + tiny) / (tot_ref + small)
        for j in range(n):
            bleu *= float(tot_correct[j] + tiny) / (tot_guess[j] + small)
            b = bleu ** (1.0 / (j + 1))
            scores.append(b * math.exp(1 - 1 / ratio) if ratio < 1 else b)
        return scores, per_image

    def method(self):
        return 'Bleu'


def _lcs(a, b):
    if len(a) < len(b):
        a, b = b, a
    prev = [0] * (len(b) + 1)
    for x in a:
        cur = [0]
        for j, y in enumerate(b, 1):
            cur.append(prev[j - 1] + 1 if x == y else max(prev[j], cur[j - 1]))
        prev = cur
    return prev[len(b)]


class Rouge(object):
    beta = 1.2

    def calc_score(self, candidate, refs):
        assert len(candidate) == 1 and len(refs) > 0
        tc = candidate[0].split(' ')
        prec, rec = [], []
        for r in refs:
            tr = r.split(' ')
            lcs = _lcs(tr, tc)
            prec.append(lcs / float(len(tc)))
            rec.append(lcs / float(len(tr)))
        p, r = max(prec), max(rec)
        if p != 0 and r != 0:
            return ((1 + self.beta ** 2) * p * r) / float(r + self.beta ** 2 * p)
        return 0.0

    def compute_score(self, gts, res):
        assert list(gts.keys()) == list(res.keys())
        scores = np.array([self.calc_score(res[k], gts[k]) for k in gts])
        return float(np.mean(scores)), scores

    def method(self):
        return 'Rouge'


class Cider(object):
    def __init__(self, n=4, sigma=6.0):
        self._n, self._sigma = n, sigma

    def compute_score(self, gts, res):
        assert list(gts.keys()) == list(res.keys())
        n, sigma = self._n, self._sigma
        crefs = [[_ngrams(r, n)[1] for r in gts[k]] for k in gts]
        ctest = [_ngrams(res[k][0], n)[1] for k in gts]
        df = defaultdict(float)
        for refs in crefs:                                   # one count per image that contains the n-gram
            for ng in set(ng for ref in refs for ng in ref):
                df[ng] += 1
        log_ref_len = np.log(float(len(crefs)))

        def vec(cnts):
            v = [dict() for _ in range(n)]
            norm = [0.0] * n
            length = 0
            for ng, tf in cnts.items():
                k = len(ng) - 1
                v[k][ng] = float(tf) * (log_ref_len - np.log(max(1.0, df.get(ng, 0.0))))
                norm[k] += pow(v[k][ng], 2)
                if k == 1:                                   # reference quirk: length = number of bigrams
                    length += tf
            return v, [np.sqrt(x) for x in norm], length

        scores = []
        for test, refs in zip(ctest, crefs):
            vh, nh, lh = vec(test)
            score = np.zeros(n)
            for ref in refs:
                vr, nr, lr = vec(ref)
                delta = float(lh - lr)
                val = np.zeros(n)
                for k in range(n):
                    for ng, w in vh[k].items():
                        r = vr[k].get(ng, 0.0)
                        val[k] += min(w, r) * r
                    if nh[k] != 0 and nr[k] != 0:
                        val[k] /= (nh[k] * nr[k])
                    val[k] *= np.e ** (-(delta ** 2) / (2 * sigma ** 2))
                score += val
            scores.append(np.mean(score) / len(refs) * 10.0)
        return float(np.mean(np.array(scores))), np.array(scores)

    def method(self):
        return 'CIDEr'


def evaluate(gts, res):
    """gts {id: [tokenised reference strings]}, res {id: [tokenised hypothesis]} -> {metric: corpus score}."""
    keys = sorted(gts.keys())
    gts = {k: gts[k] for k in keys}
    res = {k: res[k] for k in keys}
    out = {}
    b, _ = Bleu(4).compute_score(gts, res)
    for i, v in enumerate(b):
        out['Bleu_%d' % (i + 1)] = float(v)
    out['ROUGE_L'] = Rouge().compute_score(gts, res)[0]
    out['CIDEr'] = Cider().compute_score(gts, res)[0]
    return out


def evaluate_captions(annotation_file, result_json):
    """COCO-format annotations ({'annotations': [{'image_id', 'caption'}, ...]}) and a results file
    ([{'image_id', 'caption'}, ...], infer_fn.py:166-184) -> {Bleu_1..4, ROUGE_L, CIDEr} over the result images."""
    with open(annotation_file) as f:
        ann = json.load(f)
    with open(result_json) as f:
        results = json.load(f)
    res = {}
    for r in results:
        res[r['image_id']] = [tokenize(r['caption'])]
    gts = defaultdict(list)
    for a in ann['annotations']:
        if a['image_id'] in res:
            gts[a['image_id']].append(tokenize(a['caption']))
    missing = [k for k in res if k not in gts]
    if missing:
        raise ValueError('%d result images have no reference captions (first: %r)' % (len(missing), missing[0]))
    return evaluate(dict(gts), res)
