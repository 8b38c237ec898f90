"""Document-frequency statistics for the SCST CIDEr-D reward, with the reference's output
format (common/scst/prepro_ngrams.py:24-98,:122-151): `{pattern}scst-words.p` =
pickle (protocol 2) of `{'document_frequency': {ngram tuple: count}, 'ref_len': n_images}`.
References keep their trailing ` <EOS>` for these statistics (prepro_ngrams.py:131) although
scoring-time references do not (manager_image_caption.py:395) -- reference quirk, kept.

CLI: python -m comic_amd.scst.prepro_ngrams --dataset_dir D --dataset_file_pattern P --split train
"""
from __future__ import annotations

import argparse
import os
import pickle
from collections import defaultdict


def precook(s, n=4, out=False):
    """n-gram counts (orders 1..n) of a whitespace-tokenised sentence."""
    words = s.split()
    counts = defaultdict(int)
    for k in range(1, n + 1):
        for i in range(len(words) - k + 1):
            counts[tuple(words[i:i + k])] += 1
    return counts


def cook_refs(refs, n=4):
    return [precook(ref, n) for ref in refs]


def create_crefs(refs):
    return [cook_refs(ref) for ref in refs]


def compute_doc_freq(crefs):
    """Number of images (reference groups) in which each n-gram occurs."""
    document_frequency = defaultdict(float)
    for refs in crefs:
        for ngram in set(ng for ref in refs for ng in ref):
            document_frequency[ngram] += 1
    return document_frequency


def get_ngrams(refs_words, wtoi=None, params=None):
    return compute_doc_freq(create_crefs(refs_words)), None, len(refs_words)


def build(caption_lines):
    """caption_lines: iterable of `relpath,<GO> w1 ... wN <EOS>` -> the pickle's dict."""
    groups = {}
    for line in caption_lines:
        line = line.strip()
        if not line:
            continue
        path, cap = line.split(',')[:2]
        groups.setdefault(path, []).append(cap.replace('<GO> ', ''))
    ngram_words, _, ref_len = get_ngrams(list(groups.values()))
    return {'document_frequency': dict(ngram_words), 'ref_len': ref_len}


def main(argv=None):
    p = argparse.ArgumentParser()
    p.add_argument('--dataset_dir', type=str, default='')
    p.add_argument('--dataset_file_pattern', type=str, default='mscoco_{}_w5_s20_include_restval')
    p.add_argument('--split', type=str, default='train')
    args = p.parse_args(argv)
    fp = os.path.join(args.dataset_dir, 'captions', args.dataset_file_pattern.format(args.split)) + '.txt'
    with open(fp, 'r') as f:
        out = build(f.readlines())
    dst = os.path.join(args.dataset_dir, 'captions', args.dataset_file_pattern.format('scst-words')) + '.p'
    with open(dst, 'wb') as f:
        pickle.dump(out, f, 2)
    print('INFO: wrote %s (%d n-grams, %d images)' % (dst, len(out['document_frequency']), out['ref_len']))


if __name__ == '__main__':
    main()
