#!/usr/bin/env python3
"""Generate tests/golden/*.json by running the REFERENCE's own Python.

Run only in the build container (needs /root/reference); the GPU box and the test
suite read the committed fixtures, never the reference.  Nothing from the reference
is copied into the repo: the fixtures hold inputs and expected outputs only.

  * CIDEr-D ............ imported as-is from common/scst/cider_ruotianluo
  * BLEU, captionScorer, prepro_ngrams ... py2-only syntax; a temp copy under a
    TemporaryDirectory gets a `lib2to3 -w -n` pass and is imported from there
  * pycocoevalcap Bleu / Rouge / Cider (corpus-level evaluation, eval.py) ... temp 2to3 copy, as BLEU
  * number_to_base, _baseN_arr_to_dec, id_to_caption, radix table,
    captions_to_batched_ids ... `ast`-extracted function bodies exec'd with stubs
    (their modules import tensorflow at top level).

Usage:  python oracle/make_golden.py
"""
import ast
import json
import os
import pickle
import random
import shutil
import subprocess
import sys
import tempfile
import types

import numpy as np

REF = '/root/reference'
OUT = os.path.join(os.path.dirname(os.path.abspath(__file__)), '..', 'tests', 'golden')


def extract(path, names, ns):
    """exec the FunctionDef / ClassDef nodes called `names` from `path` into ns."""
    tree = ast.parse(open(path).read())
    found = {}

    def visit(body):
        for node in body:
            if isinstance(node, ast.FunctionDef) and node.name in names:
                found[node.name] = node
            elif isinstance(node, ast.ClassDef):
                visit(node.body)
    visit(tree.body)
    for n in names:
        mod = ast.Module(body=[found[n]], type_ignores=[])
        exec(compile(mod, path, 'exec'), ns)
    return ns


def prep_scorer_tmp():
    tmp = tempfile.mkdtemp(prefix='comic_golden_')
    common = os.path.join(tmp, 'common')
    os.makedirs(os.path.join(common, 'coco_caption'))
    shutil.copytree(os.path.join(REF, 'common/coco_caption/pycocoevalcap'),
                    os.path.join(common, 'coco_caption/pycocoevalcap'))
    shutil.copytree(os.path.join(REF, 'common/scst'), os.path.join(common, 'scst'))
    targets = [os.path.join(common, 'coco_caption/pycocoevalcap/bleu'),
               os.path.join(common, 'scst/scorers.py'),
               os.path.join(common, 'scst/prepro_ngrams.py')]
    subprocess.run([sys.executable, '-m', 'lib2to3', '-w', '-n'] + targets,
                   check=True, stdout=subprocess.DEVNULL, stderr=subprocess.DEVNULL)
    return tmp, common


def make_scorer_golden():
    tmp, common = prep_scorer_tmp()
    try:
        sys.path.insert(0, os.path.join(common, 'scst'))
        sys.path.insert(0, common)
        import scorers                  # noqa: the reference's captionScorer (temp 2to3 copy)
        import prepro_ngrams            # noqa
        fake = json.load(open(os.path.join(
            REF, 'common/coco_caption/results/captions_val2014_fakecap_results.json')))
        caps = [d['caption'] for d in fake]
        rnd = random.Random(1234)
        # pseudo-images: 5 consecutive captions are the references of one image
        n_img = 120
        refs_all = [caps[i * 5:(i + 1) * 5] for i in range(n_img)]
        # df statistics exactly as prepro_ngrams.__main__ builds them (refs keep ' <EOS>')
        refs_eos = [[r + ' <EOS>' for r in rl] for rl in refs_all]
        df = prepro_ngrams.compute_doc_freq(prepro_ngrams.create_crefs(refs_eos))
        pkl = os.path.join(tmp, 'scst-words.p')
        with open(pkl, 'wb') as f:
            pickle.dump({'document_frequency': df, 'ref_len': len(refs_eos)}, f, 2)
        weights = dict(ciderD=1.0, bleu=[0.0, 0.0, 0.0, 2.0])       # train.py:141-146
        scorer = scorers.captionScorer(pkl, weights)
        scorer_b = scorers.captionScorer(pkl, dict(ciderD=0.5, bleu=[1.0, 0.5, 0.25, 2.0]))
        # The reference runs on Python 2, where a dict with dense small-int keys iterates
        # in ASCENDING key order; get_hypo_scores relies on that (scores[:num_greedy] are
        # the greedy ones).  Python 3 dicts iterate in insertion order (0, N, 1, N+1, ...),
        # which would pair scores with the wrong hypotheses, so present key-sorted dicts
        # to the metric objects to reproduce the py2 behaviour.
        for s_obj in (scorer, scorer_b):
            for m_obj in s_obj._scorer.values():
                def _sorted_call(gts, res, _orig=m_obj.compute_score):
                    return _orig(dict(sorted(gts.items())), dict(sorted(res.items())))
                m_obj.compute_score = _sorted_call
        pool = caps[600:]

        def perturb(s):
            w = s.split()
            op = rnd.randrange(6)
            if op == 0 and len(w) > 2:
                del w[rnd.randrange(len(w))]
            elif op == 1:
                w.insert(rnd.randrange(len(w) + 1), rnd.choice(['a', 'the', 'zebra', 'qwertyuiop']))
            elif op == 2:
                rnd.shuffle(w)
            elif op == 3:
                w = w[:max(1, len(w) // 2)]
            elif op == 4:
                w = w + w
            return ' '.join(w)

        cases = []
        for ci, (N, mult) in enumerate([(10, 7), (10, 1), (4, 3), (32, 7)]):
            idx = rnd.sample(range(n_img), N)
            refs = [refs_all[i] for i in idx]
            greedy = [[perturb(rnd.choice(r))] for r in refs]
            sample = []
            for m in range(mult):
                for i in range(N):
                    src = rnd.choice(refs[i]) if rnd.random() < 0.7 else rnd.choice(pool)
                    sample.append([perturb(src)])
            if ci == 0:                       # edge cases
                sample[0] = ['']              # empty hypothesis
                sample[1] = ['dog']           # single word
                sample[2] = ['xyzzy plugh frobnicate']   # all n-grams unseen (df = 0)
                sample[3] = [refs[3][0]]      # exact copy of a reference
                greedy[0] = ['']
            for sc, wname, wts in ((scorer, 'default', weights),
                                   (scorer_b, 'mixed', scorer_b.weights)):
                hyp, s_s, s_g = sc.get_hypo_scores(refs, sample, greedy)
                assert hyp == sample
                cases.append(dict(weights=wts, refs=refs, sample=sample, greedy=greedy,
                                  sc_sample=[float(x) for x in s_s],
                                  sc_greedy=[float(x) for x in s_g]))
        # separate CIDEr-D and BLEU outputs for the first case
        c0 = cases[0]
        gts = {i: c0['refs'][i % 10] for i in range(len(c0['sample']))}
        res = {i: c0['sample'][i] for i in range(len(c0['sample']))}
        from pyciderevalcap.ciderD.ciderD import CiderD
        cd_mean, cd_scores = CiderD(df=pkl).compute_score(gts, res)
        b_mean, b_scores = scorers.BleuSilent(4).compute_score(gts, res)
        golden = dict(
            note='generated by oracle/make_golden.py from the reference scorer',
            corpus_refs=refs_all,
            ref_len=len(refs_eos),
            document_frequency={' '.join(k): float(v) for k, v in df.items()},
            cases=cases,
            ciderD=dict(keys=list(range(len(c0['sample']))), mean=float(cd_mean),
                        scores=[float(x) for x in cd_scores]),
            bleu=dict(mean=[float(x) for x in b_mean],
                      scores=[[float(x) for x in row] for row in b_scores]))
        with open(os.path.join(OUT, 'scorer_golden.json'), 'w') as f:
            json.dump(golden, f)
        print('scorer_golden.json:', len(cases), 'cases,', len(df), 'df entries')
    finally:
        shutil.rmtree(tmp, ignore_errors=True)


def make_cocoeval_golden():
    """Corpus-level BLEU-1..4, ROUGE-L and CIDEr of the reference's pycocoevalcap scorers (eval.py:18-62 runs them
    after the Java PTB tokenizer; here on already tokenized strings) -> tests/golden/cocoeval_golden.json."""
    tmp = tempfile.mkdtemp(prefix='comic_golden_')
    try:
        pkg = os.path.join(tmp, 'pycocoevalcap')
        shutil.copytree(os.path.join(REF, 'common/coco_caption/pycocoevalcap'), pkg)
        subprocess.run([sys.executable, '-m', 'lib2to3', '-w', '-n'] +
                       [os.path.join(pkg, d) for d in ('bleu', 'rouge', 'cider')],
                       check=True, stdout=subprocess.DEVNULL, stderr=subprocess.DEVNULL)
        sys.path.insert(0, tmp)
        for m in [k for k in sys.modules if k == 'pycocoevalcap' or k.startswith('pycocoevalcap.')]:
            sys.modules.pop(m)
        from pycocoevalcap.bleu import bleu as bleu_mod        # noqa: temp 2to3 copies of the reference
        from pycocoevalcap.rouge import rouge as rouge_mod     # noqa
        from pycocoevalcap.cider import cider as cider_mod     # noqa
        fake = json.load(open(os.path.join(REF, 'common/coco_caption/results/captions_val2014_fakecap_results.json')))
        caps = [d['caption'] for d in fake]
        rnd = random.Random(99)
        n_img = 80
        gts = {i: caps[i * 5:(i + 1) * 5] for i in range(n_img)}
        res = {}
        for i in range(n_img):
            w = rnd.choice(gts[i]).split() if rnd.random() < 0.8 else rnd.choice(caps[500:]).split()
            op = rnd.randrange(5)
            if op == 0 and len(w) > 2:
                del w[rnd.randrange(len(w))]
            elif op == 1:
                w.insert(rnd.randrange(len(w) + 1), rnd.choice(['a', 'the', 'zebra']))
            elif op == 2:
                rnd.shuffle(w)
            elif op == 3:
                w = w[:max(1, len(w) // 2)]
            res[i] = [' '.join(w)]
        res[0] = [gts[0][0]]            # exact copy of a reference
        res[1] = ['dog']                # single word
        res[2] = ['xyzzy plugh']        # nothing in common
        import io, contextlib
        with contextlib.redirect_stdout(io.StringIO()):
            b_mean, b_scores = bleu_mod.Bleu(4).compute_score(gts, res)
        r_mean, r_scores = rouge_mod.Rouge().compute_score(gts, res)
        c_mean, c_scores = cider_mod.Cider().compute_score(gts, res)
        golden = dict(note='generated by oracle/make_golden.py from the reference pycocoevalcap (bleu, rouge, cider)',
                      gts={str(k): v for k, v in gts.items()}, res={str(k): v for k, v in res.items()},
                      bleu=dict(mean=[float(x) for x in b_mean], scores=[[float(x) for x in row] for row in b_scores]),
                      rouge=dict(mean=float(r_mean), scores=[float(x) for x in r_scores]),
                      cider=dict(mean=float(c_mean), scores=[float(x) for x in c_scores]))
        with open(os.path.join(OUT, 'cocoeval_golden.json'), 'w') as f:
            json.dump(golden, f)
        print('cocoeval_golden.json: %d images, Bleu_4 %.4f ROUGE_L %.4f CIDEr %.4f' % (n_img, b_mean[3], r_mean, c_mean))
    finally:
        shutil.rmtree(tmp, ignore_errors=True)


def make_text_golden():
    ns = {'np': np}
    extract(os.path.join(REF, 'common/ops.py'), ['number_to_base'], ns)
    ns['ops'] = types.SimpleNamespace(number_to_base=ns['number_to_base'])
    extract(os.path.join(REF, 'src/infer_fn.py'), ['_baseN_arr_to_dec', 'id_to_caption'], ns)
    extract(os.path.join(REF, 'common/inputs/manager_image_caption.py'),
            ['captions_to_batched_ids'], ns)
    rnd = random.Random(7)
    # vocab in prepro_base.build_vocab layout: <PAD>=-1, words, <UNK>, <GO>, <EOS>
    words = ['w%d' % i for i in range(300)]
    wtoi = {'<PAD>': -1}
    for i, w in enumerate(words):
        wtoi[w] = i
    for tok in ('<UNK>', '<GO>', '<EOS>'):
        wtoi[tok] = len(wtoi) - 1
    itow = {str(v): k for k, v in wtoi.items()}
    out = dict(wtoi=wtoi, itow=itow)
    out['number_to_base'] = [[n, b, ns['number_to_base'](n, b)] for n, b in
                             [(0, 256), (1, 256), (255, 256), (256, 256), (9999, 256), (65535, 256),
                              (65536, 256), (302, 256), (302, 16), (7, 2), (-300, 256), (123456, 10)]]
    id_cases = []
    for token_type, base in (('radix', 256), ('radix', 16), ('word', 0), ('char', 0)):
        if token_type == 'char':
            import string
            ctoi, itoc = {}, {}
            idx = -1
            for ch in ['<PAD>', ' '] + list(string.digits + string.ascii_lowercase):
                ctoi[ch] = idx; itoc[str(idx)] = ch; idx += 1
            ctoi['<GO>'] = len(ctoi); ctoi['<EOS>'] = len(ctoi)
            itoc[str(len(itoc))] = '<GO>'; itoc[str(len(itoc))] = '<EOS>'
            cfg = types.SimpleNamespace(token_type='char', radix_base=0, itow=itoc, wtoi=ctoi)
            hi = len(itoc) - 1
        else:
            cfg = types.SimpleNamespace(token_type=token_type, radix_base=base, itow=itow, wtoi=wtoi)
            hi = base + 2 if token_type == 'radix' else len(itow) - 1
        rows = []
        for _ in range(12):
            L = rnd.randrange(1, 24)
            row = [rnd.randrange(-1, hi) for _ in range(L)]
            if token_type == 'char':        # id 37 is unassigned in the reference's char table
                row = [r for r in row if r != 37] or [0]
            rows.append(row)
        if token_type == 'radix' and base == 256:
            rows.append([256, 0, 5, 1, 3, 257, 0, 7, 2])          # post-EOS ids kept, odd tail dropped
            rows.append([256, 1, 200, 257, 257, 257])              # word id >= vocab -> skipped
            rows.append([257])
        L = max(len(r) for r in rows)
        arr = np.full((len(rows), L), -1, np.int32)
        for i, r in enumerate(rows):
            arr[i, :len(r)] = r
        caps = ns['id_to_caption'](arr, cfg)
        id_cases.append(dict(token_type=token_type, radix_base=base, ids=arr.tolist(), captions=caps,
                             itow=cfg.itow, wtoi=cfg.wtoi))
    out['id_to_caption'] = id_cases
    # radix table: manager_image_caption.py:240-254 is inline in __init__, so replay it
    # through the reference's number_to_base and record the table
    b2c = []
    for base in (256, 16):
        max_word_len = len(ns['number_to_base'](len(wtoi), base))
        table = {}
        for k in wtoi:
            if k == '<GO>':
                idx = [base]
            elif k == '<EOS>':
                idx = [base + 1]
            elif k == '<PAD>':
                idx = [-1]
            else:
                idx = ns['number_to_base'](wtoi[k], base)
                idx = [0] * (max_word_len - len(idx)) + idx
            table[k] = idx
        hypos = [[' '.join(rnd.choice(words + ['notaword']) for _ in range(rnd.randrange(0, 9)))]
                 for _ in range(9)]
        for tt in ('radix', 'word'):
            self_ = types.SimpleNamespace(
                config=types.SimpleNamespace(token_type=tt, wtoi=wtoi), radix_wtoi=table)
            ids = ns['captions_to_batched_ids'](self_, hypos)
            b2c.append(dict(token_type=tt, radix_base=base, hypos=hypos, ids=np.asarray(ids).tolist(),
                            radix_wtoi=table if tt == 'radix' else None))
    out['captions_to_batched_ids'] = b2c
    with open(os.path.join(OUT, 'text_golden.json'), 'w') as f:
        json.dump(out, f)
    print('text_golden.json written')


if __name__ == '__main__':
    os.makedirs(OUT, exist_ok=True)
    if len(sys.argv) > 1 and sys.argv[1] == 'cocoeval':      # only the corpus-level scorer fixture
        make_cocoeval_golden()
        sys.exit(0)
    make_text_golden()
    make_scorer_golden()
    make_cocoeval_golden()
