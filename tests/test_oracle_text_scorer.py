"""Oracle restatements vs golden vectors captured from the reference's own Python
(oracle/make_golden.py).  CPU only."""
import json
import os

import numpy as np

from oracle import scorer_ref, text_ref


def _load(golden_dir, name):
    with open(os.path.join(golden_dir, name)) as f:
        return json.load(f)


def test_number_to_base(golden_dir):
    g = _load(golden_dir, 'text_golden.json')
    for n, b, exp in g['number_to_base']:
        assert text_ref.number_to_base(n, b) == exp


def test_id_to_caption(golden_dir):
    g = _load(golden_dir, 'text_golden.json')
    for case in g['id_to_caption']:
        got = text_ref.id_to_caption(np.array(case['ids']), case['token_type'], case['itow'],
                                     case['wtoi'], case['radix_base'])
        assert got == case['captions'], case['token_type']


def test_radix_table_and_batched_ids(golden_dir):
    g = _load(golden_dir, 'text_golden.json')
    for case in g['captions_to_batched_ids']:
        table = None
        if case['token_type'] == 'radix':
            table = text_ref.build_radix_wtoi(g['wtoi'], case['radix_base'])
            assert table == case['radix_wtoi']
        got = text_ref.captions_to_batched_ids(case['hypos'], case['token_type'], g['wtoi'], table)
        assert got.tolist() == case['ids']


def test_document_frequency(golden_dir):
    g = _load(golden_dir, 'scorer_golden.json')
    df = scorer_ref.build_df_from_refs(g['corpus_refs'])
    assert df['ref_len'] == g['ref_len']
    got = {' '.join(k): v for k, v in df['document_frequency'].items()}
    assert got == g['document_frequency']


def _scorer(g, weights):
    df = scorer_ref.build_df_from_refs(g['corpus_refs'])
    return scorer_ref.CaptionScorer(df, weights)


def test_ciderd_and_bleu_separately(golden_dir):
    g = _load(golden_dir, 'scorer_golden.json')
    c0 = g['cases'][0]
    sc = _scorer(g, c0['weights'])
    n = len(c0['refs'])
    cd = [sc.cider.score_one(h[0], c0['refs'][i % n]) for i, h in enumerate(c0['sample'])]
    np.testing.assert_allclose(cd, g['ciderD']['scores'], rtol=1e-12, atol=1e-12)
    bl = np.array([scorer_ref.bleu_sentence_scores(h[0], c0['refs'][i % n])
                   for i, h in enumerate(c0['sample'])])
    np.testing.assert_allclose(bl.T, np.array(g['bleu']['scores']), rtol=1e-12, atol=1e-300)


def test_caption_scorer_cases(golden_dir):
    g = _load(golden_dir, 'scorer_golden.json')
    for case in g['cases']:
        sc = _scorer(g, case['weights'])
        hyp, s_s, s_g = sc.get_hypo_scores(case['refs'], case['sample'], case['greedy'])
        assert hyp == case['sample']
        np.testing.assert_allclose(s_s, case['sc_sample'], rtol=1e-12, atol=1e-12)
        np.testing.assert_allclose(s_g, case['sc_greedy'], rtol=1e-12, atol=1e-12)
