"""CPU-side tests of the PRODUCT's host logic (no GPU compute): C-ABI exports, golden
vectors for the token helpers and the C++ reward scorer, config.pkl compatibility, input
managers, CLI run-directory naming, checkpoint restore logic, schedules."""
import json
import os
import pickle
import re
import sys
import types

import numpy as np
import pytest

import comic_amd._lib as L
from comic_amd import checkpoint as ckpt, configuration as conf, decoder as cdec, nets, ops, optim

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _golden(golden_dir, name):
    with open(os.path.join(golden_dir, name)) as f:
        return json.load(f)


# ------------------------------------------------------------------ C-ABI ---------------
def test_library_exports_every_declared_symbol():
    lib = L.load()
    header = open(os.path.join(ROOT, 'include', 'comic_hip.h')).read()
    header = re.sub(r'/\*.*?\*/', '', header, flags=re.S)
    declared = set(re.findall(r'\b(comic_[a-z0-9_]+)\s*\(', header))
    assert declared, 'no declarations parsed'
    for name in sorted(declared):
        assert hasattr(lib, name), 'libcomic_hip.so does not export %s' % name
    assert set(L.EXPORTED_SYMBOLS) == declared, set(L.EXPORTED_SYMBOLS) ^ declared
    assert lib.comic_abi_version() == 1


def test_library_reads_no_environment():
    """no hidden process state behind the boundary -- the executors' A/B switches travel in
    comic_decoder_desc.flags / comic_cnn_op.min_lds, and libcomic_hip.so imports no getenv."""
    import subprocess
    und = subprocess.run(['nm', '-D', '--undefined-only', L.LIB_PATH], capture_output=True, text=True, check=True).stdout
    assert 'getenv' not in und, [l for l in und.splitlines() if 'getenv' in l]


def test_decoder_flags_follow_the_python_environment(monkeypatch):
    for k in ('COMIC_PERSIST', 'COMIC_PERSIST_BWD', 'COMIC_FUSED_STEP', 'COMIC_SPLIT_ATTN_BWD', 'COMIC_GRAD_LANES',
              'COMIC_SPLIT3', 'COMIC_PERSIST_STAMPS'):
        monkeypatch.delenv(k, raising=False)
    assert L.decoder_flags_from_env() == 0
    monkeypatch.setenv('COMIC_PERSIST', '0')
    monkeypatch.setenv('COMIC_SPLIT3', '0')
    assert L.decoder_flags_from_env() == L.DEC_NO_PERSIST | L.DEC_EXACT_GEMM
    monkeypatch.setenv('COMIC_PERSIST', '1')
    monkeypatch.setenv('COMIC_PERSIST_STAMPS', '1')
    assert L.decoder_flags_from_env() == L.DEC_EXACT_GEMM | L.DEC_STAMPS
    header = open(os.path.join(ROOT, 'include', 'comic_hip.h')).read()
    for name, val in (('NO_PERSIST', L.DEC_NO_PERSIST), ('NO_PERSIST_BWD', L.DEC_NO_PERSIST_BWD),
                      ('NO_FUSED_STEP', L.DEC_NO_FUSED_STEP), ('NO_SPLIT_ATTN_BWD', L.DEC_NO_SPLIT_ATTN_BWD),
                      ('ONE_LANE', L.DEC_ONE_LANE), ('EXACT_GEMM', L.DEC_EXACT_GEMM), ('STAMPS', L.DEC_STAMPS)):
        assert int(re.search(r'#define COMIC_DEC_%s (\d+)u' % name, header).group(1)) == val


def test_struct_layouts_match_header():
    import ctypes as C
    assert C.sizeof(L.CnnOp) == 26 * 4                    # ... flags, min_lds
    assert C.sizeof(L.AttnDesc) == 8 * 4
    assert C.sizeof(L.DecoderDesc) == 16 * 4 + 4 * 4 + 4 + 4 + 4  # ... map_loss_scale, flags, length_penalty_weight, cell
    assert C.sizeof(L.DecoderParams) == 18 * 8               # ... emb, cell_ln, K_c, b_c, status
    assert C.sizeof(L.ConvWeight) == 4 * 8                # w, scale, shift, w_frag


def test_conv_variant_ids_match_header():
    """The kernel-variant id space of comic_cnn_op.tile: the Python mirror agrees with include/comic_hip.h, and the
    always-eligible (im2col) and may-refuse (patch-resident) ids partition 1..COMIC_CONV_TILES."""
    header = open(os.path.join(ROOT, 'include', 'comic_hip.h')).read()
    assert int(re.search(r'#define COMIC_CONV_TILES (\d+)', header).group(1)) == L.CONV_TILES
    im2col = [t for t in range(1, L.CONV_TILES + 1) if L.is_im2col_tile(t)]
    patch = [t for t in range(1, L.CONV_TILES + 1) if not L.is_im2col_tile(t)]
    assert im2col == list(range(1, 13)) + list(range(26, 48))
    # may-refuse ids (incl. the weight-stationary and image-resident kernels and the walk forms 56..58 of shared-input groups)
    assert patch == list(range(13, 26)) + list(range(48, 54)) + [L.WS_TILE, L.IMG_TILE] + [56, 57, 58, 59, 60, 61]
    assert int(re.search(r'#define COMIC_WS_TILE (\d+)', header).group(1)) == L.WS_TILE
    assert int(re.search(r'#define COMIC_IMG_TILE (\d+)', header).group(1)) == L.IMG_TILE
    assert int(re.search(r'#define COMIC_OP_POOLED_SRC (\d+)', header).group(1)) == L.OP_POOLED_SRC
    assert int(re.search(r'#define COMIC_CHAIN_TILE (\d+)', header).group(1)) == L.CHAIN_TILE > L.CONV_TILES   # never an autotune candidate
    assert int(re.search(r'#define COMIC_OP_CHAIN_LINK (\d+)', header).group(1)) == L.OP_CHAIN_LINK
    assert int(re.search(r'#define COMIC_OP_CHAIN_KEEP (\d+)', header).group(1)) == L.OP_CHAIN_KEEP


def test_missing_library_fails_loudly(monkeypatch):
    monkeypatch.setattr(L, '_lib', None)
    monkeypatch.setattr(L, 'LIB_PATH', '/nonexistent/libcomic_hip.so')
    with pytest.raises(L.ComicHipError, match='no CPU fallback'):
        L.load()


# ------------------------------------------------------------------ token helpers -------
def test_product_text_helpers_match_reference_goldens(golden_dir):
    g = _golden(golden_dir, 'text_golden.json')
    for n, b, exp in g['number_to_base']:
        assert ops.number_to_base(n, b) == exp
    for case in g['id_to_caption']:
        cfg = types.SimpleNamespace(token_type=case['token_type'], radix_base=case['radix_base'], itow=case['itow'],
                                    wtoi=case['wtoi'])
        assert ops.id_to_caption(np.array(case['ids']), cfg) == case['captions']
    for case in g['captions_to_batched_ids']:
        cfg = types.SimpleNamespace(token_type=case['token_type'], wtoi=g['wtoi'])
        table = ops.build_radix_wtoi(g['wtoi'], case['radix_base']) if case['token_type'] == 'radix' else None
        if table:
            assert table == case['radix_wtoi']
        assert ops.captions_to_batched_ids(case['hypos'], cfg, table).tolist() == case['ids']


def test_cpp_scorer_matches_reference_goldens(golden_dir):
    """The C++ scorer behind the C-ABI (host code, runs without a GPU) vs outputs captured
    from the reference's captionScorer."""
    from comic_amd.scst.scorers import captionScorer
    g = _golden(golden_dir, 'scorer_golden.json')
    df = dict(document_frequency={tuple(k.split(' ')): v for k, v in g['document_frequency'].items()},
              ref_len=g['ref_len'])
    for case in g['cases']:
        sc = captionScorer(df, case['weights'], n_threads=4)
        hyp, s_s, s_g = sc.get_hypo_scores(case['refs'], case['sample'], case['greedy'])
        assert hyp == case['sample']
        np.testing.assert_allclose(s_s, case['sc_sample'], rtol=1e-12, atol=1e-12)
        np.testing.assert_allclose(s_g, case['sc_greedy'], rtol=1e-12, atol=1e-12)
    sc = captionScorer(df, g['cases'][0]['weights'])
    c0 = g['cases'][0]
    n = len(c0['refs'])
    cider, bleu = sc._score([h[0] for h in c0['sample']], [c0['refs'][i % n] for i in range(len(c0['sample']))])
    np.testing.assert_allclose(cider, g['ciderD']['scores'], rtol=1e-12, atol=1e-12)
    np.testing.assert_allclose(bleu.T, np.array(g['bleu']['scores']), rtol=1e-12, atol=1e-300)
    # best_hypo_only branch (scorers.py:136-158)
    final, best, greedy = sc.get_hypo_scores(c0['refs'], c0['sample'], c0['greedy'], best_hypo_only=True)
    assert len(final) == n and best.shape == (n,) and greedy.shape == (n,)


def test_scorer_reads_reference_pickle(tmp_path, golden_dir):
    from comic_amd.scst.scorers import captionScorer
    g = _golden(golden_dir, 'scorer_golden.json')
    df = {tuple(k.split(' ')): v for k, v in g['document_frequency'].items()}
    p = tmp_path / 'scst-words.p'
    with open(p, 'wb') as f:
        pickle.dump({'document_frequency': df, 'ref_len': g['ref_len']}, f, 2)      # prepro_ngrams.py:149-151
    sc = captionScorer(str(p), dict(ciderD=1.0, bleu=[0, 0, 0, 2]))
    c0 = g['cases'][0]
    _, s_s, _ = sc.get_hypo_scores(c0['refs'], c0['sample'], c0['greedy'])
    np.testing.assert_allclose(s_s, c0['sc_sample'], rtol=1e-12, atol=1e-12)


# ------------------------------------------------------------------ config --------------
def test_config_pickle_roundtrip_and_py2_layout(tmp_path):
    c = conf.Config(log_path=str(tmp_path), token_type='radix', radix_base=256, cnn_input_size=[224, 224],
                    cnn_fm_projection=None, scst_weight_bleu=[0.0, 0.0, 0.0, 2.0], rand_seed=48964896)
    c.save_config_to_file()
    raw = open(tmp_path / 'config.pkl', 'rb').read()
    assert raw[:2] == b'\x80\x02'                                   # protocol 2 (configuration.py:34-35)
    d = pickle.loads(raw)
    assert isinstance(d, dict) and d['radix_base'] == 256            # a plain dict, not the object
    c2 = conf.load_config(str(tmp_path / 'config.pkl'))
    assert c2.__dict__ == c.__dict__
    txt = [f for f in os.listdir(tmp_path) if f.startswith('config___')]
    assert txt and b'\r\n' in open(tmp_path / txt[0], 'rb').read()
    with pytest.raises(SystemExit):
        c.overwrite_safety_check(False)


def test_cli_run_dir_naming_and_mode_overrides(tmp_path):
    import importlib.util
    spec = importlib.util.spec_from_file_location('train_cli', os.path.join(ROOT, 'src', 'train.py'))
    train = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(train)
    args = train.create_parser().parse_args(['--log_root', str(tmp_path), '--cnn_fm_projection', 'none', '--run', '2'])
    kw, fn, overwrite = train.build_kwargs(args)
    assert os.path.basename(kw['log_path']) == 'radix_b256_add_LN_softmax_h8_non_lstm_run_02'   # train.py:214-230
    assert kw['cnn_fm_projection'] is None and kw['rand_seed'] == 88888888 and fn == 'train_fn'
    assert kw['dropout_rnn_in'] == 0.35 and kw['l2_decay'] == 1e-5 and kw['max_saves'] == 12     # train.py:281-300
    assert kw['cnn_input_size'] == [224, 224] and not overwrite
    # scst mode overrides (train.py:253-270)
    base = os.path.join(str(tmp_path), 'mscoco', 'radix_b256_add_LN_softmax_h8_tie_lstm_cnnFT_run_01')
    os.makedirs(base)
    args = train.create_parser().parse_args(['--log_root', str(tmp_path), '--train_mode', 'scst'])
    kw, fn, _ = train.build_kwargs(args)
    assert fn == 'train_fn_scst' and kw['batch_size_train'] == 10 and kw['lr_start'] == 1e-3 and kw['max_epoch'] == 10
    assert kw['scst_weight_bleu'] == [0.0, 0.0, 0.0, 2.0] and kw['checkpoint_path'] == base
    assert os.path.basename(kw['log_path']) == \
        'radix_b256_add_LN_softmax_h8_tie_lstm_cnnFT_SCST_beam_7_CrD_1.0_B1_0.0_B4_2.0_run_01'
    # `type=bool` quirk: any non-empty string is True (train.py:45)
    assert train.create_parser().parse_args(['--legacy', 'False']).legacy is True


# ------------------------------------------------------------------ inputs --------------
def test_input_managers_on_tiny_dataset(tmp_path):
    from tests import tiny_dataset
    from comic_amd import inputs
    ds = tiny_dataset.make(str(tmp_path / 'mscoco'), n_train=8, n_valid=2, n_test=2)
    c = conf.Config(dataset_dir=ds, dataset_file_pattern='mscoco_{}_w5_s20_include_restval', cnn_name='inception_v3',
                    cnn_input_size=[224, 224], cnn_input_augment=True, batch_size_train=4, batch_size_eval=2,
                    max_epoch=3, rand_seed=1, token_type='radix', radix_base=256)
    man = inputs.InputManager_Radix(c)
    assert c.split_sizes == {'train': 40, 'valid': 2} and c.max_step == int(40 / 4 * 3)    # :132,:141
    assert man.buckets == [11, 13, 15]                      # 23-word vocab -> 1 radix digit per word
    ims, caps = next(man.batch_train)
    assert ims.shape == (4, 224, 224, 3) and ims.dtype == np.float32 and -1.0 <= ims.min() and ims.max() <= 1.0
    assert caps.dtype == np.int32 and (caps[:, 0] == 256).all()
    assert all(257 in row for row in caps.tolist()) and caps.min() >= -1
    ims_e, _ = next(man.batch_eval)
    assert ims_e.shape == (2, 224, 224, 3)
    c2 = conf.Config(**{**c.__dict__, 'batch_size_train': 2, 'scst_beam_size': 3})
    scst = inputs.InputManager_SCST(c2)
    imgs, refs = next(scst.batch_train)
    assert imgs.shape[0] == 2 and len(refs) == 2 and all(len(r) == 5 and '<GO>' not in r[0] for r in refs)
    ids = scst.captions_to_batched_ids([['a man'], ['dog notaword cat']])
    assert ids.tolist() == [[256, 0, 1, 257, -1], [256, 2, c.wtoi['<UNK>'], 3, 257]]


def test_loader_prefetch_and_deterministic_augmentation(tmp_path):
    """The batches come from a prefetch thread (order = generator order, producer exceptions re-raised at next()),
    and the augmentation stream is drawn in the producer thread: two managers with the same seed yield the same
    augmented pixels whatever the decode threads do."""
    from tests import tiny_dataset
    from comic_amd import inputs
    pf = inputs.Prefetch(iter(range(10)), depth=3)
    assert list(pf) == list(range(10))
    with pytest.raises(StopIteration):
        next(pf)

    def boom():
        yield 1
        raise ValueError('decode failed')
    pf = inputs.Prefetch(boom(), depth=2)
    assert next(pf) == 1
    with pytest.raises(ValueError, match='decode failed'):
        next(pf)
    ds = tiny_dataset.make(str(tmp_path / 'mscoco'), n_train=8, n_valid=2, n_test=2)
    kw = dict(dataset_dir=ds, dataset_file_pattern='mscoco_{}_w5_s20_include_restval', cnn_name='inception_v3',
              cnn_input_size=[224, 224], cnn_input_augment=True, batch_size_train=4, batch_size_eval=2, max_epoch=3,
              rand_seed=7, token_type='radix', radix_base=256)
    a = inputs.InputManager_Radix(conf.Config(loader_threads=8, **kw))
    b = inputs.InputManager_Radix(conf.Config(loader_threads=1, loader_prefetch=1, **kw))
    assert isinstance(a.batch_train, inputs.Prefetch)
    for _ in range(3):
        (ia, ca), (ib, cb) = next(a.batch_train), next(b.batch_train)
        np.testing.assert_array_equal(ia, ib)
        np.testing.assert_array_equal(ca, cb)
    # end of a stage: the endless training generators stop instead of staying parked on a full queue
    threads = [a.batch_train._t, b.batch_train._t]
    a.close(); b.close()
    assert not any(t.is_alive() for t in threads)
    with pytest.raises(StopIteration):
        next(a.batch_train)


def test_native_coco_metrics_match_reference_goldens(golden_dir, tmp_path):
    """Corpus BLEU-1..4, ROUGE-L and CIDEr (comic_amd.coco_eval) against the reference's pycocoevalcap scorers on
    the committed fixture (oracle/make_golden.py cocoeval), and the file-level entry point of infer.py."""
    from comic_amd import coco_eval
    g = _golden(golden_dir, 'cocoeval_golden.json')
    keys = sorted(g['gts'], key=int)
    gts = {k: g['gts'][k] for k in keys}
    res = {k: g['res'][k] for k in keys}
    b_mean, b_scores = coco_eval.Bleu(4).compute_score(gts, res)
    np.testing.assert_allclose(b_mean, g['bleu']['mean'], rtol=1e-12)
    np.testing.assert_allclose(b_scores, g['bleu']['scores'], rtol=1e-12)
    r_mean, r_scores = coco_eval.Rouge().compute_score(gts, res)
    np.testing.assert_allclose(r_mean, g['rouge']['mean'], rtol=1e-12)
    np.testing.assert_allclose(r_scores, g['rouge']['scores'], rtol=1e-12)
    c_mean, c_scores = coco_eval.Cider().compute_score(gts, res)
    np.testing.assert_allclose(c_mean, g['cider']['mean'], rtol=1e-12)
    np.testing.assert_allclose(c_scores, g['cider']['scores'], rtol=1e-12, atol=1e-15)
    # tokenizer stand-in: lower case, punctuation dropped, clitics kept with their word
    assert coco_eval.tokenize("A man, riding a horse -- it's BIG!") == "a man riding a horse it's big"
    ann = dict(annotations=[dict(image_id=int(k), caption=c.capitalize() + ' .') for k in keys for c in gts[k]])
    results = [dict(image_id=int(k), caption=res[k][0]) for k in keys]
    fa, fr = tmp_path / 'ann.json', tmp_path / 'res.json'
    fa.write_text(json.dumps(ann)); fr.write_text(json.dumps(results))
    out = coco_eval.evaluate_captions(str(fa), str(fr))
    assert abs(out['Bleu_4'] - g['bleu']['mean'][3]) < 1e-12 and abs(out['CIDEr'] - g['cider']['mean']) < 1e-12
    assert abs(out['ROUGE_L'] - g['rouge']['mean']) < 1e-12


def test_tf1_bilinear_resize_matches_definition():
    from comic_amd.inputs import resize_bilinear_tf1
    img = np.arange(2 * 3 * 1, dtype=np.float32).reshape(2, 3, 1)
    out = resize_bilinear_tf1(img, 4, 6)
    # src = dst * in/out, no half-pixel offset: even indices reproduce the source pixels
    np.testing.assert_allclose(out[::2, ::2, 0], img[..., 0])
    np.testing.assert_allclose(out[0, 1, 0], 0.5)
    np.testing.assert_allclose(out[3, 5, 0], img[1, 2, 0])      # clamped at the border


# ------------------------------------------------------------------ plan / spec ---------
def test_cnn_plan_known_answers():
    plan = nets.CnnPlan('inception_v3', (224, 224))
    assert sum(1 for o in plan.ops if o['kind'] in (0, 1)) == 94
    shapes = plan.param_shapes()
    assert sum(int(np.prod(s)) for s in shapes.values()) == 21802784       # inception_v3_test.py:125-133
    assert plan.macs == 2835873120
    assert plan.buffers[plan.fm][:3] == (5, 5, 2048) and plan.buffers[plan.pooled][:3] == (1, 1, 2048)
    p299 = nets.CnnPlan('inception_v3', (299, 299))
    assert p299.buffers[p299.fm][:3] == (8, 8, 2048)
    assert p299.buffers[p299.end_points['Mixed_6e']][:3] == (17, 17, 768)  # inception_v3_test.py:93-123
    with pytest.raises(NotImplementedError):
        nets.CnnPlan('vgg_16')


def test_decoder_spec_from_config_and_errors():
    c = types.SimpleNamespace(rnn_name='LSTM', attn_alignment_method='add_LN', attn_probability_fn='softmax',
                              token_type='radix', radix_base=256, rnn_size=512, rnn_word_size=256, attn_num_heads=8,
                              cnn_fm_projection='tied', attn_context_layer=False, rnn_init_method='first_input',
                              attn_keep_prob=0.9)
    s = cdec.DecoderSpec.from_config(c, (25, 2048), 2048)
    assert (s.V, s.start_id, s.end_id, s.A, s.Cv) == (258, 256, 257, 512, 512)
    assert sum(int(np.prod(v)) if v else 1 for v in s.param_shapes().values()) == 5707011
    c.attn_alignment_method = 'add'                 # accepted by the CLI, rejected by the model (model_base.py:133-138)
    with pytest.raises(ValueError):
        cdec.DecoderSpec.from_config(c, (25, 2048), 2048)
    c.attn_alignment_method = 'add_LN'
    c.cnn_fm_projection, c.token_type = None, 'word'
    c.itow = {str(i): 'w%d' % i for i in range(-1, 99)}
    c.wtoi = {'<GO>': 97, '<EOS>': 98}
    s = cdec.DecoderSpec.from_config(c, (25, 2048), 2048)
    assert (s.V, s.A, s.Cv) == (100, 2048, 2048)    # softmax_size counts <PAD> (model_base.py:44-45)


def test_checkpoint_restore_three_way_logic(tmp_path):
    spec = cdec.DecoderSpec(D=64, E=32, C=64, Cg=64, M=4)
    plan_names = ['InceptionV3/Conv2d_1a_3x3/weights']
    cnn = {plan_names[0]: np.ones((3, 3, 3, 32), np.float32)}
    dec = cdec.init_params(spec, 0)
    path = ckpt.save(str(tmp_path / 'model'), 7, cnn, spec, dec, {'optimise/caption/adam_m': np.zeros(3)})
    assert path.endswith('model-7.npz') and ckpt.latest_checkpoint(str(tmp_path)) == path
    names = ckpt.decoder_var_names(spec)
    assert names['W_init'] == 'Model/decoder/rnn_decoder/rnn_init_input/projection/weight'
    assert names['K'].endswith('basic_lstm_cell/kernel') and names['emb'].endswith('embedding_map')
    c2, d2, extra = ckpt.restore(path, plan_names, spec, resume_training=True)
    assert set(d2) == set(dec) and 'global_step' in extra and int(extra['global_step']) == 7
    _, _, extra = ckpt.restore(path, plan_names, spec, resume_training=False)
    assert extra == {}
    # slim-style checkpoint (no Model/ prefix, no decoder): CNN only (model_base.py:468-482)
    np.savez(tmp_path / 'slim.npz', **{plan_names[0]: 2 * cnn[plan_names[0]]})
    c3, d3, _ = ckpt.restore(str(tmp_path / 'slim.npz'), plan_names, spec)
    assert d3 is None and c3[plan_names[0]].max() == 2


def test_checkpoint_variable_list_equals_the_reference_scopes(golden_dir, tmp_path):
    """f1: the variables checkpoint.py writes / expects for a TF bundle, name for name and shape for shape, against the
    list oracle/ref_var_names.py derives statically from the reference's scopes (tests/golden/ref_var_names.json; the
    [TF-1.9] scoping rules it applies are stated in its header) -- COMIC-256 on Inception-V1 and V3, the word baseline,
    LN_LSTM, GRU, the legacy encoder head, project_hidden, dot + context layer + independent values."""
    import json
    from comic_amd import encoder_head
    gold = json.load(open(os.path.join(golden_dir, 'ref_var_names.json')))['configs']
    specs = {
        'comic256_v1': cdec.DecoderSpec(C=832, Cg=1024, M=196),
        'comic256_v3': cdec.DecoderSpec(),
        'word_baseline': cdec.DecoderSpec(V=25599, H=1, fm_projection=None, token_type='word'),
        'ln_lstm': cdec.DecoderSpec(rnn_name='LN_LSTM'),
        'gru': cdec.DecoderSpec(rnn_name='GRU'),
        'legacy_v1': cdec.DecoderSpec(C=832, Cg=1024, M=196),
        'project_hidden': cdec.DecoderSpec(init_method='project_hidden'),
        'dot_context_independent': cdec.DecoderSpec(method='dot', context_layer=True, fm_projection='independent'),
    }
    assert set(specs) == set(gold)
    for name, spec in specs.items():
        names, shapes = ckpt.decoder_var_names(spec), spec.param_shapes()
        mine = {names[k]: list(shapes[k]) for k in names}
        if name == 'legacy_v1':
            mine.update({encoder_head.TF_NAMES['ln_beta']: [1024], encoder_head.TF_NAMES['ln_gamma']: [1024],
                         encoder_head.TF_NAMES['W']: [1024, 1024]})
        assert mine == gold[name], (name, sorted(set(mine) ^ set(gold[name])))
    # a checkpoint of this package's earlier rounds (flat names) still restores
    spec = cdec.DecoderSpec(D=64, E=32, C=64, Cg=64, M=4)
    dec = cdec.init_params(spec, 0)
    old = ckpt.decoder_var_names(spec, legacy_flat=True)
    assert old['W_q'] == 'Model/decoder/rnn_decoder/multi_add_attention/query_layer/kernel'
    arrays = {old[k]: v for k, v in dec.items()}
    arrays.update({ckpt.ADAM_SCOPE + old['W_q'] + '/Adam': np.ones((64, 64), np.float32)})
    np.savez(tmp_path / 'model-3.npz', global_step=np.asarray(3, np.int32), **arrays)
    _, d2, extra = ckpt.restore(str(tmp_path / 'model-3.npz'), [], spec, resume_training=True)
    assert d2 is not None and np.array_equal(d2['W_q'], dec['W_q'])
    assert ckpt.ADAM_SCOPE + ckpt.decoder_var_names(spec)['W_q'] + '/Adam' in extra


def test_schedules():
    assert np.isclose(optim.cosine_lr(0, 100, 1e-2, 1e-5), 1e-2)
    assert np.isclose(optim.cosine_lr(100, 100, 1e-2, 1e-5), 1e-5)
    assert np.isclose(optim.cosine_lr(50, 100, 1e-2, 1e-5), (1e-2 - 1e-5) / 2 + 1e-5)
    lr = 1e-3
    for epoch in range(1, 13):
        lr = optim.legacy_lr_reduce(lr, epoch, 2e-4, 4)
    assert np.isclose(lr, 2e-4)                       # 1e-3 -> 5e-4 -> 2.5e-4 -> clamp 2e-4


def test_prepro_ngrams_matches_reference_df(golden_dir):
    """Product df builder vs the document_frequency captured from the reference's
    prepro_ngrams.compute_doc_freq (refs keep ' <EOS>')."""
    from comic_amd.scst import prepro_ngrams
    g = _golden(golden_dir, 'scorer_golden.json')
    lines = []
    for i, refs in enumerate(g['corpus_refs']):
        for r in refs:
            lines.append('img%d.jpg,<GO> %s <EOS>' % (i, r))
    out = prepro_ngrams.build(lines)
    assert out['ref_len'] == g['ref_len']
    assert {' '.join(k): v for k, v in out['document_frequency'].items()} == g['document_frequency']


# ----------------------------------------------------------------------------- TF checkpoint-V2 --
def test_tf_bundle_crc_and_snappy_known_answers():
    from comic_amd import tf_bundle as tb
    assert tb.crc32c(b'123456789') == 0xE3069283          # CRC-32C check value (RFC 3720 B.4 family)
    assert tb.crc32c(bytes(32)) == 0x8A9136AA              # RFC 3720 B.4: 32 bytes of zeros
    assert tb.crc32c(bytes([0xFF] * 32)) == 0x62A8AB43     # RFC 3720 B.4: 32 bytes of ones
    assert tb.crc32c(bytes(range(32))) == 0x46DD794E       # RFC 3720 B.4: incrementing
    native = tb._load_native_crc()
    tb._native_crc = False                                 # pure-Python path agrees with the C one
    try:
        assert tb.crc32c(b'123456789') == 0xE3069283 and tb.crc32c(bytes(range(32))) == 0x46DD794E
    finally:
        tb._native_crc = native
    assert tb.unmask_crc(tb.mask_crc(0xE3069283)) == 0xE3069283 and tb.mask_crc(0) == 0xA282EAD8
    # snappy block format: literal 'abcd', 1-byte-offset copy (len 4, off 4), 2-byte-offset copy
    # (len 5, off 8), long literal with an explicit length byte
    long_lit = bytes(range(70))
    stream = bytes([4 + 4 + 5 + 70]) + bytes([3 << 2]) + b'abcd' + bytes([(0 << 2) | 1, 4]) + \
        bytes([((5 - 1) << 2) | 2, 8, 0]) + bytes([60 << 2, 69]) + long_lit
    assert tb.snappy_uncompress(stream) == b'abcdabcd' + b'abcda' + long_lit
    # overlapping copy (run-length): 'x' then copy len 7 off 1
    assert tb.snappy_uncompress(bytes([8, 0, ord('x'), ((7 - 4) << 2) | 1, 1])) == b'x' * 8


def test_tf_bundle_round_trip_and_table_structure(tmp_path, monkeypatch):
    from comic_amd import tf_bundle as tb
    rng = np.random.default_rng(0)
    t = {'Model/decoder/rnn_decoder/basic_lstm_cell/kernel': rng.standard_normal((37, 20)).astype(np.float32),
         'global_step': np.asarray(123, np.int64), 'x/ids': np.arange(5, dtype=np.int32),
         'empty': np.zeros((0, 3), np.float32)}
    for i in range(300):                                   # > 16 entries: restart points + prefix sharing
        t['InceptionV3/Mixed_%03d/weights' % i] = rng.standard_normal((3, 4)).astype(np.float32)
    monkeypatch.setattr(tb, 'BLOCK_SIZE', 2048)            # several data blocks + a real index block
    prefix = str(tmp_path / 'model_compact-123')
    tb.write_bundle(prefix, t)
    assert sorted(os.listdir(tmp_path)) == ['model_compact-123.data-00000-of-00001', 'model_compact-123.index']
    r = tb.read_bundle(prefix)
    assert set(r) == set(t)
    for k in t:
        assert r[k].dtype == t[k].dtype and r[k].shape == t[k].shape and np.array_equal(r[k], t[k]), k
    lv = tb.list_variables(prefix)
    assert lv['global_step'] == (np.dtype(np.int64), ()) and lv['empty'][1] == (0, 3)
    # table structure: footer magic, sorted keys, header entry first, >1 data block
    raw = open(prefix + '.index', 'rb').read()
    assert int.from_bytes(raw[-8:], 'little') == tb.TABLE_MAGIC and len(raw) > 48
    items = tb._read_table(prefix + '.index')
    keys = [k for k, _ in items]
    assert keys[0] == b'' and keys == sorted(keys)
    _, _, pos = tb._decode_handle(raw[-48:], 0)
    ioff, isize, _ = tb._decode_handle(raw[-48:], pos)
    assert len(tb._parse_block(tb._read_block(raw, ioff, isize))) > 3
    # a flipped payload byte is caught by the entry checksum, a flipped index byte by the block checksum
    data = bytearray(open(tb.data_path(prefix), 'rb').read())
    data[10] ^= 1
    open(tb.data_path(prefix), 'wb').write(bytes(data))
    with pytest.raises(ValueError, match='checksum'):
        tb.read_bundle(prefix)
    idx = bytearray(raw)
    idx[5] ^= 1
    open(prefix + '.index', 'wb').write(bytes(idx))
    with pytest.raises(ValueError, match='checksum'):
        tb.list_variables(prefix)


def test_tf_bundle_reads_snappy_compressed_blocks(tmp_path):
    """TF's TableBuilder compresses blocks with snappy by default; build such a file by hand: one
    data block stored as a snappy stream of literals."""
    import struct
    from comic_amd import tf_bundle as tb
    arr = np.arange(6, dtype=np.float32).reshape(2, 3)
    blk = tb._BlockBuilder()
    blk.add(b'', tb._encode_header())
    blk.add(b'v', tb._encode_entry(1, arr.shape, 0, arr.nbytes, tb.mask_crc(tb.crc32c(arr.tobytes()))))
    contents = blk.finish()
    assert len(contents) < 60
    comp = bytes([len(contents)]) + bytes([(len(contents) - 1) << 2]) + contents      # one literal
    prefix = str(tmp_path / 'm-1')
    with open(prefix + '.index', 'wb') as f:
        f.write(comp + b'\x01' + struct.pack('<I', tb.mask_crc(tb.crc32c(comp + b'\x01'))))
        h = bytearray()
        tb._put_varint(h, 0)
        tb._put_varint(h, len(comp))
        index = tb._BlockBuilder()
        index.add(b'v', bytes(h))
        meta = tb._emit_block(f, tb._BlockBuilder().finish())
        ih = tb._emit_block(f, index.finish())
        footer = meta + ih
        f.write(footer + b'\x00' * (40 - len(footer)) + struct.pack('<Q', tb.TABLE_MAGIC))
    open(tb.data_path(prefix), 'wb').write(arr.tobytes())
    out = tb.read_bundle(prefix)
    assert list(out) == ['v'] and np.array_equal(out['v'], arr)


def test_checkpoint_tf_container_three_way_restore(tmp_path):
    """checkpoint.save(fmt='tf') -> restore(): whole model, Adam slots under TF names, slim-style
    CNN-only bundle (names without the Model/encoder/cnn/ prefix), latest_checkpoint."""
    from comic_amd import checkpoint as ckpt, decoder as cdec, tf_bundle as tb
    spec = cdec.DecoderSpec(D=16, E=8, V=20, C=12, Cg=12, H=2, M=4)
    rng = np.random.default_rng(1)
    dec = {k: rng.standard_normal(v).astype(np.float32) for k, v in spec.param_shapes().items()}
    cnn = {'InceptionV3/Conv2d_1a_3x3/weights': rng.standard_normal((3, 3, 3, 4)).astype(np.float32),
           'InceptionV3/Conv2d_1a_3x3/BatchNorm/beta': rng.standard_normal(4).astype(np.float32)}
    m = {k: rng.standard_normal(v.shape).astype(np.float32) for k, v in dec.items()}
    v = {k: rng.random(v.shape).astype(np.float32) for k, v in dec.items()}
    extra = ckpt.adam_to_tf(spec, m, v, t=7)
    p = ckpt.save(str(tmp_path / 'model'), 7, cnn, spec, dec, extra, fmt='tf')
    assert p.endswith('model-7') and os.path.isfile(p + '.index')
    names = tb.list_variables(p)
    assert 'optimise/caption/beta1_power' in names and 'global_step' in names
    assert any(n.endswith('basic_lstm_cell/kernel/Adam_1') for n in names)
    assert ckpt.latest_checkpoint(str(tmp_path), 'model') == p and tb.latest_checkpoint(str(tmp_path)) == p
    c2, d2, ex = ckpt.restore(p, list(cnn), spec, resume_training=True)
    assert all(np.array_equal(c2[k], cnn[k]) for k in cnn) and all(np.array_equal(d2[k], dec[k]) for k in dec)
    assert int(ex['global_step']) == 7
    m2, v2 = ckpt.adam_from_tf(spec, ex)
    assert all(np.array_equal(m2[k], m[k]) and np.array_equal(v2[k], v[k]) for k in dec)
    np.testing.assert_allclose(ex['optimise/caption/beta2_power'], 0.999 ** 8, rtol=1e-6)
    slim = str(tmp_path / 'inception_v3.ckpt')             # slim checkpoint: bare variable names
    tb.write_bundle(slim, cnn)
    c3, d3, _ = ckpt.restore(slim, list(cnn), spec)
    assert d3 is None and all(np.array_equal(c3[k], cnn[k]) for k in cnn)
    # max_to_keep evicts the oldest bundles AND their entries of the `checkpoint` state file
    for step in (8, 9, 10):
        ckpt.save(str(tmp_path / 'model'), step, cnn, spec, dec, extra, max_to_keep=2, fmt='tf')
    state = open(str(tmp_path / 'checkpoint')).read()
    assert 'model-7"' not in state and 'model-8"' not in state and 'model-9"' in state and 'model-10"' in state
    assert not os.path.isfile(str(tmp_path / 'model-8.index')) and os.path.isfile(str(tmp_path / 'model-10.index'))
    assert ckpt.latest_checkpoint(str(tmp_path), 'model').endswith('model-10')
    # `--optimiser sgd`: tf.train.MomentumOptimizer's layout -- one `<var>/Momentum` slot, no beta power accumulators
    mom = ckpt.momentum_to_tf(spec, m)
    assert not any(k.endswith('/Adam') or 'beta1_power' in k for k in mom)
    assert any(k.endswith('basic_lstm_cell/kernel/Momentum') for k in mom)
    p2 = ckpt.save(str(tmp_path / 'sgd'), 3, cnn, spec, dec, mom, fmt='tf')
    _, _, ex2 = ckpt.restore(p2, list(cnn), spec, resume_training=True)
    acc = ckpt.momentum_from_tf(spec, ex2)
    assert acc is not None and all(np.array_equal(acc[k], m[k]) for k in dec) and ckpt.adam_from_tf(spec, ex2) is None


def test_cli_accepts_every_option_of_the_reference(tmp_path):
    """Nothing is refused any more: gradient clipping (clip_gradient_norm in the kwargs, model_base.py:394-401; round 4),
    sgd, every --initialiser value (all Xavier-uniform in the reference, model_base.py:823-831), --legacy, variational
    recurrent dropout and the LN_LSTM / GRU cells (round 3) build."""
    import importlib.util
    spec = importlib.util.spec_from_file_location('train_cli2', os.path.join(ROOT, 'src', 'train.py'))
    train = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(train)
    base = ['--log_root', str(tmp_path)]
    kw, _, _ = train.build_kwargs(train.create_parser().parse_args(base))
    kw['clip_gradient_norm'] = 5.0
    train.check_supported(kw)
    for ok in (['--optimiser', 'sgd'], ['--initialiser', 'he'], ['--initialiser', 'none'], ['--rnn_recurr_dropout', 'True'],
               ['--rnn_name', 'GRU'], ['--rnn_name', 'LN_LSTM'], ['--legacy', 'True']):
        kw, _, _ = train.build_kwargs(train.create_parser().parse_args(base + ok))
    assert kw['legacy'] and kw['cnn_name'] == 'inception_v1' and kw['adam_epsilon'] == 1e-6


def test_loader_flags_and_descriptor_layout(tmp_path):
    """The loader's additions to the CLIs reach the config, and the numpy view of comic_image_desc that the split JPEG loader
    fills vectorised has the ctypes structure's layout."""
    import ctypes as C
    import importlib.util
    from comic_amd import inputs
    spec = importlib.util.spec_from_file_location('train_cli3', os.path.join(ROOT, 'src', 'train.py'))
    train = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(train)
    kw, _, _ = train.build_kwargs(train.create_parser().parse_args(
        ['--log_root', str(tmp_path), '--loader_split_jpeg', '--loader_threads', '4', '--loader_cache_gb', '2.5']))
    assert kw['loader_split_jpeg'] is True and kw['loader_threads'] == 4 and kw['loader_cache_gb'] == 2.5
    kw, _, _ = train.build_kwargs(train.create_parser().parse_args(['--log_root', str(tmp_path)]))
    assert kw['loader_split_jpeg'] is True and kw['loader_cache_gb'] == 0.0           # the split decoder is the default loader
    kw, _, _ = train.build_kwargs(train.create_parser().parse_args(['--log_root', str(tmp_path), '--no-loader_split_jpeg']))
    assert kw['loader_split_jpeg'] is False
    spec = importlib.util.spec_from_file_location('infer_cli3', os.path.join(ROOT, 'src', 'infer.py'))
    infer = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(infer)
    a = infer.create_parser().parse_args(['--loader_split_jpeg', '--loader_threads', '3'])
    assert a.loader_split_jpeg is True and a.loader_threads == 3 and a.loader_cache_gb is None
    assert infer.create_parser().parse_args([]).loader_split_jpeg is None          # the training run's choice applies
    dt = inputs.DevicePreprocessor._DESC_DTYPE
    assert dt.itemsize == C.sizeof(L.ImageDesc) == 40
    for name, _ in L.ImageDesc._fields_:
        assert dt.fields[name][1] == getattr(L.ImageDesc, name).offset, name
        assert dt.fields[name][0].itemsize == getattr(L.ImageDesc, name).size, name


def test_data_parallel_shards_are_disjoint_and_cover_the_epoch(tmp_path):
    """Input managers under data parallelism: a common shuffle, rank r takes items r, r+W, ..; max_step counts global
    batches; same rand_seed (= same parameter initialisation) on every rank."""
    from tests import tiny_dataset
    from comic_amd import inputs
    d = tiny_dataset.make(str(tmp_path), n_train=8, n_valid=4)
    seen, steps, n_items = [], [], None
    for rank in range(2):
        c = conf.Config(dataset_dir=d, dataset_file_pattern='mscoco_{}_w5_s20_include_restval', token_type='radix',
                        radix_base=256, batch_size_train=4, batch_size_eval=4, max_epoch=2, cnn_input_size=[64, 64],
                        cnn_input_augment=True, rand_seed=7, dp_world=2, dp_rank=rank, loader_threads=1, loader_prefetch=1)
        m = inputs.InputManager_Radix(c)
        steps.append(c.max_step)
        data = list(m._read_split('train'))
        n_items = len(data)
        gen = m._gen(data, True)
        seen.append([(p, tuple(int(v) for v in cap)) for (p, cap), _ in zip(gen, range(n_items // 2))])
        m.close()
    assert n_items == 40 and steps[0] == steps[1] == int(n_items / (4 * 2) * 2)
    assert len(set(seen[0]) & set(seen[1])) == 0 and len(seen[0]) + len(seen[1]) == n_items


def test_preprocess_oracle_against_scalar_loop_and_host_path():
    """oracle/preprocess_ref.py (vectorised float32) against a scalar restatement of TF-1.9's resize_bilinear loop on a
    small image, and the product's CPU path (inputs.preprocess_image) against the oracle, bit for bit."""
    from comic_amd import inputs
    from oracle import preprocess_ref as pr
    rng = np.random.default_rng(2)
    im = rng.integers(0, 256, (7, 5, 3), dtype=np.uint8)
    f = np.float32
    x = im.astype(np.float32) * f(1.0 / 255)
    out = np.zeros((256, 256, 3), np.float32)
    sy, sx = f(7 / f(256)), f(5 / f(256))
    for yy in range(0, 256, 37):
        for xx in range(0, 256, 41):
            iy, ix = f(yy) * sy, f(xx) * sx
            y0, x0 = int(np.floor(iy)), int(np.floor(ix))
            y1, x1 = min(int(np.ceil(iy)), 6), min(int(np.ceil(ix)), 4)
            yl, xl = f(iy - f(y0)), f(ix - f(x0))
            top = x[y0, x0] + (x[y0, x1] - x[y0, x0]) * xl
            bot = x[y1, x0] + (x[y1, x1] - x[y1, x0]) * xl
            out[yy, xx] = top + (bot - top) * yl
    ref = pr.resize_bilinear(x)
    for yy in range(0, 256, 37):
        for xx in range(0, 256, 41):
            np.testing.assert_array_equal(ref[yy, xx], out[yy, xx])
    big = rng.integers(0, 256, (97, 131, 3), dtype=np.uint8)
    np.testing.assert_array_equal(inputs.preprocess_image(big, 224, 224, True, None, (True, 5, 30)),
                                  pr.preprocess_image(big, 224, 224, True, 5, 30))
    np.testing.assert_array_equal(inputs.preprocess_image(big, 224, 224, False, None), pr.preprocess_image(big, 224, 224))
    assert pr.preprocess_image(big, 224, 224).min() >= -1.0 and pr.preprocess_image(big, 224, 224).max() <= 1.0


def test_bench_encoder_group_divides_the_timed_steps():
    """bench.py: steps per encoder forward = the largest divisor of K up to DEFAULT_ENC_GROUP, so K timed steps issue
    exactly K * BATCH images of encoder work (the driver runs --steps 20, the default run 30)."""
    import importlib.util
    spec = importlib.util.spec_from_file_location('bench_mod', os.path.join(ROOT, 'bench.py'))
    bench = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(bench)
    assert bench.pick_encoder_group(30) == 30 and bench.pick_encoder_group(20) == 20 and bench.pick_encoder_group(5) == 5
    assert bench.pick_encoder_group(50) == 25 and bench.pick_encoder_group(100) == 25 and bench.pick_encoder_group(7) == 7
    assert bench.pick_encoder_group(1) == 1 and bench.pick_encoder_group(37) == bench.DEFAULT_ENC_GROUP   # prime above the cap
    for k in range(1, 121):
        g = bench.pick_encoder_group(k)
        assert 1 <= g <= max(bench.DEFAULT_ENC_GROUP, 1) and (k % g == 0 or g == bench.DEFAULT_ENC_GROUP)



def test_backward_schedule_of_the_branch_lanes():
    """nets.backward_schedule (comic_cnn_backward_sched): every op of the plan exactly once, consumers before producers
    inside a lane, lane 1 only between a FORK and its JOIN_ADD, and a lane-1 op that feeds a block's shared input gradient
    goes to the alternate buffer that the block's join adds in."""
    from comic_amd import nets
    for name in ('inception_v3',):
        plan = nets.CnnPlan(name, (224, 224))
        sched, alt = nets.backward_schedule(plan)
        runs = [r for r in sched if r[0] == nets.SCHED_RUN]
        assert sorted(int(r[1]) for r in runs) == [i for i, o in enumerate(plan.ops) if o['kind'] not in (5, 6)]
        assert sum(1 for r in sched if r[0] == nets.SCHED_FORK) == sum(1 for r in sched if r[0] == nets.SCHED_JOIN_ADD) == 11
        open_, produced_later = False, set()
        pos = {int(r[1]): k for k, r in enumerate(sched) if r[0] == nets.SCHED_RUN}
        for k, r in enumerate(sched):
            if r[0] == nets.SCHED_FORK:
                assert not open_
                open_ = True
            elif r[0] == nets.SCHED_JOIN_ADD:
                assert open_ and int(r[1]) in alt
                open_ = False
            else:
                o = plan.ops[int(r[1])]
                assert r[2] == 0 or open_
                if r[3]:
                    assert r[2] == 1 and o['src'] == o['block_in'] and o['src'] in alt
                elif r[2] == 1:
                    assert o['src'] != o.get('block_in')           # lane 1 never touches the shared gradient directly
                # every consumer of this op's output runs earlier in the schedule (reverse topological order)
                for j, c in enumerate(plan.ops):
                    if c['kind'] not in (5, 6) and c['src'] == o['dst'] and j in pos:
                        assert pos[j] < k, (j, int(r[1]))
        assert not open_
        assert len(alt) == 11 and plan.input not in alt
        # Inside a fork / join region the lanes run concurrently: a gradient buffer one lane READS (its op's dst) or accumulates
        # into (its op's src; 'alt' copies apart) must not be accumulated into by the other lane.  (The head's global pool adds
        # d im_embed into the gradient of the last block's OUTPUT, which every branch of that block reads: it has to run before
        # the fork -- with the fork in front of it lane 1 raced it, 1e-4 run-to-run differences in the CNN gradients.)
        region, regions = None, []
        for r in sched:
            if r[0] == nets.SCHED_FORK:
                region = []
            elif r[0] == nets.SCHED_JOIN_ADD:
                regions.append(region)
                region = None
            elif region is not None:
                region.append((int(r[2]), plan.ops[int(r[1])], int(r[3])))
        assert len(regions) == 11
        for reg in regions:
            assert all(o.get('branch') is not None for _, o, _ in reg)
            for lane in (0, 1):
                writes = {('alt' if a else 'main', o['src']) for ln, o, a in reg if ln == lane and o['kind'] <= 4 and o['src'] != plan.input}
                other_reads = {('main', o['dst']) for ln, o, a in reg if ln != lane}
                other_writes = {('alt' if a else 'main', o['src']) for ln, o, a in reg if ln != lane and o['src'] != plan.input}
                assert not (writes & other_reads) and not (writes & other_writes), (lane, writes & (other_reads | other_writes))
        gap = [k for k, r in enumerate(sched) if r[0] == nets.SCHED_RUN and plan.ops[int(r[1])]['kind'] == 4]
        first_fork = min(k for k, r in enumerate(sched) if r[0] == nets.SCHED_FORK)
        assert gap and gap[0] < first_fork


def test_fused_chain_plans_are_well_formed():
    """CnnPlan(fuse_chains=...) (csrc/conv_img.hip conv_img_chain_kernel; inception_v3.py:262-366): the chain groups hold one or two
    chains, chain after chain, every conv but a chain's last linked to the NEXT op of the table; every op still runs after
    the producer of its source; frozen plans do not keep the intermediate maps, trainable plans do; the default builds the
    depth-major sibling that encoders take for small batches; other map sizes (299 px: 17x17) and backbones have no chains."""
    from comic_amd import nets
    for kw, keep in ((dict(pool_after_projection=True, fuse_pools=True), 0), (dict(), L.OP_CHAIN_KEEP)):
        plan = nets.CnnPlan('inception_v3', (224, 224), **kw)
        sib = plan.small_batch_plan
        assert plan.fuse_chains and sib is not None and not sib.fuse_chains and sib.small_batch_plan is None
        assert sib.buffers == plan.buffers and sib.weights == plan.weights and sib.end_points == plan.end_points and sib.macs == plan.macs
        assert not any(o.get('tile') == L.CHAIN_TILE or o.get('flags', 0) & L.OP_CHAIN_LINK for o in sib.ops)
        assert sorted((o['kind'], o['src'], o['dst'], o.get('weight', -1)) for o in sib.ops) == \
            sorted((o['kind'], o['src'], o['dst'], o.get('weight', -1)) for o in plan.ops)
        groups = {}
        for i, o in enumerate(plan.ops):
            if o.get('tile') == L.CHAIN_TILE:
                groups.setdefault(o['group'], []).append(i)
        assert len(groups) == 5 and sorted(len(g) for g in groups.values()) == [2, 6, 6, 6, 6]
        for g in groups.values():
            assert g == list(range(g[0], g[0] + len(g)))                      # adjacent in the table
            ops = [plan.ops[i] for i in g]
            assert len({o['Cin'] for o in ops}) == 1 and ops[0]['Cin'] in nets.CnnPlan.CHAIN_CHANNELS
            n_chains = 0
            for a, b in zip(ops, ops[1:] + [None]):
                link = a.get('flags', 0) & L.OP_CHAIN_LINK
                assert (a.get('flags', 0) & L.OP_CHAIN_KEEP) == (keep if link else 0)
                if link:
                    assert b is not None and a['dst'] == b['src'] and a['Cout'] == a['Cin'] and a['dst_coff'] == 0
                    assert sum(1 for q in plan.ops if q['src'] == a['dst']) == 1   # nobody else reads the map that stays in the LDS
                else:
                    assert a['Cout'] == 192
                    n_chains += 1
            assert 1 <= n_chains <= 2
        written = {plan.input}
        for o in plan.ops:                                                   # table order == a valid execution order
            if o['kind'] in (5, 6):
                continue
            assert o['src'] in written, o
            written.add(o['dst'])
    assert not any(o.get('tile') == L.CHAIN_TILE for o in nets.CnnPlan('inception_v3', (299, 299), pool_after_projection=True, fuse_pools=True).ops)
    assert nets.CnnPlan('inception_v3', (224, 224), x3=True).small_batch_plan is None
    assert not nets.CnnPlan('inception_v3', (224, 224), x3=True).fuse_chains and not nets.CnnPlan('inception_v1', (224, 224)).fuse_chains
    forced = nets.CnnPlan('inception_v3', (224, 224), fuse_chains=True)
    assert forced.fuse_chains and forced.small_batch_plan is None
    with pytest.raises(ValueError):
        nets.CnnPlan('inception_v3', (224, 224), x3=True, fuse_chains=True)


def test_gradient_clip_chunk_table():
    """optim.GradClip: a variable is cut into chunks of <= 8192 elements; every record names its variable's first chunk and
    chunk count (the second kernel sums exactly those partials in order)."""
    from comic_amd import decoder as cdec, optim
    shapes = {'a': (3, 5000), 'b': (700,), 'c': (), 'd': (40000,)}
    p = cdec.FlatParams(shapes, 'cpu')
    gc = optim.GradClip(p, 2.5)
    t = gc.chunks.numpy()
    assert t.shape == (2 + 1 + 1 + 5, 5) and gc.partial.numel() == 9
    for seg, k in enumerate(shapes):
        rows = t[t[:, 0] == seg]
        n = int(np.prod(shapes[k])) if shapes[k] else 1
        assert rows[:, 2].sum() == n and (rows[:, 2] <= 8192).all()
        assert (rows[:, 3] == np.flatnonzero(t[:, 0] == seg)[0]).all() and (rows[:, 4] == len(rows)).all()
        assert rows[0, 1] == p.offsets[k] and (np.diff(rows[:, 1]) == 8192).all()


def test_bench_gpus_n_builds_the_launcher_command(monkeypatch):
    """`python bench.py --gpus N` without a launcher (no WORLD_SIZE): bench.launch_ranks starts
    `python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 --master-port P bench.py <same
    arguments>` as a CHILD process (never an exec) and hands back its exit code; main() takes that branch before torch is
    imported (the launcher process must not touch the GPU)."""
    import importlib
    import subprocess
    sys.path.insert(0, ROOT)
    bench = importlib.import_module('bench')
    seen = {}

    def fake_run(cmd, env=None, **kw):
        seen['cmd'], seen['env'] = cmd, env
        return subprocess.CompletedProcess(cmd, 7)
    monkeypatch.setattr(subprocess, 'run', fake_run)
    monkeypatch.setattr(sys, 'argv', ['bench.py', '--gpus', '8', '--steps', '20', '--warmup', '5'])
    monkeypatch.delenv('WORLD_SIZE', raising=False)
    with pytest.raises(SystemExit) as e:
        bench.main()
    assert e.value.code == 7
    cmd = seen['cmd']
    assert cmd[:3] == [sys.executable, '-m', 'torch.distributed.run'] and '--nnodes=1' in cmd
    assert cmd[cmd.index('--nproc-per-node') + 1] == '8' and cmd[cmd.index('--master-addr') + 1] == '127.0.0.1'
    assert 0 < int(cmd[cmd.index('--master-port') + 1]) < 65536
    i = cmd.index(os.path.join(ROOT, 'bench.py'))
    assert cmd[i + 1:] == ['--gpus', '8', '--steps', '20', '--warmup', '5']
    assert seen['env']['HSA_ENABLE_IPC_MODE_LEGACY'] == '0'


def test_radix_round_trip_equals_the_two_conversions():
    """ops.radix_ids_to_captions_and_ids (the SCST loop's ids -> text -> target ids in one pass) == id_to_caption followed by
    captions_to_batched_ids, on random rollouts: invalid digits, <GO> / <EOS> in the middle, odd digit counts, empty captions,
    word ids that decode to the special tokens or beyond the vocabulary (a KeyError in both), one- and two-digit words."""
    def mk(nw, base):
        wtoi = {'<PAD>': -1}
        for i in range(nw):
            wtoi['w%d' % i] = i
        for tok in ('<UNK>', '<GO>', '<EOS>'):
            wtoi[tok] = len(wtoi) - 1
        cfg = types.SimpleNamespace(token_type='radix', radix_base=base, wtoi=wtoi, itow={str(v): k for k, v in wtoi.items()})
        return cfg, ops.build_radix_wtoi(wtoi, base)
    rng = np.random.default_rng(5)
    for nw, base in ((10000, 256), (200, 256), (60000, 256), (100, 16), (9, 16)):
        cfg, table = mk(nw, base)
        for trial in range(25):
            N, T = int(rng.integers(1, 30)), int(rng.integers(1, 41))
            ids = rng.integers(-1, base + 2, (N, T))
            if trial % 3 == 0:
                ids[:, T // 2:] = base + 1
            if trial % 5 == 0:
                ids[0] = base + 1
            if trial % 7 == 0:
                ids = np.clip(ids, 0, 3)
            try:
                caps, err = ops.id_to_caption(ids, cfg), None
                want = ops.captions_to_batched_ids([[c] for c in caps], cfg, table)
            except KeyError as e:
                err = str(e)
            try:
                caps2, got = ops.radix_ids_to_captions_and_ids(ids, cfg, table)
                err2 = None
            except KeyError as e:
                err2 = str(e)
            assert err == err2
            if err is None:
                assert caps == caps2 and want.shape == got.shape and (want == got).all()
    # a vocabulary where join + split is not the identity takes the two calls
    cfg, table = mk(50, 16)
    cfg.itow['3'] = 'two words'
    cfg.wtoi['two words'] = 3
    table = ops.build_radix_wtoi(cfg.wtoi, 16)
    ids = np.array([[0, 3, 0, 5, 17, -1]])
    caps, got = ops.radix_ids_to_captions_and_ids(ids, cfg, table)
    assert caps == ops.id_to_caption(ids, cfg) and (got == ops.captions_to_batched_ids([[c] for c in caps], cfg, table)).all()
