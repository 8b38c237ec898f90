"""End-to-end CLI test on the GPU: train.py (decoder-mode XE) -> infer.py (beam 3) ->
train.py --train_mode cnn_finetune -> train.py --train_mode scst, chained through the run
directories exactly as the reference does (train.py:232-269), on a tiny dataset in the
reference's file formats."""
import glob
import json
import os
import shutil
import sys

import numpy as np
import pytest

pytestmark = pytest.mark.gpu

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _run(module_path, argv):
    import importlib.util
    spec = importlib.util.spec_from_file_location('cli_' + os.path.basename(module_path)[:-3], module_path)
    mod = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(mod)
    mod.main(argv)


def test_train_infer_scst_cli(tmp_path):
    from tests import tiny_dataset
    from comic_amd import configuration as conf
    ds = tiny_dataset.make(str(tmp_path / 'mscoco'), n_train=16, n_valid=4, n_test=4)
    logs = str(tmp_path / 'experiments')
    common = ['--dataset_dir', ds, '--log_root', logs, '--cnn_name', 'inception_v3', '--cnn_fm_attention', 'Mixed_7c',
              '--cnn_input_size', '139,139', '--batch_size_eval', '4', '--rnn_size', '128', '--rnn_word_size', '64']
    _run(os.path.join(ROOT, 'src', 'train.py'), common + ['--train_mode', 'decoder', '--batch_size_train', '8',
                                                          '--max_epoch', '2'])
    run_dir = os.path.join(logs, 'mscoco', 'radix_b256_add_LN_softmax_h8_tie_lstm_run_01')
    assert not glob.glob(os.path.join(logs, 'mscoco', 'error__*')), open(glob.glob(os.path.join(logs, 'mscoco', 'error__*'))[0]).read()
    c = conf.load_config(os.path.join(run_dir, 'config.pkl'))
    assert c.token_type == 'radix' and c.cnn_fm_projection == 'tied' and c.rand_seed == 48964896
    ckpts = sorted(glob.glob(os.path.join(run_dir, 'model_compact-*.npz')))
    assert ckpts, os.listdir(run_dir)
    assert os.path.isfile(os.path.join(run_dir, 'model_size.txt'))
    # ---- inference ----
    _run(os.path.join(ROOT, 'src', 'infer.py'), ['--infer_checkpoints_dir', run_dir, '--dataset_dir', ds,
                                                 '--infer_set', 'test', '--batch_size_infer', '2',
                                                 '--annotations_file', 'captions_test_annotations.json'])
    out_dir = os.path.join(run_dir, 'infer_test_beam_3_lpen_0.0')
    caps = glob.glob(os.path.join(out_dir, 'captions___*.json'))
    assert caps
    data = json.load(open(caps[0]))
    assert len(data) == 4 and all(set(d) == {'image_id', 'caption'} for d in data)
    assert os.path.isfile(os.path.join(out_dir, 'infer_speed.txt'))
    # native metric scores (BLEU-1..4, ROUGE-L, CIDEr) in the reference's report files
    scores = open(os.path.join(out_dir, 'metric_scores.txt')).read()
    assert all(m in scores for m in ('Bleu_1', 'Bleu_4', 'ROUGE_L', 'CIDEr')) and 'METEOR' not in scores
    assert len(open(os.path.join(out_dir, 'metric_scores.csv')).read().strip().split(',')) == 7
    # the inference loop runs the encoder of batch i + 1 under the decode of batch i and keeps the decode loops of two batches
    # in flight (CaptionModel.infer_pipelined): the serial loop -- one batch at a time, encoder then decode -- writes the same
    # captions
    os.rename(caps[0], caps[0] + '.pipelined')
    os.environ['COMIC_PIPELINE_INFER'] = '0'
    os.environ['COMIC_INFER_IN_FLIGHT'] = '1'
    try:
        _run(os.path.join(ROOT, 'src', 'infer.py'), ['--infer_checkpoints_dir', run_dir, '--dataset_dir', ds,
                                                     '--infer_set', 'test', '--batch_size_infer', '2',
                                                     '--get_metric_score', ''])
    finally:
        del os.environ['COMIC_PIPELINE_INFER']
        del os.environ['COMIC_INFER_IN_FLIGHT']
    assert json.load(open(caps[0])) == json.load(open(caps[0] + '.pipelined'))
    # ---- CNN fine-tune: restores the decoder run, trains CNN + decoder, saves both ----
    _run(os.path.join(ROOT, 'src', 'train.py'), common + ['--train_mode', 'cnn_finetune', '--batch_size_train', '8',
                                                          '--max_epoch', '1', '--checkpoint_format', 'tf'])
    errs = glob.glob(os.path.join(logs, 'mscoco', 'error__*'))
    assert not errs, open(errs[0]).read()
    ft_dir = run_dir.replace('_run_01', '_cnnFT_run_01')
    # saved as TF checkpoint-V2 tensor bundles (the reference's own container)
    from comic_amd import tf_bundle
    ft_ckpts = sorted(glob.glob(os.path.join(ft_dir, 'model_compact-*.index')))
    assert ft_ckpts and os.path.isfile(os.path.join(ft_dir, 'checkpoint')), os.listdir(ft_dir)
    a, b = np.load(ckpts[-1]), tf_bundle.read_bundle(ft_ckpts[-1][:-len('.index')])
    k = 'Model/encoder/cnn/InceptionV3/Mixed_7c/Branch_0/Conv2d_0a_1x1/weights'
    assert a[k].shape == b[k].shape and not np.array_equal(a[k], b[k]), 'CNN variables were not trained'
    km = 'Model/encoder/cnn/InceptionV3/Mixed_7c/Branch_0/Conv2d_0a_1x1/BatchNorm/moving_mean'
    np.testing.assert_array_equal(a[km], b[km])         # BN statistics stay frozen (model_base.py:76)
    full = tf_bundle.list_variables(sorted(glob.glob(os.path.join(ft_dir, 'model-*.index')))[-1][:-len('.index')])
    assert 'optimise/caption/beta1_power' in full and 'global_step' in full
    # the CNN variables' optimiser slots: per variable, under the optimiser's scope, of the variable's shape
    kk = 'optimise/caption/' + k
    assert kk + '/Adam' in full and kk + '/Adam_1' in full and not any('cnn_w_adam' in n for n in full)
    slots = tf_bundle.read_bundle(sorted(glob.glob(os.path.join(ft_dir, 'model-*.index')))[-1][:-len('.index')])
    assert slots[kk + '/Adam'].shape == b[k].shape and float(np.abs(slots[kk + '/Adam_1']).max()) > 0
    # ---- SCST on top of the fine-tuned run (restores the TF bundle) ----
    _run(os.path.join(ROOT, 'src', 'train.py'), common + ['--train_mode', 'scst', '--max_epoch', '2',
                                                          '--scst_beam_size', '3'])
    errs = glob.glob(os.path.join(logs, 'mscoco', 'error__*'))
    assert not errs, open(errs[0]).read()
    scst_dirs = glob.glob(os.path.join(logs, 'mscoco', '*_cnnFT_SCST_beam_3_*'))
    assert scst_dirs and glob.glob(os.path.join(scst_dirs[0], 'model_compact-*.npz'))


def test_train_infer_cli_default_backbone(tmp_path):
    """The reference's default backbone and feature map (--cnn_name inception_v1, --cnn_fm_attention
    Mixed_4f: train.py:56,65): one decoder-mode epoch and beam-3 inference through the CLIs."""
    from tests import tiny_dataset
    from comic_amd import configuration as conf
    ds = tiny_dataset.make(str(tmp_path / 'mscoco'), n_train=8, n_valid=4, n_test=4)
    logs = str(tmp_path / 'experiments')
    _run(os.path.join(ROOT, 'src', 'train.py'), ['--dataset_dir', ds, '--log_root', logs, '--batch_size_eval', '4',
                                                 '--rnn_size', '128', '--rnn_word_size', '64', '--train_mode', 'decoder',
                                                 '--batch_size_train', '4', '--max_epoch', '1', '--no-loader_split_jpeg'])
    errs = glob.glob(os.path.join(logs, 'mscoco', 'error__*'))
    assert not errs, open(errs[0]).read()
    run_dir = os.path.join(logs, 'mscoco', 'radix_b256_add_LN_softmax_h8_tie_lstm_run_01')
    c = conf.load_config(os.path.join(run_dir, 'config.pkl'))
    assert c.cnn_name == 'inception_v1' and c.cnn_fm_attention == 'Mixed_4f'
    ck = sorted(glob.glob(os.path.join(run_dir, 'model_compact-*.npz')))
    assert ck
    z = np.load(ck[-1])
    assert z['Model/encoder/cnn/InceptionV1/Mixed_4c/Branch_2/Conv2d_0a_1x1/weights'].shape == (1, 1, 512, 24)
    assert z['Model/decoder/rnn_decoder/memory_layer/kernel'].shape[0] == 832     # attention over Mixed_4f
    _run(os.path.join(ROOT, 'src', 'infer.py'), ['--infer_checkpoints_dir', run_dir, '--dataset_dir', ds,
                                                 '--infer_set', 'test', '--batch_size_infer', '4',
                                                 '--get_metric_score', '', '--no-loader_split_jpeg'])
    caps = glob.glob(os.path.join(run_dir, 'infer_test_beam_3_lpen_0.0', 'captions___*.json'))
    assert caps and len(json.load(open(caps[0]))) == 4
    # --loader_split_jpeg (Huffman decoding on C threads, the pixels on the device): the same captions, and a training run
    os.rename(caps[0], caps[0] + '.pil')
    _run(os.path.join(ROOT, 'src', 'infer.py'), ['--infer_checkpoints_dir', run_dir, '--dataset_dir', ds,
                                                 '--infer_set', 'test', '--batch_size_infer', '4',
                                                 '--get_metric_score', '', '--loader_split_jpeg', '--loader_threads', '3'])
    assert json.load(open(caps[0])) == json.load(open(caps[0] + '.pil'))
    logs2 = str(tmp_path / 'experiments_split')
    _run(os.path.join(ROOT, 'src', 'train.py'), ['--dataset_dir', ds, '--log_root', logs2, '--batch_size_eval', '4',
                                                 '--rnn_size', '128', '--rnn_word_size', '64', '--train_mode', 'decoder',
                                                 '--batch_size_train', '4', '--max_epoch', '1', '--loader_split_jpeg'])
    errs = glob.glob(os.path.join(logs2, 'mscoco', 'error__*'))
    assert not errs, open(errs[0]).read()
    ck2 = sorted(glob.glob(os.path.join(logs2, 'mscoco', 'radix_b256_add_LN_softmax_h8_tie_lstm_run_01', 'model_compact-*.npz')))
    assert ck2
    z2 = np.load(ck2[-1])
    # same files, same seed, bit-identical input tensors: the run with the split decoder ends at the same weights
    np.testing.assert_array_equal(z2['Model/decoder/rnn_decoder/memory_layer/kernel'], z['Model/decoder/rnn_decoder/memory_layer/kernel'])


def test_train_infer_cli_bf16x3_plan(tmp_path):
    """--cnn_dtype bf16x3 (the bf16 matrix cores at fp32-class accuracy: nets.CnnPlan(x3=True)): one decoder-mode epoch on
    InceptionV3, one cnn_finetune epoch on top of it (the x3 backward: fp32 gradient buffers, three-product weight
    gradients) and beam-3 inference from the run directory (the plan follows config.pkl)."""
    from tests import tiny_dataset
    from comic_amd import configuration as conf
    ds = tiny_dataset.make(str(tmp_path / 'mscoco'), n_train=8, n_valid=4, n_test=4)
    logs = str(tmp_path / 'experiments')
    common = ['--dataset_dir', ds, '--log_root', logs, '--cnn_name', 'inception_v3', '--cnn_fm_attention', 'Mixed_7c',
              '--cnn_input_size', '139,139', '--batch_size_eval', '4', '--rnn_size', '128', '--rnn_word_size', '64',
              '--cnn_dtype', 'bf16x3']
    _run(os.path.join(ROOT, 'src', 'train.py'), common + ['--train_mode', 'decoder', '--batch_size_train', '4',
                                                          '--max_epoch', '1'])
    errs = glob.glob(os.path.join(logs, 'mscoco', 'error__*'))
    assert not errs, open(errs[0]).read()
    run_dir = os.path.join(logs, 'mscoco', 'radix_b256_add_LN_softmax_h8_tie_lstm_run_01')
    assert conf.load_config(os.path.join(run_dir, 'config.pkl')).cnn_dtype == 'bf16x3'
    assert sorted(glob.glob(os.path.join(run_dir, 'model_compact-*.npz')))
    _run(os.path.join(ROOT, 'src', 'infer.py'), ['--infer_checkpoints_dir', run_dir, '--dataset_dir', ds,
                                                 '--infer_set', 'test', '--batch_size_infer', '4',
                                                 '--get_metric_score', ''])
    caps = glob.glob(os.path.join(run_dir, 'infer_test_beam_3_lpen_0.0', 'captions___*.json'))
    assert caps and len(json.load(open(caps[0]))) == 4
    # cnn_finetune on top of the decoder run, same plan: the CNN variables move
    z0 = np.load(sorted(glob.glob(os.path.join(run_dir, 'model_compact-*.npz')))[-1])
    _run(os.path.join(ROOT, 'src', 'train.py'), common + ['--train_mode', 'cnn_finetune', '--batch_size_train', '4',
                                                          '--max_epoch', '1'])
    errs = glob.glob(os.path.join(logs, 'mscoco', 'error__*'))
    assert not errs, open(errs[0]).read()
    ft_dir = run_dir.replace('_run_01', '_cnnFT_run_01')
    ck = sorted(glob.glob(os.path.join(ft_dir, 'model_compact-*.npz')))
    assert ck, os.listdir(os.path.join(logs, 'mscoco'))
    z1 = np.load(ck[-1])
    k = 'Model/encoder/cnn/InceptionV3/Mixed_7c/Branch_0/Conv2d_0a_1x1/weights'
    assert np.isfinite(z1[k]).all() and not np.array_equal(z1[k], z0[k]), 'CNN variables were not trained'


@pytest.mark.parametrize('rnn,var', [('LN_LSTM', 'layer_norm_basic_lstm_cell/state/gamma'), ('GRU', 'gru_cell/candidate/kernel')])
def test_train_infer_cli_other_cells(tmp_path, rnn, var):
    """--rnn_name LN_LSTM / GRU (train.py:73-75, model_base.py:622-629): one decoder-mode epoch, the cell's variables under
    their TensorFlow names in the checkpoint, beam-3 inference from it."""
    from tests import tiny_dataset
    ds = tiny_dataset.make(str(tmp_path / 'mscoco'), n_train=8, n_valid=4, n_test=4)
    logs = str(tmp_path / 'experiments')
    _run(os.path.join(ROOT, 'src', 'train.py'), ['--dataset_dir', ds, '--log_root', logs, '--batch_size_eval', '4',
                                                 '--cnn_name', 'inception_v3', '--cnn_fm_attention', 'Mixed_7c',
                                                 '--cnn_input_size', '139,139', '--rnn_name', rnn, '--name', rnn.lower(),
                                                 '--rnn_size', '128', '--rnn_word_size', '64', '--train_mode', 'decoder',
                                                 '--batch_size_train', '4', '--max_epoch', '2'])
    errs = glob.glob(os.path.join(logs, 'mscoco', 'error__*'))
    assert not errs, open(errs[0]).read()
    run_dir = os.path.join(logs, 'mscoco', 'radix_b256_add_LN_softmax_h8_tie_%s_run_01' % rnn.lower())
    ck = sorted(glob.glob(os.path.join(run_dir, 'model_compact-*.npz')))
    assert ck, os.listdir(os.path.join(logs, 'mscoco'))
    z = np.load(ck[-1])
    names = [n for n in z.files if n.startswith('Model/decoder/rnn_decoder/rnn_init_input/')]
    assert 'Model/decoder/rnn_decoder/rnn_init_input/' + var in names, names
    assert not any('basic_lstm_cell/' in n and 'layer_norm' not in n for n in names), names
    if rnn == 'GRU':
        assert z['Model/decoder/rnn_decoder/rnn_init_input/gru_cell/gates/kernel'].shape == (64 + 128 + 128, 256)
        assert z['Model/decoder/rnn_decoder/rnn_init_input/gru_cell/gates/bias'].shape == (256,)
    else:
        assert z['Model/decoder/rnn_decoder/rnn_init_input/layer_norm_basic_lstm_cell/kernel'].shape == (320, 512)
        assert not any(n.endswith('layer_norm_basic_lstm_cell/bias') for n in names)
    _run(os.path.join(ROOT, 'src', 'infer.py'), ['--infer_checkpoints_dir', run_dir, '--dataset_dir', ds,
                                                 '--infer_set', 'test', '--batch_size_infer', '4',
                                                 '--get_metric_score', ''])
    caps = glob.glob(os.path.join(run_dir, 'infer_test_beam_3_lpen_0.0', 'captions___*.json'))
    assert caps and len(json.load(open(caps[0]))) == 4


def test_legacy_head_tf_checkpoint_slots_and_resume(tmp_path):
    """--legacy (model_base.py:80-91: LN_tanh + im_embed head, trained) with --checkpoint_format tf: the head's optimiser
    slots are written per variable under the optimiser's scope like the decoder's (`optimise/caption/<var>/Adam[_1]`), and a
    second invocation of the same run resumes from the bundle (step counter and slots restored)."""
    from tests import tiny_dataset
    from comic_amd import tf_bundle
    ds = tiny_dataset.make(str(tmp_path / 'mscoco'), n_train=8, n_valid=4, n_test=4)
    logs = str(tmp_path / 'experiments')
    args = ['--dataset_dir', ds, '--log_root', logs, '--batch_size_eval', '4', '--rnn_size', '128', '--rnn_word_size', '64',
            '--train_mode', 'decoder', '--batch_size_train', '4', '--legacy', 'True', '--checkpoint_format', 'tf']
    _run(os.path.join(ROOT, 'src', 'train.py'), args + ['--max_epoch', '1'])
    errs = glob.glob(os.path.join(logs, 'mscoco', 'error__*'))
    assert not errs, open(errs[0]).read()
    run_dirs = glob.glob(os.path.join(logs, 'mscoco', '*_run_01'))
    assert len(run_dirs) == 1
    full = sorted(glob.glob(os.path.join(run_dirs[0], 'model-*.index')))
    assert full
    names = tf_bundle.list_variables(full[-1][:-len('.index')])
    for v in ('Model/encoder/LN_tanh/beta', 'Model/encoder/LN_tanh/gamma', 'Model/encoder/im_embed/weight'):
        assert v in names and 'optimise/caption/' + v + '/Adam' in names and 'optimise/caption/' + v + '/Adam_1' in names, v
    assert not any('head_adam' in n for n in names)
    step1 = int(tf_bundle.read_bundle(full[-1][:-len('.index')])['global_step'])
    _run(os.path.join(ROOT, 'src', 'train.py'), args + ['--max_epoch', '2'])      # the run directory exists: resumes
    errs = glob.glob(os.path.join(logs, 'mscoco', 'error__*'))
    assert not errs, open(errs[0]).read()
    full2 = sorted(glob.glob(os.path.join(run_dirs[0], 'model-*.index')), key=lambda f: int(f.split('model-')[-1].split('.')[0]))
    step2 = int(tf_bundle.read_bundle(full2[-1][:-len('.index')])['global_step'])
    assert step2 > step1


def test_run_train_step_pipelined_equals_serial(tmp_path):
    """CaptionModel.run_train_step (the reference-API entry, == sess.run(m_train.dec_log_ppl)) with the frozen-CNN
    pipelining on -- the encoder forward of the next step(s) on a second stream, `encoder_group` steps per forward --
    gives the losses and the parameters of the serial step sequence bit for bit, on batches drawn from the real
    input pipeline (JPEG decode -> device preprocessing -> bucketed captions)."""
    import importlib.util
    import torch
    from tests import tiny_dataset
    from comic_amd import model as mdl, train_fn as train
    ds = tiny_dataset.make(str(tmp_path / 'mscoco'), n_train=32, n_valid=4, n_test=4)
    spec = importlib.util.spec_from_file_location('cli_train_probe', os.path.join(ROOT, 'src', 'train.py'))
    cli = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(cli)
    args = cli.create_parser().parse_args(
        ['--dataset_dir', ds, '--log_root', str(tmp_path / 'experiments'), '--cnn_name', 'inception_v3',
         '--cnn_fm_attention', 'Mixed_7c', '--cnn_input_size', '139,139', '--batch_size_eval', '4', '--rnn_size', '128',
         '--rnn_word_size', '64', '--train_mode', 'decoder', '--batch_size_train', '4', '--max_epoch', '2'])
    kwargs, _, overwrite = cli.build_kwargs(args)
    runs = []

    def probe(config):
        assert config.encoder_group == 0                 # the CLI default: auto (train.py --encoder_group)
        for pipe, group in ((False, 1), (True, 1), (True, 2), (True, 0)):
            mdl.reset_default_graph()
            config.pipeline_encoder, config.encoder_group = pipe, group
            man = train._manager(config)
            try:
                man.enable_device_preprocess('cuda:0')
                m = mdl.CaptionModel(config, mode='train', batch_ops=man.batch_train, reuse=False, name='train',
                                     device='cuda:0')
                losses = [float(m.run_train_step()) for _ in range(7)]
                torch.cuda.synchronize()
                runs.append((losses, m.decoder.params.data.cpu().numpy().copy()))
            finally:
                man.close()
    train.try_to_train(train_fn=probe, try_block=False, overwrite=overwrite, **kwargs)
    (l0, p0), (l1, p1), (l2, p2), (l3, p3) = runs
    assert len(set(round(v, 6) for v in l0)) == len(l0)        # the steps see different batches
    assert l1 == l0 and l2 == l0
    np.testing.assert_array_equal(p1, p0)
    np.testing.assert_array_equal(p2, p0)
    # auto group: 16 steps (64 images) per forward.  At that many pixels the plan executor switches two launches to
    # forms with another fp32 summation order (row-walking pool + BN + ReLU, weight-stationary 1x1 groups), so this
    # run agrees to rounding, not bit for bit
    np.testing.assert_allclose(l3, l0, rtol=2e-4)
    assert np.abs(p3 - p0).max() <= 2e-3 * np.abs(p0).max()
    assert mdl.auto_encoder_group(64) == 20 and mdl.auto_encoder_group(32) == 40 and mdl.auto_encoder_group(4096) == 1


def test_model_autotunes_encoder_and_caches_variants(tmp_path):
    """At real problem sizes the reference-API model picks the conv kernel variants by timing them once
    (CnnEncoder.autotune) and caches the choice in the run directory; a second model of the same shape loads the cache
    and trains to the same loss (every variant gives the same bits)."""
    import importlib.util
    import torch
    from tests import tiny_dataset
    from comic_amd import model as mdl, train_fn as train
    ds = tiny_dataset.make(str(tmp_path / 'mscoco'), n_train=32, n_valid=4, n_test=4)
    spec = importlib.util.spec_from_file_location('cli_train_probe2', os.path.join(ROOT, 'src', 'train.py'))
    cli = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(cli)
    args = cli.create_parser().parse_args(
        ['--dataset_dir', ds, '--log_root', str(tmp_path / 'experiments'), '--cnn_name', 'inception_v3',
         '--cnn_fm_attention', 'Mixed_7c', '--cnn_input_size', '224,224', '--batch_size_eval', '4', '--rnn_size', '128',
         '--rnn_word_size', '64', '--train_mode', 'decoder', '--batch_size_train', '16', '--max_epoch', '2'])
    kwargs, _, overwrite = cli.build_kwargs(args)
    out = []

    def probe(config):
        cache = os.path.join(config.log_path, 'conv_variants.json')
        for _ in range(2):
            mdl.reset_default_graph()
            man = train._manager(config)
            try:
                man.enable_device_preprocess('cuda:0')
                m = mdl.CaptionModel(config, mode='train', batch_ops=man.batch_train, reuse=False, name='train',
                                     device='cuda:0')
                enc = m._encoder_for(16)
                tiles = [int(enc._ops[i].tile) for i in range(len(m.plan.ops))]
                losses = [float(m.run_train_step()) for _ in range(2)]
                torch.cuda.synchronize()
                out.append((tiles, losses, os.path.isfile(cache)))
            finally:
                man.close()
    train.try_to_train(train_fn=probe, try_block=False, overwrite=overwrite, **kwargs)
    (t0, l0, c0), (t1, l1, c1) = out
    assert c0 and c1 and any(t > 0 for t in t0), 'the encoder was not tuned / the cache was not written'
    assert t1 == t0 and l1 == l0
