#!/usr/bin/env python3
"""infer.py -- same flags, checkpoint discovery, `config.pkl` overlay and output directory
naming as the reference CLI (reference src/infer.py:23-141)."""
import argparse
import os
import sys

os.environ.setdefault('GPU_MAX_HW_QUEUES', '8')      # before the HIP runtime initialises: see comic_amd/__init__.py

CURR_DIR = os.path.dirname(os.path.realpath(__file__))
BASE_DIR = os.path.dirname(CURR_DIR)
sys.path.insert(0, BASE_DIR)
pjoin = os.path.join


def create_parser():
    p = argparse.ArgumentParser(formatter_class=argparse.RawDescriptionHelpFormatter)
    a = p.add_argument
    a('--infer_set', type=str, default='test', choices=['test', 'valid', 'coco_test', 'coco_valid'],
      help='The split to perform inference on.')
    a('--infer_checkpoints_dir', type=str, default=pjoin('mscoco', 'radix_b256_add_LN_softmax_h8_tie_lstm_run_01'),
      help='The directory containing the checkpoint files.')
    a('--infer_checkpoints', type=str, default='all', help='The checkpoint numbers to be evaluated. Comma-separated.')
    a('--annotations_file', type=str, default='captions_val2014.json',
      help='The annotations / reference file for calculating scores.')
    a('--dataset_dir', type=str, default=pjoin(BASE_DIR, 'datasets', 'mscoco'), help='Dataset directory.')
    a('--run_inference', type=bool, default=True, help='Whether to perform inference.')
    a('--get_metric_score', type=bool, default=True, help='Whether to perform metric score calculations.')
    a('--save_attention_maps', type=bool, default=False, help='Whether to save attention maps to disk as pickle file.')
    a('--gpu', type=str, default='0', help='The gpu number.')
    a('--per_process_gpu_memory_fraction', type=float, default=0.75, help='The fraction of GPU memory allocated.')
    a('--infer_beam_size', type=int, default=3, help='The beam size.')
    a('--infer_length_penalty_weight', type=float, default=0.0, help='The length penalty weight used in beam search.')
    a('--infer_max_length', type=int, default=30, help='The maximum caption length allowed during inference.')
    a('--batch_size_infer', type=int, default=25, help='The batch size.')
    # additions of this framework (the training run's choices apply when left out)
    a('--loader_split_jpeg', action=argparse.BooleanOptionalAction, default=None,
      help='Split JPEG decode: C threads undo the entropy coding, the device does the pixels (bit-identical to PIL).')
    a('--loader_threads', type=int, default=None, help='Decode threads of the loader.')
    a('--loader_cache_gb', type=float, default=None, help='Coefficient cache of the split JPEG decoder, GB.')
    return p


def main(argv=None):
    from comic_amd import configuration as conf, infer_fn as infer
    from comic_amd.configuration import natural_keys
    ckpt_prefix = 'model_compact-'
    args = create_parser().parse_args(argv)
    if not os.path.isabs(args.infer_checkpoints_dir):
        args.infer_checkpoints_dir = pjoin(BASE_DIR, 'experiments', args.infer_checkpoints_dir)
    if args.infer_checkpoints == 'all':
        files = sorted(os.listdir(args.infer_checkpoints_dir), key=natural_keys)
        # `.npz` containers and TF tensor bundles (`model_compact-N.index`, like the reference's own)
        files = [f[len(ckpt_prefix):].rsplit('.', 1)[0] for f in files
                 if f.startswith(ckpt_prefix) and (f.endswith('.npz') or f.endswith('.index'))]
        files = sorted(set(files), key=natural_keys)
        if len(files) > 20:
            files = files[-12:]
        args.infer_checkpoints = files
    else:
        args.infer_checkpoints = args.infer_checkpoints.split(',')
        if len(args.infer_checkpoints) < 1:
            raise ValueError('`infer_checkpoints` must be either `all` or a list of comma-separated checkpoint numbers.')
    c = conf.load_config(pjoin(args.infer_checkpoints_dir, 'config.pkl'))
    c.__dict__.update({k: v for k, v in args.__dict__.items() if v is not None})
    save_name = 'beam_{}_lpen_{}'.format(c.infer_beam_size, c.infer_length_penalty_weight)
    save_name = {'test': 'infer_test_', 'valid': 'infer_valid_', 'coco_test': 'infer_cocoTest_',
                 'coco_valid': 'infer_cocoValid_'}[c.infer_set] + save_name
    c.infer_save_path = pjoin(c.infer_checkpoints_dir, save_name)
    if os.path.exists(c.infer_save_path):
        print('\nINFO: `eval_log_path` already exists.')
    else:
        print('\nINFO: `eval_log_path` will be created.')
        os.mkdir(c.infer_save_path)
    import torch
    torch.cuda.set_device(int(str(c.gpu).split(',')[0]))
    scores_combined = {}
    for ckpt_num in c.infer_checkpoints:
        path = pjoin(c.infer_checkpoints_dir, ckpt_prefix + ckpt_num)
        path = path + '.npz' if os.path.isfile(path + '.npz') else path       # else: TF bundle prefix
        # metric scores: the native BLEU / ROUGE-L / CIDEr scorers when the annotation file is there (the reference
        # shells out to the Java COCO toolkit for METEOR / SPICE / PTB tokenisation as well, infer_fn.py:295-315)
        ann = c.annotations_file if os.path.isabs(c.annotations_file) else pjoin(c.dataset_dir, 'captions', c.annotations_file)
        evaluator = None
        if c.get_metric_score and os.path.isfile(ann):
            from comic_amd import coco_eval
            c.annotations_file = ann
            evaluator = coco_eval.evaluate_captions
        elif c.get_metric_score:
            print('INFO: annotation file `{}` not found: captions are written, metric scores skipped.'.format(ann))
        infer.evaluate_model(config=c, curr_ckpt_path=path, scores_combined=scores_combined, evaluate_captions=evaluator)
        print('\n')


if __name__ == '__main__':
    main()
