#!/usr/bin/env python3
"""train.py -- same flags, mode chaining, run-directory naming and kwargs as the reference
CLI (reference src/train.py:25-312), driving the MI355X-native path.

Differences (all additive): `--cnn_dtype {bf16,f32,bf16x3}`, `--checkpoint_format {npz,tf}`; launched under
`python -m torch.distributed.run --nproc-per-node N` it trains data-parallel (one process
per GPU, RCCL gradient all-reduce); the slim checkpoint is not downloaded (no network):
pass `--checkpoint_path` (an .npz with slim variable names) or train the CNN from random init.
"""
import argparse
import os
import sys

os.environ.setdefault('GPU_MAX_HW_QUEUES', '8')      # before the HIP runtime initialises: see comic_amd/__init__.py

CURR_DIR = os.path.dirname(os.path.realpath(__file__))
sys.path.insert(0, os.path.dirname(CURR_DIR))
pjoin = os.path.join


def create_parser():
    p = argparse.ArgumentParser(formatter_class=argparse.RawDescriptionHelpFormatter)
    a = p.add_argument
    a('--name', type=str, default='lstm', help='The logging name.')
    a('--dataset_dir', type=str, default='', help='The dataset directory.')
    a('--dataset_file_pattern', type=str, default='mscoco_{}_w5_s20_include_restval',
      help='The dataset text files naming pattern.')
    a('--train_mode', type=str, default='decoder', choices=['decoder', 'cnn_finetune', 'scst'],
      help='Str. The training regime.')
    a('--legacy', type=bool, default=False, help='If True, will match settings as described in paper.')
    a('--token_type', type=str, default='radix', choices=['radix', 'word', 'char'], help='The language model.')
    a('--radix_base', type=int, default=256, help='The base for Radix models.')
    a('--cnn_name', type=str, default='inception_v1', help='The CNN model name.')
    a('--cnn_input_size', type=str, default='224,224', help='The network input size.')
    a('--cnn_input_augment', type=bool, default=True, help='Whether to augment input images.')
    a('--cnn_fm_attention', type=str, default='Mixed_4f', help='String, name of feature map for attention.')
    a('--cnn_fm_projection', type=str, default='tied', choices=['none', 'independent', 'tied'],
      help='String, feature map projection, from `none`, `independent`, `tied`.')
    a('--rnn_name', type=str, default='LSTM', choices=['LSTM', 'LN_LSTM', 'GRU'],
      help='The type of RNN, from `LSTM`, `LN_LSTM` and `GRU`.')
    a('--rnn_size', type=int, default=512, help='Int, number of RNN units.')
    a('--rnn_word_size', type=int, default=256, help='The word size.')
    a('--rnn_init_method', type=str, default='first_input', choices=['project_hidden', 'first_input'],
      help='The RNN init method.')
    a('--rnn_recurr_dropout', type=bool, default=False, help='Whether to enable variational recurrent dropout.')
    a('--attn_num_heads', type=int, default=8, help='The number of attention heads.')
    a('--attn_context_layer', type=bool, default=False,
      help='If True, add linear projection after multi-head attention.')
    a('--attn_alignment_method', type=str, default='add_LN', choices=['add_LN', 'add', 'dot'],
      help='Str, The alignment method / composition method.')
    a('--attn_probability_fn', type=str, default='softmax', choices=['softmax', 'sigmoid'],
      help='Str, The attention map probability function.')
    a('--attn_keep_prob', type=float, default=0.9, help='Float, The keep rate for attention map dropout.')
    a('--initialiser', type=str, default='xavier', choices=['xavier', 'he', 'none'],
      help='The initialiser: `xavier`, `he`, tensorflow default.')
    a('--optimiser', type=str, default='adam', choices=['adam', 'sgd'], help='The optimiser: `adam`, `sgd`.')
    a('--batch_size_train', type=int, default=32, help='The batch size for training.')
    a('--batch_size_eval', type=int, default=61, help='The batch size for validation.')
    a('--max_epoch', type=int, default=30, help='The max epoch training.')
    a('--lr_start', type=float, default=1e-2, help='Float, determines the starting learning rate.')
    a('--lr_end', type=float, default=1e-5, help='Float, determines the ending learning rate.')
    a('--cnn_grad_multiplier', type=float, default=1.0,
      help='Float, determines the gradient multiplier when back-prop thru CNN.')
    a('--adam_epsilon', type=float, default=1e-2, help='Float, determines the epsilon value of ADAM.')
    a('--scst_beam_size', type=int, default=7, help='The beam size for SCST sampling.')
    a('--scst_weight_ciderD', type=float, default=1.0, help='The weight for CIDEr-D metric during SCST training.')
    a('--scst_weight_bleu', type=str, default='0,0,0,2', help='The weight for BLEU metrics during SCST training.')
    a('--freeze_scopes', type=str, default='Model/encoder/cnn', help='The scopes to freeze / do not train.')
    a('--checkpoint_path', type=str, default=None, help='The checkpoint path.')
    a('--checkpoint_exclude_scopes', type=str, default='', help='The scopes to exclude when restoring from checkpoint.')
    a('--gpu', type=str, default='0', help='The gpu number.')
    a('--run', type=int, default=1, help='The run number.')
    # additions of this framework
    a('--cnn_dtype', type=str, default='bf16', choices=['bf16', 'f32', 'bf16x3'],
      help='CNN activation / MFMA input type.  bf16x3: hi/lo-split activations and filters on the bf16 matrix cores '
           '(fp32-class accuracy, frozen-CNN modes).')
    a('--checkpoint_format', type=str, default='npz', choices=['npz', 'tf'],
      help='Container of saved checkpoints: .npz or the TF checkpoint-V2 tensor bundle (both restore).')
    a('--log_root', type=str, default='', help='Root of the experiments directory (default: ../experiments).')
    a('--encoder_group', type=int, default=0,
      help='Frozen-CNN modes: training steps served by ONE encoder forward (group x batch images per forward, on a '
           'second stream under the decoder steps). 0 = auto (about 1280 images per forward), 1 = one forward per step.')
    a('--loader_processes', type=int, default=0,
      help='JPEG decode in this many worker processes (shared-memory staging); 0 = decode threads in this process.')
    a('--loader_split_jpeg', action=argparse.BooleanOptionalAction, default=True,
      help='Split JPEG decode (default): `loader_threads` C threads undo the entropy coding, the device does inverse DCT / '
           'upsampling / colour conversion (bit-identical to PIL); files it does not take (progressive, CMYK, PNG) go through PIL.  '
           '--no-loader_split_jpeg: PIL decode on threads / --loader_processes.')
    a('--loader_threads', type=int, default=0, help='Decode threads of the loader (0 = min(16, cores)).')
    a('--loader_cache_gb', type=float, default=0.0,
      help='With --loader_split_jpeg: keep the decoded DCT coefficients of the images in host memory (as their non-zeros: about the '
           'size of the JPEG files) up to this many GB; the epochs after the first skip file reads and Huffman decoding.')
    return p


def build_kwargs(args):
    """Everything between argument parsing and `try_to_train` in the reference (train.py:167-302)."""
    args.cnn_input_size = [int(v) for v in str(args.cnn_input_size).split(',')]
    if args.legacy:
        print('LEGACY mode enabled. Some arguments will be overridden.')
        args.__dict__.update(cnn_name='inception_v1', cnn_input_size=[224, 224], cnn_input_augment=True,
                             cnn_fm_attention='Mixed_4f', rnn_name='LSTM', rnn_size=512, rnn_word_size=256,
                             rnn_init_method='project_hidden', rnn_recurr_dropout=False, attn_context_layer=False,
                             attn_alignment_method='add_LN', attn_probability_fn='softmax', attn_keep_prob=1.0,
                             lr_start=1e-3, lr_end=2e-4, lr_reduce_every_n_epochs=4, cnn_grad_multiplier=1.0,
                             initialiser='xavier', optimiser='adam', batch_size_train=32, adam_epsilon=1e-6)
    rand_seed = {1: 48964896, 2: 88888888, 3: 123456789}[args.run]
    dataset = args.dataset_file_pattern.split('_')[0]
    log_root = args.log_root or pjoin(os.path.dirname(CURR_DIR), 'experiments', dataset)
    if args.log_root:
        log_root = pjoin(args.log_root, dataset)
    if args.dataset_dir == '':
        args.dataset_dir = pjoin(os.path.dirname(CURR_DIR), 'datasets', dataset)
    token = 'radix_b{}'.format(args.radix_base) if args.token_type == 'radix' else args.token_type
    name = '_'.join([token, args.attn_alignment_method, args.attn_probability_fn,
                     'h{}'.format(args.attn_num_heads), args.cnn_fm_projection[:3], args.name])
    if args.legacy:
        name = 'legacy_' + name
    dec_dir = pjoin(log_root, '{}_run_{:02d}'.format(name, args.run))
    cnnft_dir = pjoin(log_root, '{}_cnnFT_run_{:02d}'.format(name, args.run))
    train_fn_name = 'train_fn'
    if args.train_mode == 'decoder':
        assert args.freeze_scopes == 'Model/encoder/cnn'
        log_path = dec_dir
    elif args.train_mode == 'cnn_finetune':
        if args.legacy:
            raise NotImplementedError
        if not os.path.exists(dec_dir):
            raise ValueError('Decoder training log path not found: {}'.format(dec_dir))
        args.lr_start, args.max_epoch, args.freeze_scopes, args.checkpoint_path = 1e-3, 10, '', dec_dir
        log_path = cnnft_dir
    else:
        if args.legacy:
            raise NotImplementedError
        if not os.path.exists(cnnft_dir):
            raise ValueError('CNN finetune log path not found: {}'.format(cnnft_dir))
        args.scst_weight_bleu = [float(w) for w in args.scst_weight_bleu.split(',')]
        args.batch_size_train, args.lr_start, args.max_epoch = 10, 1e-3, 10
        args.freeze_scopes, args.checkpoint_path = 'Model/encoder/cnn', cnnft_dir
        scst = 'beam_{}_CrD_{}_B1_{}_B4_{}'.format(args.scst_beam_size, args.scst_weight_ciderD,
                                                   args.scst_weight_bleu[0], args.scst_weight_bleu[-1])
        log_path = pjoin(log_root, '{}_cnnFT_SCST_{}_run_{:02d}'.format(name, scst, args.run))
        train_fn_name = 'train_fn_scst'
    args.resume_training = overwrite = os.path.exists(log_path)
    for k, v in list(args.__dict__.items()):
        if v == 'none':
            args.__dict__[k] = None
    kwargs = dict(rnn_layers=1, dropout_rnn_in=0.35, dropout_rnn_out=0.35, rnn_map_loss_scale=1.0, l2_decay=1e-5,
                  clip_gradient_norm=0, max_saves=12, num_logs_per_epoch=100, per_process_gpu_memory_fraction=None,
                  rand_seed=rand_seed, add_image_summaries=True, add_vars_summaries=False, add_grad_summaries=False,
                  log_path=log_path, save_path=pjoin(log_path, 'model'))
    kwargs.update(args.__dict__)
    kwargs.pop('log_root', None)
    check_supported(kwargs)
    return kwargs, train_fn_name, overwrite


def check_supported(kw):
    """Options of the reference that this hot path does not implement fail HERE instead of silently training a different
    model.  Nothing is refused at present: --clip_gradient_norm (model_base.py:394-401) is per-variable tf.clip_by_norm
    in front of the optimiser (optim.GradClip), --rnn_name LN_LSTM / GRU (model_base.py:622-629) run on the per-step
    launch chain."""
    bad = []
    # --initialiser: `he` / `none` select TensorFlow's default initialiser in the reference (model_base.py:823-831:
    # every value but `xavier` returns None), which for these float variables is glorot_uniform == Xavier-uniform
    # [TF-1.9 get_variable default]: all three choices build the same model, here too.
    if bad:
        raise NotImplementedError('not implemented on the MI355X hot path: ' + '; '.join(bad))


def main(argv=None):
    args = create_parser().parse_args(argv)
    import torch
    import torch.distributed as dist
    world = int(os.environ.get('WORLD_SIZE', '1'))
    local_rank = int(os.environ.get('LOCAL_RANK', str(args.gpu).split(',')[0] if world == 1 else '0'))
    # COMIC_DIST_BACKEND=gloo: data-parallel rehearsal on fewer GPUs than ranks (the ranks share the visible devices)
    backend = os.environ.get('COMIC_DIST_BACKEND', 'nccl')
    if backend != 'nccl':
        local_rank %= max(1, torch.cuda.device_count())
        if int(os.environ.get('LOCAL_WORLD_SIZE', str(world))) > torch.cuda.device_count():
            # ranks sharing a GPU cannot both keep a persistent loop's workgroups resident: per-step launches (bench.py)
            os.environ.setdefault('COMIC_PERSIST', '0')
    torch.cuda.set_device(local_rank)
    device = 'cuda:%d' % local_rank
    if world > 1:
        if backend == 'nccl':
            dist.init_process_group('nccl', device_id=torch.device(device))
        else:
            dist.init_process_group(backend)
    # every rank decides "new run or resume" from the SAME state of the experiments directory: the decision is taken
    # (build_kwargs looks at log_path) before any rank may create it (try_to_train, behind this barrier)
    kwargs, train_fn_name, overwrite = build_kwargs(args)
    if world > 1:
        dist.barrier()
    from comic_amd import train_fn as train
    from comic_amd.trainer import DataParallel
    dp = None
    if world > 1:
        dp = DataParallel(dist)
        # rand_seed stays the SAME on every rank: it seeds the parameter initialisers and the common shuffle; the
        # input managers shard the shuffled list by rank (config.dp_world / dp_rank)
        kwargs['dp_world'], kwargs['dp_rank'] = dp.world, dp.rank
    fn = getattr(train, train_fn_name)
    train.try_to_train(train_fn=lambda cfg: fn(cfg, device=device, dp=dp), try_block=True, overwrite=overwrite, **kwargs)
    if world > 1:
        dist.destroy_process_group()


if __name__ == '__main__':
    main()
