"""Data-parallel path on real kernels without a multi-GPU node: two ranks share cuda:0 and exchange through gloo
(COMIC_DIST_BACKEND=gloo; the driver's scaling runs use nccl = RCCL).  Covers what tests/test_dp_gloo.py cannot on the
CPU: the reference CLI harness (train.py -> train_fn) under a DataParallel, bench.py's world > 1 branch, and the
rank-mean step against the single-process step of the global batch with the HIP decoder in the loop."""
import glob
import json
import os
import socket
import subprocess
import sys

import numpy as np
import pytest

pytestmark = pytest.mark.gpu

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _port():
    s = socket.socket()
    s.bind(('127.0.0.1', 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _launch(script_args, nproc=2, timeout=900, extra_env=None):
    env = dict(os.environ, COMIC_DIST_BACKEND='gloo', HSA_ENABLE_IPC_MODE_LEGACY='0', MASTER_ADDR='127.0.0.1')
    env.update(extra_env or {})
    cmd = [sys.executable, '-m', 'torch.distributed.run', '--nnodes=1', '--nproc-per-node', str(nproc),
           '--master-addr', '127.0.0.1', '--master-port', str(_port())] + script_args
    r = subprocess.run(cmd, cwd=ROOT, env=env, capture_output=True, text=True, timeout=timeout)
    if r.returncode != 0:                    # pytest shows the captured output of a failing test in full
        print(r.stdout[-6000:])
        print(r.stderr[-12000:])
    assert r.returncode == 0, 'torch.distributed.run exited with %d (output above)' % r.returncode
    return r


def test_two_ranks_through_train_fn_end_with_equal_parameters(tmp_path):
    """train.py under torch.distributed.run, world 2: both ranks walk the same shuffled epoch (disjoint shards), reduce
    the XE denominator and the flat gradient every step, and must end with bit-equal decoder parameters; only rank 0
    writes config.pkl and checkpoints."""
    from tests import tiny_dataset
    ds = tiny_dataset.make(str(tmp_path / 'mscoco'), n_train=24, n_valid=4, n_test=4)
    logs, out = str(tmp_path / 'experiments'), str(tmp_path / 'out')
    _launch([os.path.join(ROOT, 'tests', 'dp_worker.py'), 'cli', out, '--dataset_dir', ds, '--log_root', logs,
             '--cnn_name', 'inception_v3', '--cnn_fm_attention', 'Mixed_7c', '--cnn_input_size', '139,139',
             '--batch_size_eval', '4', '--rnn_size', '128', '--rnn_word_size', '64', '--train_mode', 'decoder',
             '--batch_size_train', '4', '--max_epoch', '1',
             '--loader_split_jpeg', '--loader_threads', '2'])       # every rank with its own pool of decode threads
    assert not glob.glob(os.path.join(logs, 'mscoco', 'error__*')), open(glob.glob(os.path.join(logs, 'mscoco', 'error__*'))[0]).read()
    p0, p1 = (np.load(os.path.join(out, 'params_rank%d.npy' % r)) for r in (0, 1))
    s0, s1 = (int(np.load(os.path.join(out, 'step_rank%d.npy' % r))[0]) for r in (0, 1))
    assert s0 == s1 == 15                                  # 24 images x 5 captions / (2 ranks x 4): GLOBAL batches of one epoch
    assert np.isfinite(p0).all() and np.array_equal(p0, p1)
    run_dir = os.path.join(logs, 'mscoco', 'radix_b256_add_LN_softmax_h8_tie_lstm_run_01')
    assert os.path.isfile(os.path.join(run_dir, 'config.pkl')) and glob.glob(os.path.join(run_dir, 'model_compact-*.npz'))


def test_rank_mean_step_equals_single_process_global_batch_step(tmp_path):
    """Three XE steps (InceptionV3 fp32 forward, HIP decoder step, device-side global XE denominator, gradient
    all-reduce, TF-Adam) on two ranks holding 4 rows each == the same steps in one process on all 8 rows, to fp32
    summation order; the two ranks are bit-equal."""
    import torch
    from tests import dp_worker
    from comic_amd.trainer import DataParallel
    out = str(tmp_path / 'out')
    _launch([os.path.join(ROOT, 'tests', 'dp_worker.py'), 'step', out, '3'])
    p0, p1 = (np.load(os.path.join(out, 'params_rank%d.npy' % r)) for r in (0, 1))
    assert np.array_equal(p0, p1)
    want, losses = dp_worker.run_steps(DataParallel(None), 'cuda:0', 3)
    torch.cuda.synchronize()
    scale = np.abs(want).max()
    assert np.abs(p0 - want).max() <= 2e-5 * scale, np.abs(p0 - want).max() / scale
    assert not np.array_equal(want, dp_worker.run_steps(DataParallel(None), 'cuda:0', 1)[0])     # the steps did train
    l0, l1 = (np.load(os.path.join(out, 'loss_rank%d.npy' % r)) for r in (0, 1))
    assert np.isfinite(l0).all() and np.isfinite(l1).all() and len(l0) == len(losses) == 3


def test_bench_world_two_branch_runs(tmp_path):
    """bench.py --gpus 2 as the driver launches it (torch.distributed.run, one rank per process), rehearsed on one GPU
    over gloo: the world > 1 branch (broadcast, per-step all-reduces, MAX over ranks of the timed region) executes and
    rank 0 prints ONE JSON line with n_gpus 2 and the whole-job image rate.  (Per-step decoder launches here: the
    persistent time loops want every CU of a GPU for one process -- two ranks sharing one card is a rehearsal, not a
    deployment.)"""
    r = _launch([os.path.join(ROOT, 'bench.py'), '--gpus', '2', '--steps', '4', '--warmup', '1', '--no-cpu-baseline',
                 '--no-extras'], timeout=1100, extra_env={'COMIC_AUTOTUNE': '0', 'COMIC_PERSIST': '0'})
    lines = [l for l in r.stdout.splitlines() if l.startswith('{"metric"')]
    assert len(lines) == 1, r.stdout[-2000:]
    j = json.loads(lines[0])
    assert j['n_gpus'] == 2 and j['config']['global_batch'] == 128 and j['scaling'] == 'weak'
    assert j['value'] > 0 and abs(j['value'] - 128 * 4 / (j['ms_per_step'] * 4e-3)) < 1e-3 * j['value']
    assert np.isfinite(j['final_loss'])


def test_bench_gpus_two_launches_itself(tmp_path):
    """`python bench.py --gpus 2 ...` with NO launcher and no WORLD_SIZE in the environment: bench.py starts its own
    torch.distributed.run child before anything touches the GPU, relays rank 0's single JSON line on stdout and leaves
    with the child's exit code (the shape of the driver's 1-GPU command with another N)."""
    env = {k: v for k, v in os.environ.items() if k not in ('WORLD_SIZE', 'RANK', 'LOCAL_RANK', 'MASTER_ADDR', 'MASTER_PORT')}
    env.update(COMIC_DIST_BACKEND='gloo', COMIC_AUTOTUNE='0', COMIC_PERSIST='0', HSA_ENABLE_IPC_MODE_LEGACY='0')
    r = subprocess.run([sys.executable, os.path.join(ROOT, 'bench.py'), '--gpus', '2', '--steps', '4', '--warmup', '1',
                        '--no-extras', '--no-cpu-baseline'], cwd=ROOT, env=env, capture_output=True, text=True, timeout=1100)
    if r.returncode != 0:
        print(r.stdout[-4000:])
        print(r.stderr[-12000:])
    assert r.returncode == 0
    lines = [l for l in r.stdout.splitlines() if l.startswith('{"metric"')]
    assert len(lines) == 1, r.stdout[-2000:]
    j = json.loads(lines[0])
    assert j['n_gpus'] == 2 and j['config']['global_batch'] == 128 and j['value'] > 0 and np.isfinite(j['final_loss'])
    assert list(j)[-1] == 'summary' and j['summary']['voided_steps'] == 0
