"""One rank of the data-parallel GPU rehearsals of tests/test_gpu_dp.py (launched by torch.distributed.run with
COMIC_DIST_BACKEND=gloo: two ranks share ONE GPU, gloo moves the CUDA tensors through the host).

  dp_worker.py cli  <out_dir> <train.py args...>   the reference CLI path: src/train.py -> try_to_train -> train_fn with a
                                                   DataParallel; every rank then writes its decoder parameters
  dp_worker.py step <out_dir> <steps>              CaptionTrainer.xe_step on this rank's rows of a fixed global batch
                                                   (no dropout): the rank-mean step == the single-process step
"""
import importlib.util
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def fixed_problem(B, size=139):
    """Global batch of the `step` rehearsal (same on every rank and in the parent test)."""
    from comic_amd import decoder as cdec, nets
    rng = np.random.default_rng(77)
    plan = nets.CnnPlan('inception_v3', (size, size))
    cnn_p = plan.init_params(3)
    spec = cdec.DecoderSpec(D=128, E=64, V=258, C=2048, Cg=2048, H=8, M=9)
    images = rng.uniform(-1, 1, (B, size, size, 3)).astype(np.float32)
    caps = np.full((B, 9), -1, np.int64)
    for b in range(B):
        n = int(rng.integers(2, 7))
        caps[b, 0], caps[b, 1:1 + n], caps[b, 1 + n] = 256, rng.integers(0, 256, n), 257
    return plan, cnn_p, spec, images, caps


def run_steps(dp, device, steps, B_global=8):
    import torch
    from comic_amd import trainer
    plan, cnn_p, spec, images, caps = fixed_problem(B_global)
    lo, hi = dp.shard(B_global)
    tr = trainer.CaptionTrainer(cnn_p, spec, None, hi - lo, (139, 139), 'f32', device, dp=dp, seed=5, plan=plan)
    x = torch.from_numpy(images[lo:hi]).to(device)
    losses = []
    for _ in range(steps):
        res = tr.xe_step(x, caps[lo:hi], training=False)       # no dropout: the only rank dependence is the shard
        losses.append(float(res['loss']))
    torch.cuda.synchronize()
    return tr.decoder.params.data.cpu().numpy().copy(), losses


def main():
    mode, out_dir = sys.argv[1], sys.argv[2]
    rank = int(os.environ.get('RANK', '0'))
    os.makedirs(out_dir, exist_ok=True)
    if mode == 'cli':
        from comic_amd import model as mdl
        made = []
        init = mdl.CaptionModel.__init__

        def recording_init(self, *a, **k):
            init(self, *a, **k)
            made.append(self)
        mdl.CaptionModel.__init__ = recording_init
        spec = importlib.util.spec_from_file_location('cli_train_dp', os.path.join(ROOT, 'src', 'train.py'))
        cli = importlib.util.module_from_spec(spec)
        spec.loader.exec_module(cli)
        cli.main(sys.argv[3:])
        train = [m for m in made if m.mode == 'train']
        assert train, 'train_fn built no training model'
        np.save(os.path.join(out_dir, 'params_rank%d.npy' % rank), train[0].decoder.params.data.cpu().numpy())
        np.save(os.path.join(out_dir, 'step_rank%d.npy' % rank), np.array([train[0].global_step]))
        return
    import torch
    import torch.distributed as dist
    from comic_amd.trainer import DataParallel
    torch.cuda.set_device(0)
    dist.init_process_group(os.environ.get('COMIC_DIST_BACKEND', 'gloo'))
    params, losses = run_steps(DataParallel(dist), 'cuda:0', int(sys.argv[3]))
    np.save(os.path.join(out_dir, 'params_rank%d.npy' % rank), params)
    np.save(os.path.join(out_dir, 'loss_rank%d.npy' % rank), np.array(losses))
    dist.destroy_process_group()


if __name__ == '__main__':
    main()
