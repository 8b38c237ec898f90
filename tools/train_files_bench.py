"""Training throughput FROM FILES (decoder mode, COMIC-256 defaults, batch 64): the whole chain -- JPEG files -> loader ->
device preprocessing -> encoder -> decoder step -- with the loaders: decode threads (default), decode processes
(--loader_processes 16), split JPEG decode (--loader_split_jpeg, 16 C threads), the same with its coefficient cache
(MODES=cache: --loader_cache_gb 4; the timed steps come after the first epoch, i.e. from the cache).  bench.py's headline uses synthetic images
already resident in HBM; this is the same step fed by the input pipeline.  640x480 quality-90 4:2:0 re-encodes of two
camera photographs (scikit-learn's sample images), 512 files."""
import importlib.util, os, sys, tempfile, time
ROOT = os.path.join(os.path.dirname(os.path.abspath(__file__)), '..')
sys.path.insert(0, ROOT)
import numpy as np, torch
from PIL import Image
from tests import tiny_dataset
from comic_amd import model as mdl, train_fn as train

tmp = tempfile.mkdtemp()
N = int(os.environ.get('FILES_N', '512'))
ds = tiny_dataset.make(os.path.join(tmp, 'mscoco'), n_train=N, n_valid=64, n_test=4)
import sklearn
sd = os.path.join(os.path.dirname(sklearn.__file__), 'datasets', 'images')
photos = [Image.open(os.path.join(sd, f)).convert('RGB') for f in ('china.jpg', 'flower.jpg')]
for i in range(N):
    im = photos[i % 2].crop((i % 40, i % 27, 600 + i % 40, 400 + i % 27)).resize((640, 480), Image.BICUBIC)
    im.save(os.path.join(ds, 'images', 'COCO_train2014_%012d.jpg' % (i + 1)), quality=90, subsampling=2)
# captions of MS-COCO length (8 .. 14 words: 18 .. 30 radix tokens, the synthetic bench's distribution) instead of the tiny data
# set's 4 .. 8 words: the decoder step costs what it costs in bench.py
if os.environ.get('CAPTIONS', 'coco') == 'coco':
    rng = np.random.default_rng(1)
    words = ['a', 'man', 'dog', 'cat', 'on', 'the', 'table', 'sitting', 'red', 'bench', 'with', 'frisbee', 'park', 'two', 'people',
             'standing', 'near', 'train', 'street', 'sign']
    fp = os.path.join(ds, 'captions', 'mscoco_{}_w5_s20_include_restval'.format('train') + '.txt')
    lines = []
    for i in range(N):
        rel = os.path.join('images', 'COCO_train2014_%012d.jpg' % (i + 1))
        for _ in range(5):
            lines.append('%s,<GO> %s <EOS>' % (rel, ' '.join(rng.choice(words, int(rng.integers(8, 15))))))
    with open(fp, 'w', newline='') as f:
        f.write('\r\n'.join(lines))
spec = importlib.util.spec_from_file_location('cli_train_bench', os.path.join(ROOT, 'src', 'train.py'))
cli = importlib.util.module_from_spec(spec)
spec.loader.exec_module(cli)
STEPS, WARM = int(os.environ.get('STEPS', '120')), int(os.environ.get('WARM', '60'))
modes = [m for m in os.environ.get('MODES', 'split,processes,threads').split(',') if m]
for mode in modes:
    extra = {'split': ['--loader_split_jpeg', '--loader_threads', '16'], 'processes': ['--no-loader_split_jpeg', '--loader_processes', '16'],
             'cache': ['--loader_split_jpeg', '--loader_threads', '16', '--loader_cache_gb', '4'],
             'threads': ['--no-loader_split_jpeg', '--loader_threads', '16']}[mode]
    args = cli.create_parser().parse_args(['--dataset_dir', ds, '--log_root', os.path.join(tmp, 'exp_' + mode), '--train_mode',
                                           'decoder', '--batch_size_train', '64', '--batch_size_eval', '64', '--max_epoch', '50'] + extra)
    kwargs, _, overwrite = cli.build_kwargs(args)

    def probe(config):
        mdl.reset_default_graph()
        man = train._manager(config)
        try:
            man.enable_device_preprocess('cuda:0')
            m = mdl.CaptionModel(config, mode='train', batch_ops=man.batch_train, reuse=False, name='train', device='cuda:0')
            steps = STEPS if mode != 'threads' else max(20, STEPS // 4)
            for _ in range(WARM if mode != 'threads' else 25):
                m.run_train_step()
            torch.cuda.synchronize()
            acc = {'finish': 0.0, 'n': 0}
            if os.environ.get('PROFILE') == '1':          # host time of the loader's consumer half, per batch
                inner = man._devpre.finish

                def timed(packed):
                    a = time.perf_counter(); r = inner(packed); acc['finish'] += time.perf_counter() - a; acc['n'] += 1
                    return r
                man._devpre.finish = timed
            prof = None
            if os.environ.get('CPROFILE') == '1':
                import cProfile
                prof = cProfile.Profile()
                prof.enable()
            t0 = time.time()
            for _ in range(steps):
                loss = m.run_train_step()
            t_issue = time.time() - t0
            if prof is not None:
                import pstats
                prof.disable()
                pstats.Stats(prof).sort_stats('tottime').print_stats(22)
            torch.cuda.synchronize()
            dt = time.time() - t0
            if acc['n']:
                print('  host: %.3f ms per step to issue (of which %.3f ms in DevicePreprocessor.finish), %.3f ms per step in all'
                      % (t_issue / steps * 1e3, acc['finish'] / max(acc['n'], 1) * 1e3, dt / steps * 1e3), flush=True)
            print('loader %-9s : %7.0f images/s  (%.2f ms per step of 64 images, loss %.4f)'
                  % (mode, steps * 64 / dt, dt / steps * 1e3, float(loss)), flush=True)
        finally:
            man.close()
    train.try_to_train(train_fn=probe, try_block=False, overwrite=overwrite, **kwargs)
