"""Loader throughput on 640x480 JPEGs: all-numpy, thread decode + device preprocessing, process decode (DecodePool) + device
preprocessing (the output tensors of the last two are compared bit for bit)."""
import sys, os, time, tempfile, random
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), '..'))
import numpy as np, torch
from PIL import Image
from concurrent.futures import ThreadPoolExecutor
from comic_amd import inputs
d = tempfile.mkdtemp()
rng = np.random.default_rng(0)
paths = []
for i in range(256):
    a = (rng.random((30, 40, 3)) * 255).astype(np.uint8)
    p = os.path.join(d, '%d.jpg' % i); Image.fromarray(a).resize((640, 480), Image.BICUBIC).save(p, quality=90); paths.append(p)
r = random.Random(0)
pre = inputs.DevicePreprocessor('cuda:0', 224, 224)
for nt in [int(v) for v in os.environ.get('THREADS', '1,4,16').split(',') if v]:
    pool = ThreadPoolExecutor(max_workers=nt)
    for mode in ('numpy', 'device'):
        n, t0 = 0, time.time()
        for rep in range(2 if mode == 'numpy' else 6):
            for b in range(0, len(paths), 64):
                ps = paths[b:b + 64]
                params = [inputs.draw_augmentation(True, 224, 224, r) for _ in ps]
                if mode == 'numpy':
                    ims = np.stack(list(pool.map(lambda a: inputs.preprocess_image(a[0], 224, 224, True, r, a[1]), zip(ps, params))))
                    t = torch.from_numpy(ims).to('cuda:0')
                else:
                    t = pre(list(pool.map(inputs.decode_image, ps)), params)
                n += len(ps)
        torch.cuda.synchronize()
        print('threads %2d  %-6s : %6.0f images/s' % (nt, mode, n / (time.time() - t0)))
ref = pre(list(map(inputs.decode_image, paths[:64])), [(False, 16, 16)] * 64).cpu()
for nproc in [int(v) for v in os.environ.get('NPROCS', '8,16').split(',') if v]:
    pool = inputs.DecodePool(nproc)
    got = pre.finish(pre.pack_paths(pool, paths[:64], [(False, 16, 16)] * 64)).cpu()
    assert torch.equal(got, ref), 'process decode differs from thread decode'
    n, t0 = 0, time.time()
    inflight = []                        # as the loader's prefetch thread does: a few batches decode at once
    for rep in range(12):
        for b in range(0, len(paths), 64):
            ps = paths[b:b + 64]
            params = [inputs.draw_augmentation(True, 224, 224, r) for _ in ps]
            inflight.append(pre.pack_paths(pool, ps, params))
            if len(inflight) > 3:
                t = pre.finish(inflight.pop(0))
            n += len(ps)
    while inflight:
        t = pre.finish(inflight.pop(0))
    torch.cuda.synchronize()
    print('processes %2d device : %6.0f images/s' % (nproc, n / (time.time() - t0)))
    pool.close()
print('cpus', os.cpu_count(), 'usable', len(os.sched_getaffinity(0)))
