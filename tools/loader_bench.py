"""Loader throughput on 640x480 JPEGs: all-numpy, thread decode + device preprocessing, process decode (DecodePool) + device
preprocessing, split decode (entropy decoding on C threads, pixels on the device: JpegSplitPool) + device preprocessing; the
output tensors of the last three are compared bit for bit.  FILES=photo re-encodes two camera photographs (scikit-learn's
sample images) at 640x480, quality 90, 4:2:0 -- 106 / 64 KB files, MS-COCO-like -- instead of the smooth synthetic set."""
import sys, os, time, tempfile, random
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), '..'))
import numpy as np, torch
from PIL import Image
from concurrent.futures import ThreadPoolExecutor
from comic_amd import inputs
d = tempfile.mkdtemp()
rng = np.random.default_rng(0)
paths = []
if os.environ.get('FILES', 'smooth') == 'photo':
    import sklearn
    sd = os.path.join(os.path.dirname(sklearn.__file__), 'datasets', 'images')
    photos = [Image.open(os.path.join(sd, f)).convert('RGB') for f in ('china.jpg', 'flower.jpg')]
    for i in range(256):
        # (every file its own crop: no two files share their bytes)
        im = photos[i % 2].crop((i % 40, i % 27, 600 + i % 40, 400 + i % 27)).resize((640, 480), Image.BICUBIC)
        p = os.path.join(d, '%d.jpg' % i); im.save(p, quality=90, subsampling=2); paths.append(p)
else:
    for i in range(256):
        a = (rng.random((30, 40, 3)) * 255).astype(np.uint8)
        p = os.path.join(d, '%d.jpg' % i); Image.fromarray(a).resize((640, 480), Image.BICUBIC).save(p, quality=90); paths.append(p)
print('files: %s, %.0f KB on average' % (os.environ.get('FILES', 'smooth'), sum(os.path.getsize(p) for p in paths) / len(paths) / 1024))
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
    pre.unregister_shm()
    pool.close()
for nt in [int(v) for v in os.environ.get('SPLIT_THREADS', '8,16').split(',') if v]:
    jpool = inputs.JpegSplitPool(nt, max_batch=64)
    pre2 = inputs.DevicePreprocessor('cuda:0', 224, 224)
    pre2.enable_split(jpool, 6)
    got = pre2.finish(pre2.pack_paths_split(paths[:64], [(False, 16, 16)] * 64)).cpu()
    assert torch.equal(got, ref), 'split decode differs from thread decode'
    n, t0 = 0, time.time()
    inflight = []
    for rep in range(40):
        for b in range(0, len(paths), 64):
            ps = paths[b:b + 64]
            params = [inputs.draw_augmentation(True, 224, 224, r) for _ in ps]
            inflight.append(pre2.pack_paths_split(ps, params))
            if len(inflight) > 3:
                t = pre2.finish(inflight.pop(0))
            n += len(ps)
    while inflight:
        t = pre2.finish(inflight.pop(0))
    torch.cuda.synchronize()
    print('split threads %2d device : %6.0f images/s' % (nt, n / (time.time() - t0)))
    # where the time goes: host threads alone (no device work), and the device half alone
    import ctypes as C
    from comic_amd import _lib as L
    bufs = [(np.zeros(64, L.JPEG_INFO_DTYPE), np.zeros(64, np.int32), torch.empty(64 * jpool.slot_elems, dtype=torch.int16).pin_memory())
            for _ in range(4)]
    hs = []
    for rep in range(-1, 24):
        if rep == 0:                     # (round -1 touches every page of the fresh pinned buffers)
            for h in hs:
                jpool.wait(h)
            hs, t0 = [], time.time()
        for k, b in enumerate(range(0, len(paths), 64)):
            if len(hs) == 4:
                jpool.wait(hs.pop(0))
            infos, status, coef = bufs[(rep * 4 + k) % 4]
            hs.append(jpool.submit(paths[b:b + 64], infos.ctypes.data, status.ctypes.data, coef.data_ptr(), coef.numel()))
    for h in hs:
        jpool.wait(h)
    print('split threads %2d host half alone : %6.0f images/s' % (nt, 24 * len(paths) / (time.time() - t0)))
    jpool.close()
print('cpus', os.cpu_count(), 'usable', len(os.sched_getaffinity(0)))
