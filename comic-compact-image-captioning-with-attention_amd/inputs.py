"""Input managers with the reference's interface (`common/inputs/manager_image_caption.py`):
`InputManager`, `InputManager_Radix`, `InputManager_Char`, `InputManager_SCST` exposing
`.config` (adds itow / wtoi / vocab_size / split_sizes / max_step, :55,:104-108,:132,:141),
`.batch_train / .batch_eval / .batch_infer` (here: Python iterators of numpy batches),
`.filenames_infer` and `captions_to_batched_ids`.

Files are consumed unchanged: `{dataset_dir}/captions/{pattern.format(split)}.txt`
(`relpath,<GO> w1 ... <EOS>`), `{pattern.format('wtoi'|'itow')}.json`,
`filenames_{valid,test}.txt` (manager_image_caption.py:75-80,:98-108,:129-131).

Image path (SURVEY §8f 'next'): decode (PIL) -> float [0,1] -> TF-1 bilinear resize to
256x256 (align_corners=False) -> train: random flip + random crop / eval: central crop ->
scale to [-1,1] (inception_preprocessing_radix.py:191-199,:229-234,:269-271), on host threads.
"""
from __future__ import annotations

import json
import os
import random
import string
import queue
import threading
from concurrent.futures import ThreadPoolExecutor

import numpy as np

from . import ops

pjoin = os.path.join


def resize_bilinear_tf1(img, out_h, out_w):
    """tf.image.resize_bilinear(..., align_corners=False) [TF-1.9 resize_bilinear_op.cc]: src = dst * in/out, float32
    interpolation a + (b - a) * w (host path for CPU-only use; the training path runs csrc/preprocess.hip)."""
    f = np.float32
    in_h, in_w = img.shape[:2]
    ys = np.arange(out_h, dtype=np.float32) * f(in_h / f(out_h))
    xs = np.arange(out_w, dtype=np.float32) * f(in_w / f(out_w))
    y0 = np.floor(ys).astype(np.int64); x0 = np.floor(xs).astype(np.int64)
    y1 = np.minimum(np.ceil(ys).astype(np.int64), in_h - 1); x1 = np.minimum(np.ceil(xs).astype(np.int64), in_w - 1)
    wy = (ys - y0.astype(np.float32))[:, None, None]; wx = (xs - x0.astype(np.float32))[None, :, None]
    top = img[y0][:, x0] + (img[y0][:, x1] - img[y0][:, x0]) * wx
    bot = img[y1][:, x0] + (img[y1][:, x1] - img[y1][:, x0]) * wx
    return top + (bot - top) * wy


class Prefetch(object):
    """Iterator over `gen` that a daemon thread keeps `depth` items ahead of the consumer (the
    `dataset.prefetch` of the reference's tf.data pipeline): decoding / resizing of the next batches
    overlaps the GPU step.  Order is the generator's order; an exception in the producer is re-raised
    at the consumer's next() call."""
    _END = object()

    def __init__(self, gen, depth=4, finish=None):
        # finish(item) -> item runs in the CONSUMER thread at next() (e.g. the device half of the preprocessing:
        # the producer thread never touches the GPU runtime)
        self._finish = finish
        self._closed = False
        self._q = queue.Queue(maxsize=max(1, int(depth)))
        self._t = threading.Thread(target=self._run, args=(gen,), daemon=True)
        self._t.start()

    def _put(self, entry):
        while not self._closed:
            try:
                self._q.put(entry, timeout=0.05)
                return True
            except queue.Full:
                pass
        return False

    def _run(self, gen):
        try:
            for item in gen:
                if not self._put((item, None)):
                    return
            self._put((self._END, None))
        except BaseException as e:           # noqa: B902 -- handed to the consumer
            self._put((self._END, e))

    def close(self):
        """Stops the producer (an endless training generator would otherwise stay parked on a full queue, holding
        its batches, for the life of the process)."""
        self._closed = True
        if self._t is not threading.current_thread():
            self._t.join(timeout=10.0)       # the producer leaves its put() within one poll interval
        while True:
            try:
                self._q.get_nowait()
            except queue.Empty:
                break
        try:
            self._q.put_nowait((self._END, None))
        except queue.Full:
            pass

    def __iter__(self):
        return self

    def __next__(self):
        item, err = self._q.get()
        if item is self._END:
            self._q.put((self._END, err))    # stay exhausted
            if err is not None:
                raise err
            raise StopIteration
        return self._finish(item) if self._finish is not None else item


def draw_augmentation(augment, height, width, rng):
    """(flip, oy, ox) of one image, drawn in the PRODUCER thread so that the augmentation stream is a function of
    the seed and the sample order, not of how the decode threads are scheduled."""
    if not augment:
        return False, (256 - height) // 2, (256 - width) // 2
    flip = rng.random() < 0.5
    return flip, rng.randrange(0, 256 - height + 1), rng.randrange(0, 256 - width + 1)


def decode_image(path_or_array):
    """JPEG / PNG file (or an already decoded array) -> uint8 RGB [H, W, 3].  The only per-image host work when the
    rest of the preprocessing runs on the device (DevicePreprocessor)."""
    if isinstance(path_or_array, np.ndarray):
        return path_or_array
    from PIL import Image
    with Image.open(path_or_array) as im:
        return np.asarray(im.convert('RGB'))


class DecodePool(object):
    """Multi-process JPEG decode into shared-memory staging blocks (config.loader_processes > 0).

    The producer thread reads the image sizes from the file headers, lays the batch out back to back in a free block
    and hands (path, block, offset) tasks to `spawn`ed workers (`_decode_worker.decode_into`), which write the decoded
    pixels in place; the consumer registers each block once as pinned host memory (cudaHostRegister) and copies from
    it directly.  Blocks return to the free list when the event behind their copy has completed."""

    def __init__(self, processes, blocks=6, block_bytes=None, slot_bytes=640 * 640 * 3, max_batch=64, timeout_s=120.0):
        import multiprocessing as mp
        self.timeout_s = float(timeout_s)                     # config.loader_timeout_s: bound of the wait for one batch
        self._retired = []                                    # blocks surviving workers may still write into
        self.slot_bytes = int(slot_bytes)                     # MS-COCO images are at most 640 x 640
        block_bytes = int(block_bytes or self.slot_bytes * max_batch)
        from multiprocessing import shared_memory
        # `spawn` re-imports the parent's __main__ in every worker (train.py, pytest, a notebook ...): the workers need
        # none of it, so the main module is hidden from the preparation data while they start
        import sys
        main = sys.modules.get('__main__')
        saved = (getattr(main, '__file__', None), getattr(main, '__spec__', None))
        try:
            if main is not None:
                main.__file__, main.__spec__ = None, None
            self._pool = mp.get_context('spawn').Pool(int(processes))
        finally:
            if main is not None:
                main.__file__, main.__spec__ = saved
        self._blocks = [shared_memory.SharedMemory(create=True, size=int(block_bytes)) for _ in range(int(blocks))]
        self._free = queue.Queue()
        for b in self._blocks:
            self._free.put(b)
        self.block_bytes = int(block_bytes)

    def decode_batch(self, paths):
        """-> (block, [(offset, h, w)], bytes spanned); waits for the pixels."""
        blk, off, res = self.decode_batch_async(paths)
        return blk, self.geometry(paths, res, blk), off

    def decode_batch_async(self, paths):
        """-> (block, bytes spanned, async result).  Every image gets a fixed slot of `slot_bytes` in a free staging
        block (no header parsing in the producer thread: its Python work per image capped the loader before), the
        workers decode in place while the caller goes on (the loader keeps `loader_prefetch` batches in flight);
        `geometry(paths, result)` waits and returns [(offset, h, w)].  Blocks only when every block is in flight."""
        from . import _decode_worker as W
        n = len(paths)
        if n * self.slot_bytes > self.block_bytes:
            raise ValueError('batch of %d images x %d bytes exceeds the decode staging block (%d)'
                             % (n, self.slot_bytes, self.block_bytes))
        blk = self._free.get()
        res = self._pool.map_async(W.decode_into, [(p, blk.name, i * self.slot_bytes, self.slot_bytes)
                                                   for i, p in enumerate(paths)])
        return blk, n * self.slot_bytes, res

    def geometry(self, paths, res, blk=None, timeout=None):
        """Waits for the workers of one batch.  multiprocessing.Pool silently replaces a worker that dies (out of
        memory, a crash inside libjpeg on a corrupt file) and the task it held never completes: the wait is bounded,
        the batch fails with the file names instead of hanging, and its staging block is RETIRED, not returned to the
        free list -- workers of that batch that are still alive may yet write pixels into it, which must not land in
        the next batch's images (a fresh block takes its place, so the pool keeps its depth)."""
        import multiprocessing as mp
        timeout = self.timeout_s if timeout is None else timeout
        try:
            sizes = res.get(timeout=timeout)
        except mp.TimeoutError:
            if blk is not None:
                from multiprocessing import shared_memory
                self._retired.append(blk)
                fresh = shared_memory.SharedMemory(create=True, size=self.block_bytes)
                self._blocks.append(fresh)
                self._free.put(fresh)
            raise RuntimeError('JPEG decode workers did not return within %.0f s (a worker process died?) for: %s'
                               % (timeout, ', '.join(str(p) for p in paths[:4]) + (' ...' if len(paths) > 4 else '')))
        out = []
        for i, (p, (h, w)) in enumerate(zip(paths, sizes)):
            if h < 0:
                raise ValueError('%s decodes to %dx%d: larger than the loader slot of %d bytes (config.loader_slot_bytes)'
                                 % (p, -h, -w, self.slot_bytes))
            out.append((i * self.slot_bytes, h, w))
        return out

    def release(self, blk):
        self._free.put(blk)

    def close(self):
        self._pool.terminate()
        self._pool.join()
        for b in self._blocks:
            try:
                b.unlink()               # the name goes now; the pages when the last mapping does
            except FileNotFoundError:
                pass
            try:
                b.close()
            except BufferError:          # a pinned host tensor of the consumer still maps the block
                pass
        self._blocks = []


class JpegSplitPool(object):
    """Split JPEG decode (config.loader_split_jpeg): host threads undo the entropy coding only, the device does the rest.

    A persistent pool of C threads (libcomic_jpeg.so, include/comic_jpeg.h: no interpreter work per image, no GPU runtime)
    turns the files of a batch into quantised DCT coefficients in PACKED form (a descriptor and the DC value per block, a
    16-bit entry per non-zero AC coefficient: 3-4x fewer bytes than the dense blocks), written into a pinned staging slot
    (one host-to-device copy per batch); inverse DCT,
    chroma upsampling and colour conversion run on the device in libjpeg's integer arithmetic (comic_jpeg_pixels), so the
    RGB bytes are PIL's.  Stands where the reference's tf.data map decodes on host cores
    (common/inputs/manager_image_caption.py:163-175).  Files the split decoder does not take (progressive, CMYK, ...) are
    decoded by PIL when their batch is finished."""

    def __init__(self, threads, slot_elems=640 * 640 * 3 // 2, max_batch=64, timeout_s=120.0, cache_gb=0.0):
        from . import _lib as L
        self.L, self.lib = L, L.load_jpeg()
        self.threads = int(threads)
        # coefficient budget per image: a staging slot holds max_batch x slot_elems elements, the images of a batch back to
        # back (640 x 640 at 4:2:0 is 614 400; images the rest of a slot cannot take go through PIL)
        self.slot_elems = (int(slot_elems) + 7) // 8 * 8
        self.max_batch = int(max_batch)
        self.timeout_s = float(timeout_s)
        self._pool = self.lib.comic_jpeg_pool_create(self.threads)
        if not self._pool:
            raise RuntimeError('comic_jpeg_pool_create(%d) failed' % self.threads)
        self._lock = threading.Lock()              # submit() of a loader thread against close() of the main thread
        self.closing = False                       # set at the start of a stage's shutdown: loader threads parked on staging leave
        # config.loader_cache_gb: the decoded coefficients of every image stay in host memory (as their non-zeros: about the
        # size of the file) up to this many GB -- the epochs after the first skip file reads and Huffman decoding
        if cache_gb and cache_gb > 0:
            L.check(self.lib.comic_jpeg_pool_enable_cache(self._pool, int(float(cache_gb) * (1 << 30))), 'jpeg_pool_enable_cache')

    def submit(self, paths, infos_ptr, status_ptr, coef_ptr, capacity):
        import ctypes as C
        arr = (C.c_char_p * len(paths))(*[os.fsencode(p) for p in paths])
        with self._lock:
            if not self._pool:
                raise RuntimeError('the JPEG decode pool is closed')
            h = self.lib.comic_jpeg_pool_submit_packed(self._pool, arr, len(paths), infos_ptr, status_ptr, coef_ptr, int(capacity))
        if not h:
            raise RuntimeError('comic_jpeg_pool_submit failed')
        return h

    def wait(self, handle, paths=()):
        """-> (16-bit units of the packed blob in use, bytes of the images' component planes back to back -- coef_base of the
        images assigned).  Bounded like DecodePool.geometry."""
        import ctypes as C
        used, total = C.c_int64(0), C.c_int64(0)
        rc = self.lib.comic_jpeg_pool_wait(self._pool, handle, self.timeout_s, C.byref(used), C.byref(total))
        if rc < 0:
            raise ValueError('comic_jpeg_pool_wait: bad arguments (pool closed or not a batch handle)')
        if rc != 0:
            raise RuntimeError('JPEG decode threads did not return within %.0f s for: %s'
                               % (self.timeout_s, ', '.join(str(p) for p in list(paths)[:4])))
        return int(used.value), int(total.value)

    @property
    def closed(self):
        return not self._pool

    def cache_stats(self):
        """-> (bytes in use, images cached, hits so far) of the coefficient cache."""
        import ctypes as C
        b, e, h = C.c_int64(0), C.c_int64(0), C.c_int64(0)
        with self._lock:
            if self._pool:
                self.lib.comic_jpeg_pool_cache_stats(self._pool, C.byref(b), C.byref(e), C.byref(h))
        return int(b.value), int(e.value), int(h.value)

    def close(self):
        with self._lock:
            pool, self._pool = self._pool, None
        if pool:
            self.lib.comic_jpeg_pool_destroy(pool)                # waits for queued work


class PackedImages(object):
    """A batch of decoded images packed back to back (+ one comic_image_desc per image) by the producer thread."""
    __slots__ = ('slot', 'blob', 'desc', 'n', 'total')

    def __init__(self, slot, blob, desc, n, total):
        self.slot, self.blob, self.desc, self.n, self.total = slot, blob, desc, n, total


class DevicePreprocessor(object):
    """resize / flip / crop / scale of a batch in ONE launch of comic_image_preprocess (csrc/preprocess.hip); the
    result is a device tensor [n, h, w, 3] fp32 with the bits of `preprocess_image`.

    Two halves, because the GPU runtime is only ever called from the consumer (training) thread -- concurrent
    allocation / copy / launch calls from a loader thread next to hipGraph captures and replays aborted the process
    intermittently:
      pack(images, params)   producer thread, CPU only: the decoded uint8 images back to back + their descriptors
                             into a pinned staging slot taken from the free list (plain numpy arrays while no slot
                             is free or large enough)
      finish(packed)         consumer thread: host-to-device copies + the launch; staging slots come back to the free
                             list once the event recorded behind their copies has completed (polled here)."""

    def __init__(self, device, height, width, resize=256, slots=8):
        import torch
        from . import _lib as L
        self.torch, self.L, self.lib = torch, L, L.load()
        self.device, self.h, self.w, self.resize = device, int(height), int(width), int(resize)
        self._max_slots = int(slots)
        self._n_slots = 0
        self._free = queue.Queue()
        self._pending = []                       # (slot, event) in issue order
        self._dev_blob = None

    # ---- producer side (no GPU calls) -------------------------------------------------------------------------
    def pack(self, images_u8, params):
        import ctypes as C
        L = self.L
        n = len(images_u8)
        total = sum(int(im.shape[0]) * int(im.shape[1]) * 3 for im in images_u8)
        dbytes = n * C.sizeof(L.ImageDesc)
        slot = None
        try:
            slot = self._free.get_nowait()
        except queue.Empty:
            pass
        if slot is not None and (slot['blob'].numel() < total or slot['desc'].numel() < dbytes):
            self._free.put(slot)                 # too small for this batch: the consumer will make a larger one
            slot = None
        if slot is not None:
            blob, dbuf = slot['blob'].numpy(), slot['desc'].numpy()
        else:
            blob, dbuf = np.empty(total, np.uint8), np.zeros(dbytes, np.uint8)
        desc = (L.ImageDesc * n).from_buffer(dbuf)
        off = 0
        for i, (im, (flip, oy, ox)) in enumerate(zip(images_u8, params)):
            assert im.dtype == np.uint8 and im.ndim == 3 and im.shape[2] == 3, (im.dtype, im.shape)
            ih, iw = int(im.shape[0]), int(im.shape[1])
            nb = ih * iw * 3
            blob[off:off + nb] = im.reshape(-1)
            d = desc[i]
            d.offset, d.in_h, d.in_w, d.flip, d.oy, d.ox = off, ih, iw, int(bool(flip)), int(oy), int(ox)
            d.sy, d.sx = np.float32(ih / self.resize), np.float32(iw / self.resize)
            off += nb
        del desc
        return PackedImages(slot, blob, dbuf, n, total)

    def pack_paths(self, pool, paths, params):
        """Producer half for a DecodePool: the worker processes decode straight into a shared-memory block (no copy in
        this thread); only the descriptors are filled here."""
        import ctypes as C
        L = self.L
        blk, total, pending = pool.decode_batch_async(paths)
        return PackedImages(('shm', pool, blk, pending, list(paths), list(params)), None, None, len(paths), total)

    # ---- split JPEG decode: coefficients from the host threads, pixels on the device ------------------------------------
    _DESC_DTYPE = np.dtype({'names': ['offset', 'in_h', 'in_w', 'flip', 'oy', 'ox', 'sy', 'sx'],
                            'formats': ['<i8', '<i4', '<i4', '<i4', '<i4', '<i4', '<f4', '<f4'],
                            'offsets': [0, 8, 12, 16, 20, 24, 28, 32], 'itemsize': 40})

    def enable_split(self, jpool, slots):
        """Consumer thread, once: the pinned staging slots the decode threads write coefficients into (max_batch x slot_elems
        elements each, the images of a batch back to back; the producer blocks while all of them are in flight) and the device
        buffers of one batch."""
        torch = self.torch
        self._jpool = jpool
        self._free_coef = queue.Queue()
        # every staging slot stays referenced HERE for the life of the preprocessor: the decode threads write into a slot until
        # comic_jpeg_pool_wait / _destroy has returned (include/comic_jpeg.h), which a batch dropped at a stage's shutdown
        # (Prefetch.close drains its queue) never waits for -- the slot's memory must not go back to an allocator before the
        # pool is gone
        self._all_coef = []
        n = jpool.max_batch
        for _ in range(int(slots)):
            slot = dict(coef=torch.empty(n * jpool.slot_elems, dtype=torch.int16).pin_memory(),
                        infos=torch.zeros(n * 512, dtype=torch.uint8).pin_memory(),
                        desc=torch.zeros(n * 40, dtype=torch.uint8).pin_memory(),
                        status=np.zeros(n, np.int32))
            self._all_coef.append(slot)
            self._free_coef.put(slot)
        with torch.cuda.device(self.device):
            self._dev_coef = torch.empty(n * jpool.slot_elems, dtype=torch.int16, device=self.device)
            self._dev_planes = torch.empty(n * jpool.slot_elems, dtype=torch.uint8, device=self.device)
            # (persistent: a `.to()` per batch cost the consumer thread 0.19 ms each -- allocation + the pinned-memory query)
            self._dev_infos = torch.empty(n * 512, dtype=torch.uint8, device=self.device)
            self._dev_desc = torch.empty(n * 40, dtype=torch.uint8, device=self.device)

    def pack_paths_split(self, paths, params):
        """Producer half: queue the files with the decode threads (which write into a pinned slot) and go on."""
        jpool = self._jpool
        if len(paths) > jpool.max_batch:
            raise ValueError('batch of %d images exceeds the split decoder\'s staging slots (%d)' % (len(paths), jpool.max_batch))
        import time
        t_end = time.monotonic() + jpool.timeout_s
        while True:                  # (short polls: a loader thread parked here must notice the end of its stage)
            try:
                slot = self._free_coef.get(timeout=0.05)
                break
            except queue.Empty:
                if jpool.closed or jpool.closing:
                    raise RuntimeError('the JPEG decode pool is closed')
                if time.monotonic() > t_end:
                    raise RuntimeError('no coefficient staging slot came back within %.0f s (consumer stalled?)' % jpool.timeout_s)
        handle = jpool.submit(paths, slot['infos'].data_ptr(), slot['status'].ctypes.data, slot['coef'].data_ptr(),
                              slot['coef'].numel())
        # files known (from an earlier batch) to need the PIL path start decoding now, on a few threads of their own, so that
        # the consumer finds their pixels ready instead of decoding them inside its step
        known = self.__dict__.setdefault('_pil_known', set())
        early = {i: self._pil_threads().submit(decode_image, p) for i, p in enumerate(paths) if p in known} if known else {}
        return PackedImages(('split', jpool, handle, slot, list(paths), list(params), early), None, None, len(paths), 0)

    def _pil_threads(self):
        pool = self.__dict__.get('_pil_pool')
        if pool is None:
            pool = self._pil_pool = ThreadPoolExecutor(max_workers=4)
        return pool

    def _finish_split(self, packed):
        import ctypes as C
        torch, L = self.torch, self.L
        _, jpool, handle, slot, paths, params = packed.slot[:6]
        n = packed.n
        try:
            used, pixel_bytes = jpool.wait(handle, paths)
        except RuntimeError:
            # threads of this batch may still write into the slot: it is retired with the batch, a fresh one takes its place
            fresh = dict(coef=torch.empty_like(slot['coef']).pin_memory(), infos=torch.zeros_like(slot['infos']).pin_memory(),
                         desc=torch.zeros_like(slot['desc']).pin_memory(), status=np.zeros_like(slot['status']))
            self._all_coef.append(fresh)             # (the retired slot stays in the list too)
            self._free_coef.put(fresh)
            raise
        try:
            return self._launch_split(packed, used, pixel_bytes)
        except Exception:
            self._free_coef.put(slot)             # (the wait has returned: no thread writes into it any more)
            raise

    def _launch_split(self, packed, used, pixel_bytes):
        torch, L = self.torch, self.L
        _, jpool, handle, slot, paths, params = packed.slot[:6]
        early = packed.slot[6] if len(packed.slot) > 6 else {}
        n = packed.n
        status = slot['status'][:n]
        infos = slot['infos'].numpy()[:n * 512].view(L.JPEG_INFO_DTYPE)
        ok = status == L.JPEG_OK
        h, w = infos['height'].astype(np.int64), infos['width'].astype(np.int64)
        off = infos['pixel_off'].astype(np.int64)
        # files the split decoder does not take: PIL; their RGB bytes go to the blob the preprocessing kernel reads for images
        # with ncomp == 0 (the images decoded on the device are never written as RGB)
        late = []
        pixel_planes, pixel_bytes = pixel_bytes, 0           # (the wait's second figure: bytes of the component planes)
        bad = [int(i) for i in np.nonzero(~ok)[0]]
        known = self.__dict__.setdefault('_pil_known', set())
        fresh = [i for i in bad if i not in early]
        if fresh:                                # first sight of these files: decode them side by side, remember the paths
            decoded = dict(zip(fresh, self._pil_threads().map(decode_image, [paths[i] for i in fresh])))
            # only files the split decoder can NEVER take; a slot that ran full (TOO_SMALL) or a read error says nothing about the
            # file's next visit
            known.update(paths[i] for i in fresh if isinstance(paths[i], str) and len(known) < 1_000_000
                         and int(status[i]) in (L.JPEG_UNSUPPORTED, L.JPEG_CORRUPT))
        for i in bad:
            im = early[i].result() if i in early else decoded[i]
            infos['ncomp'][i] = 0
            h[i], w[i], off[i] = im.shape[0], im.shape[1], pixel_bytes
            late.append((int(pixel_bytes), im))
            pixel_bytes += (im.size + 15) // 16 * 16
        desc = slot['desc'].numpy()[:n * 40].view(self._DESC_DTYPE)      # pinned: every copy of the batch is asynchronous
        desc['offset'], desc['in_h'], desc['in_w'] = off, h, w
        desc['flip'] = [int(bool(p[0])) for p in params]
        desc['oy'] = [int(p[1]) for p in params]
        desc['ox'] = [int(p[2]) for p in params]
        desc['sy'] = (h / self.resize).astype(np.float32)
        desc['sx'] = (w / self.resize).astype(np.float32)
        with torch.cuda.device(self.device):
            if late and (self._dev_blob is None or self._dev_blob.numel() < pixel_bytes):
                self._dev_blob = torch.empty(int(pixel_bytes * 1.3) + 4096, dtype=torch.uint8, device=self.device)
            st = L.stream_ptr()
            dev_infos = self._dev_infos
            dev_infos[:n * 512].copy_(slot['infos'][:n * 512], non_blocking=True)
            if used > 0:
                self._dev_coef[:used].copy_(slot['coef'][:used], non_blocking=True)      # the batch's ONE (packed) coefficient copy
            if pixel_planes > self._dev_planes.numel():
                self._dev_planes = torch.empty(int(pixel_planes * 1.2), dtype=torch.uint8, device=self.device)
            for o, im in late:
                self._dev_blob[o:o + im.size].copy_(torch.from_numpy(np.array(im, copy=True).reshape(-1)))
            dev_desc = self._dev_desc
            dev_desc[:n * 40].copy_(slot['desc'][:n * 40], non_blocking=True)
            out = torch.empty((n, self.h, self.w, 3), dtype=torch.float32, device=self.device)
            # packed blocks -> inverse DCT, then resize / flip / crop / scale with the taps converted from the component planes
            L.check(self.lib.comic_jpeg_preprocess_packed(self._dev_coef.data_ptr(), dev_infos.data_ptr(), n,
                                                          int(infos['coef_count'][ok].max()) // 64 if ok.any() else 0,
                                                          self._dev_planes.data_ptr(), self._dev_blob.data_ptr() if late else None,
                                                          dev_desc.data_ptr(), out.data_ptr(), self.h, self.w, self.resize, st),
                    'jpeg_preprocess_packed')
            ev = torch.cuda.Event()
            ev.record(torch.cuda.current_stream())
        self._pending.append((('coef', slot), ev))
        return out

    def _fill_desc(self, geo, params):
        import ctypes as C
        L = self.L
        n = len(geo)
        dbuf = np.zeros(n * C.sizeof(L.ImageDesc), np.uint8)
        desc = (L.ImageDesc * n).from_buffer(dbuf)
        for i, ((off, ih, iw), (flip, oy, ox)) in enumerate(zip(geo, params)):
            d = desc[i]
            d.offset, d.in_h, d.in_w, d.flip, d.oy, d.ox = off, ih, iw, int(bool(flip)), int(oy), int(ox)
            d.sy, d.sx = np.float32(ih / self.resize), np.float32(iw / self.resize)
        del desc
        return dbuf

    # ---- consumer side ------------------------------------------------------------------------------------------
    def _reap(self):
        while self._pending and self._pending[0][1].query():
            slot = self._pending.pop(0)[0]
            if isinstance(slot, tuple) and slot[0] == 'coef':    # coefficient staging slot of the split JPEG decoder
                self._free_coef.put(slot[1])
            elif isinstance(slot, tuple):        # shared-memory block of a DecodePool
                slot[1].release(slot[2])
            else:
                self._free.put(slot)

    def _shm_tensor(self, blk):
        """uint8 host tensor over a DecodePool block, registered once as pinned memory (so the H2D copy is asynchronous
        and DMA-able in place)."""
        torch = self.torch
        cache = self.__dict__.setdefault('_shm_tensors', {})
        t = cache.get(blk.name)
        if t is None:
            t = torch.frombuffer(blk.buf, dtype=torch.uint8)
            rc = torch.cuda.cudart().cudaHostRegister(t.data_ptr(), t.numel(), 0)
            if int(rc) != 0:
                raise RuntimeError('cudaHostRegister of a decode staging block failed (%s)' % rc)
            cache[blk.name] = t
        return t

    def unregister_shm(self):
        """Before a DecodePool is closed: its blocks stop being registered pinned memory (a mapping that goes away while
        the runtime still lists it as pinned made later host-to-device copies of OTHER buffers fail)."""
        torch = self.torch
        cache = self.__dict__.pop('_shm_tensors', {})
        if cache:
            torch.cuda.synchronize()
        for t in cache.values():
            torch.cuda.cudart().cudaHostUnregister(t.data_ptr())

    def finish(self, packed):
        """Consumer half.  The copies and launches go to a stream of the loader's own: the host-to-device DMA of batch k+1
        (60 MB for 64 images of 640 x 480) runs beside the kernels of step k that the consumer's stream still holds, instead
        of queueing behind them and in front of step k+1; the consumer's stream waits for the loader's work only (the staging
        and device buffers of the loader are touched by its own stream alone, the batch tensor is handed over with
        record_stream)."""
        torch = self.torch
        if os.environ.get('COMIC_LOADER_STREAM', '1') != '1':
            out = self._finish(packed)
        else:
            with torch.cuda.device(self.device):
                main = torch.cuda.current_stream()
                side = self.__dict__.get('_side')
                if side is None:
                    side = self._side = torch.cuda.Stream()
                with torch.cuda.stream(side):
                    out = self._finish(packed)
                main.wait_stream(side)
                out.record_stream(main)
        # Back-pressure: at most two staging slots / blocks wait for the device.  Staging returns to the producer's free list
        # only when the consumer polls its events (here), so a consumer that runs ahead of the device and then blocks on an
        # empty prefetch queue would starve the producer of staging for good; with the cap, queue depth + the producer's
        # one + two pending never exceed the depth + 4 slots there are.
        while len(self._pending) > 2:
            self._pending[0][1].synchronize()
            self._reap()
        return out

    def _finish(self, packed):
        import ctypes as C
        torch, L = self.torch, self.L
        self._reap()
        n, total = packed.n, packed.total
        dbytes = n * C.sizeof(L.ImageDesc)
        slot = packed.slot
        if isinstance(slot, tuple) and slot[0] == 'split':
            return self._finish_split(packed)
        if isinstance(slot, tuple):
            # waits: the workers have written every image of this batch (on a timeout the pool retires the block itself;
            # any other failure -- an oversized image -- leaves no writer behind: the block goes back)
            try:
                geo = slot[1].geometry(slot[4], slot[3], slot[2])
            except ValueError:
                slot[1].release(slot[2])
                raise
            packed.desc = self._fill_desc(geo, slot[5])
            host = self._shm_tensor(slot[2])
            with torch.cuda.device(self.device):
                if self._dev_blob is None or self._dev_blob.numel() < total:
                    self._dev_blob = torch.empty(int(total * 1.3) + 4096, dtype=torch.uint8, device=self.device)
                dev_desc = torch.from_numpy(packed.desc[:dbytes]).to(self.device)
                self._dev_blob[:total].copy_(host[:total], non_blocking=True)
                out = torch.empty((n, self.h, self.w, 3), dtype=torch.float32, device=self.device)
                L.check(self.lib.comic_image_preprocess(self._dev_blob.data_ptr(), dev_desc.data_ptr(), n, out.data_ptr(),
                                                        self.h, self.w, self.resize, L.stream_ptr()), 'image_preprocess')
                ev = torch.cuda.Event()
                ev.record(torch.cuda.current_stream())
            self._pending.append((slot, ev))
            return out
        if slot is None:
            # no pinned slot was available to the producer: stage through a new (or a larger) one, made here
            slot = dict(blob=torch.empty(int(total * 1.3) + 4096, dtype=torch.uint8).pin_memory(),
                        desc=torch.empty(max(dbytes, 64 * C.sizeof(L.ImageDesc)), dtype=torch.uint8).pin_memory())
            slot['blob'].numpy()[:total] = packed.blob[:total]
            slot['desc'].numpy()[:dbytes] = packed.desc[:dbytes]
            recycle = self._n_slots < self._max_slots          # joins the free list once its copy has executed
            self._n_slots += int(recycle)
        else:
            recycle = True
        with torch.cuda.device(self.device):
            if self._dev_blob is None or self._dev_blob.numel() < total:
                self._dev_blob = torch.empty(int(total * 1.3) + 4096, dtype=torch.uint8, device=self.device)
            dev_desc = torch.empty(dbytes, dtype=torch.uint8, device=self.device)
            self._dev_blob[:total].copy_(slot['blob'][:total], non_blocking=True)
            dev_desc.copy_(slot['desc'][:dbytes], non_blocking=True)
            out = torch.empty((n, self.h, self.w, 3), dtype=torch.float32, device=self.device)
            L.check(self.lib.comic_image_preprocess(self._dev_blob.data_ptr(), dev_desc.data_ptr(), n, out.data_ptr(),
                                                    self.h, self.w, self.resize, L.stream_ptr()), 'image_preprocess')
            ev = torch.cuda.Event()
            ev.record(torch.cuda.current_stream())
        if recycle:
            self._pending.append((slot, ev))
        return out

    def __call__(self, images_u8, params):
        """Both halves in the calling thread (tests, single-threaded use)."""
        return self.finish(self.pack(images_u8, params))


def preprocess_image(path_or_array, height, width, augment, rng, params=None):
    img = decode_image(path_or_array)
    img = img.astype(np.float32) * np.float32(1.0 / 255)      # tf.image.convert_image_dtype [TF-1.9]
    img = resize_bilinear_tf1(img, 256, 256)
    flip, oy, ox = params if params is not None else draw_augmentation(augment, height, width, rng)
    if flip:
        img = img[:, ::-1]
    img = img[oy:oy + height, ox:ox + width]
    return np.ascontiguousarray((img - np.float32(0.5)) * np.float32(2.0))


class InputManager(object):
    """ Input Manager object."""
    _default_image_size = {'inception_v3': 299, 'inception_v1': 224}

    def __init__(self, config, is_inference=False):
        c = config
        s = c.cnn_input_size
        if not (isinstance(s, list) and len(s) == 2 and 0 not in s):
            c.cnn_input_size = [self._default_image_size[c.cnn_name]] * 2
        self._setup(c, is_inference)

    def close(self):
        """End of a train / eval / inference stage: stop the prefetch threads and the decode pool."""
        if getattr(self, '_jpeg_pool', None) is not None:
            self._jpeg_pool.closing = True            # (a producer waiting for a staging slot would sit out the join below)
        for name in ('batch_train', 'batch_eval', 'batch_infer'):
            it = getattr(self, name, None)
            if isinstance(it, Prefetch):
                it.close()
        self._pool.shutdown(wait=False)
        if getattr(self, '_decode_pool', None) is not None:
            if getattr(self, '_devpre', None) is not None:
                self._devpre.unregister_shm()
            self._decode_pool.close()
            self._decode_pool = None
        if getattr(self, '_jpeg_pool', None) is not None:
            self._jpeg_pool.close()
            self._jpeg_pool = None
        pil = getattr(getattr(self, '_devpre', None), '_pil_pool', None)
        if pil is not None:
            pil.shutdown(wait=False)

    def _setup(self, config, is_inference):
        config.split_sizes = {}
        self.config = c = config
        self.is_inference = is_inference
        self._rng = random.Random(c.rand_seed)            # random.seed(c.rand_seed), :58
        # data parallel (config.dp_world ranks): every rank shuffles with the SAME stream and takes every dp_world-th
        # item, so the shards are disjoint and their union is the single-process order; the augmentation draws come
        # from a per-rank stream (they must not advance the shuffle stream by rank-dependent amounts)
        self._world = max(1, int(getattr(c, 'dp_world', 1) or 1))
        self._rank = int(getattr(c, 'dp_rank', 0) or 0)
        self._aug_rng = random.Random(c.rand_seed + 7919 * (self._rank + 1)) if self._world > 1 else self._rng
        # decode / resize workers (the reference maps with num_parallel_calls=3, :169; one MI355X consumes three
        # orders of magnitude more images per second than a TF-1 CPU pipeline) and batches kept ahead of the step
        self._pool = ThreadPoolExecutor(max_workers=int(getattr(c, 'loader_threads', 0)) or min(16, os.cpu_count() or 3))
        self._prefetch_depth = int(getattr(c, 'loader_prefetch', 4))
        self._get_vocab()
        if is_inference:
            if 'coco' in c.infer_set:
                coco_set = 'test2014' if c.infer_set == 'coco_test' else 'val2014'
                if c.infer_set != 'coco_test':
                    c.batch_size_infer = 61
                self.filenames_infer = [pjoin(c.dataset_dir, coco_set, f)
                                        for f in os.listdir(pjoin(c.dataset_dir, coco_set))]
            else:
                fname = 'filenames_test.txt' if c.infer_set == 'test' else 'filenames_valid.txt'
                with open(pjoin(c.dataset_dir, 'captions', fname)) as f:
                    self.filenames_infer = [l.strip() for l in f.readlines()]
        if 'coco' in c.dataset_file_pattern:
            self.buckets = [11, 13, 15]
        elif 'insta' in c.dataset_file_pattern:
            self.buckets = [7, 10, 13]
        else:
            self.buckets = [11, 13, 15]
        self._post_vocab_setup()
        if is_inference:
            self.batch_infer = self._batch_setup('infer')
        else:
            self.batch_train = self._batch_setup('train')
            self.batch_eval = self._batch_setup('valid')
        print('INFO: Input pipelines setup complete.')

    def _post_vocab_setup(self):
        pass

    def _get_vocab(self):
        c = self.config
        if '{}' not in c.dataset_file_pattern:
            raise ValueError('`dataset_file_pattern` must have `{}`.')
        with open(pjoin(c.dataset_dir, 'captions', c.dataset_file_pattern.format('itow')) + '.json') as f:
            c.itow = json.load(f)
        with open(pjoin(c.dataset_dir, 'captions', c.dataset_file_pattern.format('wtoi')) + '.json') as f:
            c.wtoi = json.load(f)
        c.vocab_size = len(c.itow)

    # ---- caption tokenisation (per manager) ---------------------------------------------
    def _encode(self, words):
        c = self.config
        return np.array([c.wtoi.get(w, c.wtoi['<UNK>']) for w in words], np.int32)

    def _read_split(self, split):
        c = self.config
        fp = pjoin(c.dataset_dir, 'captions', c.dataset_file_pattern.format(split)) + '.txt'
        if not os.path.isfile(fp):
            return None
        with open(fp, 'r') as f:
            data = [l.strip().split(',') for l in f.readlines() if l.strip()]
        return [[l[0], l[1].split(' ')] for l in data]

    def _batch_setup(self, split):
        c = self.config
        is_training = 'train' in split and not self.is_inference
        if self.is_inference:
            batch_size = c.batch_size_infer
            data = [[f, ['null']] for f in self.filenames_infer]
            assert len(data) % batch_size == 0
            c.split_sizes['infer'] = len(data)
        else:
            data = self._read_split(split)
            if data is None:
                return None
            c.split_sizes[split] = len(data)
            if is_training:
                gs = getattr(c, 'accum_grads_step', 1)
                batch_size = c.batch_size_train
                c.max_step = int(len(data) / (batch_size * self._world) * c.max_epoch / gs)    # global batch
            else:
                batch_size = c.batch_size_eval
                assert len(data) % batch_size == 0
        augment = is_training and c.cnn_input_augment
        print('INFO: Augment {} images: {}'.format(split, augment))
        return Prefetch(self._batches(data, batch_size, is_training, augment), self._prefetch_depth, self._finish_batch)

    def _shard(self, data):
        """This rank's share of the (commonly shuffled) list: items rank, rank + W, ...; equal counts on every rank."""
        if self._world == 1:
            return data
        n = len(data) // self._world * self._world
        return data[self._rank:n:self._world]

    def _gen(self, data, is_training):
        c = self.config
        if is_training:
            self._rng.shuffle(data)
        while True:
            for d in (self._shard(data) if is_training else data):
                yield pjoin(c.dataset_dir, d[0]) if not os.path.isabs(d[0]) else d[0], self._encode(d[1])
            if is_training:
                self._rng.shuffle(data)

    def _load(self, path, augment, params=None):
        h, w = self.config.cnn_input_size
        return preprocess_image(path, h, w, augment, self._aug_rng, params)

    def enable_device_preprocess(self, device='cuda:0'):
        """From the next batch on: the host only decodes (thread pool), resize / flip / crop / scale run on `device`
        and the batches carry device tensors (bit-identical values; `CaptionModel` takes either)."""
        if not str(device).startswith('cuda') or getattr(self, '_devpre', None) is not None:
            return                                   # (a second call: the loader of this manager is set up)
        h, w = self.config.cnn_input_size
        self._devpre = DevicePreprocessor(device, h, w)
        c = self.config
        # the split JPEG decoder is the default loader; --loader_processes N (an explicit request) or --no-loader_split_jpeg
        # select the PIL loaders
        if (getattr(c, 'loader_split_jpeg', True) and not int(getattr(c, 'loader_processes', 0) or 0)
                and getattr(self, '_jpeg_pool', None) is None):
            self._jpeg_pool = JpegSplitPool(int(getattr(c, 'loader_threads', 0)) or min(16, os.cpu_count() or 3),
                                            slot_elems=int(getattr(c, 'loader_slot_bytes', 640 * 640 * 3)) // 2,
                                            max_batch=max(c.batch_size_train, getattr(c, 'batch_size_eval', 1),
                                                          getattr(c, 'batch_size_infer', 1)),
                                            timeout_s=float(getattr(c, 'loader_timeout_s', 120.0)),
                                            cache_gb=float(getattr(c, 'loader_cache_gb', 0.0) or 0.0))
            self._devpre.enable_split(self._jpeg_pool, self._prefetch_depth + 4)
            return
        nproc = int(getattr(self.config, 'loader_processes', 0) or 0)
        if nproc > 0 and getattr(self, '_decode_pool', None) is None:
            c = self.config
            self._decode_pool = DecodePool(nproc, blocks=self._prefetch_depth + 4,
                                           slot_bytes=int(getattr(c, 'loader_slot_bytes', 640 * 640 * 3)),
                                           max_batch=max(c.batch_size_train, getattr(c, 'batch_size_eval', 1),
                                                         getattr(c, 'batch_size_infer', 1)),
                                           timeout_s=float(getattr(c, 'loader_timeout_s', 120.0)))

    def _finish_batch(self, item):
        """Consumer-thread half of a batch: packed images -> device tensor (see DevicePreprocessor)."""
        ims, rest = item[0], item[1:]
        if isinstance(ims, PackedImages):
            ims = self._devpre.finish(ims)
        return (ims,) + tuple(rest)

    def _load_many(self, paths, augment):
        h, w = self.config.cnn_input_size
        params = [draw_augmentation(augment, h, w, self._aug_rng) for _ in paths]
        devpre = getattr(self, '_devpre', None)
        dpool = getattr(self, '_decode_pool', None)
        if devpre is not None and getattr(self, '_jpeg_pool', None) is not None and all(isinstance(p, str) for p in paths):
            return devpre.pack_paths_split(paths, params)       # entropy decode on C threads, pixels on the device
        if devpre is not None and dpool is not None and all(isinstance(p, str) for p in paths):
            return devpre.pack_paths(dpool, paths, params)      # decode in worker processes, straight into staging
        if devpre is not None:       # CPU half here (producer thread); the device half runs in Prefetch's consumer hook
            return devpre.pack(list(self._pool.map(decode_image, paths)), params)
        return np.stack(list(self._pool.map(lambda a: self._load(a[0], augment, a[1]), zip(paths, params))))

    def _batches(self, data, batch_size, is_training, augment):
        """bucket_by_sequence_length(boundaries=self.buckets, pad -> wtoi['<PAD>']) (:177-183)."""
        c = self.config
        pad = c.wtoi['<PAD>']
        buckets = [[] for _ in range(len(self.buckets) + 1)]
        for path, cap in self._gen(data, is_training):
            k = sum(1 for b in self.buckets if len(cap) >= b)
            buckets[k].append((path, cap))
            if len(buckets[k]) == batch_size:
                items, buckets[k] = buckets[k], []
                ims = self._load_many([it[0] for it in items], augment)
                L = max(len(it[1]) for it in items)
                caps = np.full((batch_size, L), pad, np.int32)
                for i, it in enumerate(items):
                    caps[i, :len(it[1])] = it[1]
                yield ims, caps


class InputManager_Radix(InputManager):
    """ Input Manager object for Radix-token models."""

    def _post_vocab_setup(self):
        c = self.config
        max_word_len = len(ops.number_to_base(len(c.wtoi), c.radix_base))
        self.buckets = [b * max_word_len for b in self.buckets]
        self.radix_wtoi = ops.build_radix_wtoi(c.wtoi, c.radix_base)

    def _encode(self, words):
        t = self.radix_wtoi
        return np.concatenate([t.get(w, t['<UNK>']) for w in words]).astype(np.int32)


class InputManager_Char(InputManager):
    """ Input Manager object for character-token models."""

    def _post_vocab_setup(self):
        c = self.config
        self.buckets = [45, 55, 70] if 'coco' in c.dataset_file_pattern else [29, 42, 61]

    def _get_vocab(self):
        c = self.config
        with open(pjoin(c.dataset_dir, 'captions', c.dataset_file_pattern.format('wtoi')) + '.json') as f:
            wtoi = json.load(f)
        idx = wtoi['<PAD>']
        ctoi, itoc = {}, {}
        for ch in ['<PAD>', ' '] + list(string.digits + string.ascii_lowercase):
            ctoi[ch] = idx; itoc[str(idx)] = ch; idx += 1
        ctoi['<GO>'] = len(ctoi); ctoi['<EOS>'] = len(ctoi)
        itoc[str(len(itoc))] = '<GO>'; itoc[str(len(itoc))] = '<EOS>'
        c.itow, c.wtoi, c.vocab_size = itoc, ctoi, len(itoc)

    def _encode(self, words):
        c = self.config
        cap = [c.wtoi[ch] for ch in ' '.join(words[1:-1])]
        return np.array([c.wtoi['<GO>']] + cap + [c.wtoi['<EOS>']], np.int32)


class InputManager_SCST(InputManager_Radix):
    """ batch_train yields (images, refs) with refs = up to 5 reference strings per image."""

    def _batch_setup(self, split):
        c = self.config
        is_training = 'train' in split and not self.is_inference
        if self.is_inference:
            return InputManager._batch_setup(self, split)
        data = self._read_split(split)
        if data is None or not is_training:
            return None
        groups = {}
        for path, words in data:
            s = ' '.join(words).replace('<GO> ', '').replace(' <EOS>', '')
            groups.setdefault(path, []).append(s)
        data = list(groups.items())
        c.split_sizes[split] = len(data)
        batch_size = c.batch_size_train
        c.max_step = int(len(data) / (batch_size * self._world) * c.max_epoch / getattr(c, 'accum_grads_step', 1))
        augment = is_training and c.cnn_input_augment
        return Prefetch(self._scst_batches(data, batch_size, augment), self._prefetch_depth, self._finish_batch)

    def _scst_batches(self, data, batch_size, augment):
        c = self.config
        self._rng.shuffle(data)
        while True:
            mine = self._shard(data)
            for i in range(0, len(mine) - batch_size + 1, batch_size):      # batch_and_drop_remainder
                items = mine[i:i + batch_size]
                ims = self._load_many([pjoin(c.dataset_dir, it[0]) if not os.path.isabs(it[0]) else it[0]
                                        for it in items], augment)
                yield ims, [list(it[1][:5]) for it in items]
            self._rng.shuffle(data)

    def captions_to_batched_ids(self, hypos):
        return ops.captions_to_batched_ids(hypos, self.config, getattr(self, 'radix_wtoi', None))
