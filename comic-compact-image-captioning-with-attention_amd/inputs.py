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
from concurrent.futures import ThreadPoolExecutor

import numpy as np

from . import ops

pjoin = os.path.join


def resize_bilinear_tf1(img, out_h, out_w):
    """tf.image.resize_images(..., BILINEAR, align_corners=False) [TF-1.9]: src = dst * in/out."""
    in_h, in_w = img.shape[:2]
    ys = np.arange(out_h, dtype=np.float32) * np.float32(in_h / out_h)
    xs = np.arange(out_w, dtype=np.float32) * np.float32(in_w / out_w)
    y0 = np.floor(ys).astype(np.int64); x0 = np.floor(xs).astype(np.int64)
    y1 = np.minimum(y0 + 1, in_h - 1); x1 = np.minimum(x0 + 1, in_w - 1)
    wy = (ys - y0)[:, None, None]; wx = (xs - x0)[None, :, None]
    top = img[y0][:, x0] * (1 - wx) + img[y0][:, x1] * wx
    bot = img[y1][:, x0] * (1 - wx) + img[y1][:, x1] * wx
    return (top * (1 - wy) + bot * wy).astype(np.float32)


def preprocess_image(path_or_array, height, width, augment, rng):
    if isinstance(path_or_array, np.ndarray):
        img = path_or_array
    else:
        from PIL import Image
        with Image.open(path_or_array) as im:
            img = np.asarray(im.convert('RGB'))
    img = img.astype(np.float32) / np.float32(255.0)
    img = resize_bilinear_tf1(img, 256, 256)
    if augment:
        if rng.random() < 0.5:
            img = img[:, ::-1]
        oy = rng.randrange(0, 256 - height + 1); ox = rng.randrange(0, 256 - width + 1)
    else:
        oy = (256 - height) // 2; ox = (256 - width) // 2
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

    def _setup(self, config, is_inference):
        config.split_sizes = {}
        self.config = c = config
        self.is_inference = is_inference
        self._rng = random.Random(c.rand_seed)            # random.seed(c.rand_seed), :58
        self._pool = ThreadPoolExecutor(max_workers=3)     # num_parallel_calls=3, :169
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
                c.max_step = int(len(data) / batch_size * c.max_epoch / gs)
            else:
                batch_size = c.batch_size_eval
                assert len(data) % batch_size == 0
        augment = is_training and c.cnn_input_augment
        print('INFO: Augment {} images: {}'.format(split, augment))
        return self._batches(data, batch_size, is_training, augment)

    def _gen(self, data, is_training):
        c = self.config
        if is_training:
            self._rng.shuffle(data)
        while True:
            for d in data:
                yield pjoin(c.dataset_dir, d[0]) if not os.path.isabs(d[0]) else d[0], self._encode(d[1])
            if is_training:
                self._rng.shuffle(data)

    def _load(self, path, augment):
        h, w = self.config.cnn_input_size
        return preprocess_image(path, h, w, augment, self._rng)

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
                ims = list(self._pool.map(lambda it: self._load(it[0], augment), items))
                L = max(len(it[1]) for it in items)
                caps = np.full((batch_size, L), pad, np.int32)
                for i, it in enumerate(items):
                    caps[i, :len(it[1])] = it[1]
                yield np.stack(ims), caps


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
        c.max_step = int(len(data) / batch_size * c.max_epoch / getattr(c, 'accum_grads_step', 1))
        augment = is_training and c.cnn_input_augment
        return self._scst_batches(data, batch_size, augment)

    def _scst_batches(self, data, batch_size, augment):
        c = self.config
        self._rng.shuffle(data)
        while True:
            for i in range(0, len(data) - batch_size + 1, batch_size):      # batch_and_drop_remainder
                items = data[i:i + batch_size]
                ims = list(self._pool.map(
                    lambda it: self._load(pjoin(c.dataset_dir, it[0]) if not os.path.isabs(it[0]) else it[0], augment),
                    items))
                yield np.stack(ims), [list(it[1][:5]) for it in items]
            self._rng.shuffle(data)

    def captions_to_batched_ids(self, hypos):
        return ops.captions_to_batched_ids(hypos, self.config, getattr(self, 'radix_wtoi', None))
