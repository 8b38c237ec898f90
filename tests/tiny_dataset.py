"""Builds a tiny dataset in the reference's on-disk formats (coco_prepro.py:123-162,
prepro_base.py:186-253, prepro_ngrams.py:149-151) for CLI / pipeline tests."""
import json
import os
import pickle

import numpy as np


def make(root, n_train=16, n_valid=4, n_test=4, seed=0, pattern='mscoco_{}_w5_s20_include_restval'):
    from PIL import Image
    from oracle import scorer_ref
    rng = np.random.default_rng(seed)
    words = ['a', 'man', 'dog', 'cat', 'on', 'the', 'table', 'sitting', 'red', 'bench', 'with', 'frisbee', 'park',
             'two', 'people', 'standing', 'near', 'train', 'street', 'sign']
    os.makedirs(os.path.join(root, 'captions'), exist_ok=True)
    os.makedirs(os.path.join(root, 'images'), exist_ok=True)
    wtoi = {'<PAD>': -1}
    for i, w in enumerate(words):
        wtoi[w] = i
    for tok in ('<UNK>', '<GO>', '<EOS>'):
        wtoi[tok] = len(wtoi) - 1
    itow = {str(v): k for k, v in wtoi.items()}
    for name, obj in (('wtoi', wtoi), ('itow', itow)):
        with open(os.path.join(root, 'captions', pattern.format(name) + '.json'), 'w') as f:
            json.dump(obj, f)

    def caption():
        n = int(rng.integers(4, 9))
        return ' '.join(rng.choice(words, n))
    refs_all = []
    annotations = []          # COCO-style annotation file of the test split (for the native metric scores)
    for split, n_img in (('train', n_train), ('valid', n_valid), ('test', n_test)):
        lines, names = [], []
        for i in range(n_img):
            rel = os.path.join('images', 'COCO_%s2014_%012d.jpg' % (split, i + 1))
            arr = rng.integers(0, 256, (int(rng.integers(180, 300)), int(rng.integers(180, 300)), 3), dtype=np.uint8)
            Image.fromarray(arr).save(os.path.join(root, rel), quality=90)
            names.append(os.path.join(root, rel))
            caps = [caption() for _ in range(5 if split == 'train' else 1)]
            if split == 'train':
                refs_all.append(caps)
            for cpt in caps:
                lines.append('%s,<GO> %s <EOS>' % (rel, cpt))
            if split == 'test':
                annotations += [dict(image_id=i + 1, caption=c_) for c_ in caps + [caption()]]
        with open(os.path.join(root, 'captions', pattern.format(split) + '.txt'), 'w', newline='') as f:
            f.write('\r\n'.join(lines))
        if split != 'train':
            with open(os.path.join(root, 'captions', 'filenames_%s.txt' % split), 'w') as f:
                f.write('\n'.join(names))
    with open(os.path.join(root, 'captions', 'captions_test_annotations.json'), 'w') as f:
        json.dump(dict(annotations=annotations), f)
    df = scorer_ref.build_df_from_refs(refs_all)
    with open(os.path.join(root, 'captions', pattern.format('scst-words') + '.p'), 'wb') as f:
        pickle.dump({'document_frequency': dict(df['document_frequency']), 'ref_len': df['ref_len']}, f, 2)
    return root
