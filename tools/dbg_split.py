import sys, os, tempfile, pathlib
sys.path.insert(0, os.getcwd())
import numpy as np, torch
from tests import test_jpeg_split as T
from comic_amd import inputs, _lib as L
tmp = pathlib.Path(tempfile.mkdtemp())
paths = T._mixed_files(tmp, 10)
pre = inputs.DevicePreprocessor('cuda:0', 224, 224)
jpool = inputs.JpegSplitPool(4, slot_elems=640 * 640 * 3 // 2, max_batch=16)
pre.enable_split(jpool, 3)
params = [(False, 16, 16)] * len(paths)
ref = pre(list(map(inputs.decode_image, paths)), params).cpu()
got = pre.finish(pre.pack_paths_split(paths, params)).cpu()
from PIL import Image
for i, p in enumerate(paths):
    im = Image.open(p)
    same = torch.equal(got[i], ref[i])
    print(i, im.size, im.mode, getattr(im, 'layer', None), 'same' if same else 'DIFF maxabs %.4f first diff at %s' % ((got[i]-ref[i]).abs().max().item(), (got[i]-ref[i]).abs().flatten().nonzero()[0].item()))
# single-image batches
for i, p in enumerate(paths):
    g1 = pre.finish(pre.pack_paths_split([p], [params[i]])).cpu()
    print('single', i, torch.equal(g1[0], ref[i]))
