"""Generates tests/golden/jpeg_split_golden.npz: small baseline JPEG files (bytes) and the RGB pixels PIL's libjpeg-turbo
decodes them to -- the known answers that pin oracle/jpeg_ref.py + libcomic_jpeg.so (and, on the GPU, comic_jpeg_pixels)
independently of the Pillow build present when the tests run.  Run from the repo root: python tests/golden/make_jpeg_golden.py
(Pillow 12.2.0 / libjpeg-turbo as bundled by the wheel wrote the committed file)."""
import io
import os
import numpy as np
from PIL import Image

rng = np.random.default_rng(20261004)
yy, xx = np.mgrid[0:61, 0:83]
smooth = np.stack([127 + 120 * np.sin(xx / 9.0) * np.cos(yy / 7.0), 127 + 100 * np.cos((xx + yy) / 11.0),
                   (xx * 3 + yy * 2) % 256], -1)
img = np.clip(smooth + rng.normal(0, 12, smooth.shape), 0, 255).astype(np.uint8)
cases = {
    'h2v2_q90': dict(quality=90, subsampling=2),
    'h2v1_q75': dict(quality=75, subsampling=1),
    'h1v1_q95': dict(quality=95, subsampling=0),
    'h2v2_q60_restart': dict(quality=60, subsampling=2, restart_marker_blocks=3),
    'h2v2_q85_optimised': dict(quality=85, subsampling=2, optimize=True),
}
out = {}
for name, kw in cases.items():
    b = io.BytesIO()
    Image.fromarray(img).save(b, 'JPEG', **kw)
    data = b.getvalue()
    out[name + '.jpg'] = np.frombuffer(data, np.uint8)
    out[name + '.rgb'] = np.asarray(Image.open(io.BytesIO(data)).convert('RGB'))
b = io.BytesIO()
Image.fromarray(img[:37, :50, 1]).save(b, 'JPEG', quality=80)
out['grey_q80.jpg'] = np.frombuffer(b.getvalue(), np.uint8)
out['grey_q80.rgb'] = np.asarray(Image.open(io.BytesIO(b.getvalue())).convert('RGB'))
np.savez_compressed(os.path.join(os.path.dirname(os.path.abspath(__file__)), 'jpeg_split_golden.npz'), **out)
print({k: v.shape for k, v in out.items()})
