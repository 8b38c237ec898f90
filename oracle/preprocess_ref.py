"""CPU restatement of the reference's image preprocessing: TEST INFRASTRUCTURE (only tests/ may import it).

Follows common/inputs/preprocessing/inception_preprocessing_radix.py:158-278 as called by
common/inputs/manager_image_caption.py:163-189:
  :269-270  tf.image.convert_image_dtype(uint8 -> float32)   [TF-1.9 image_ops_impl.py: cast * (1 / 255)]
  :271      tf.image.resize_bilinear(image, [256, 256])       [TF-1.9 kernels/resize_bilinear_op.cc, align_corners False:
                                                               scale = in / float(out); in_y = y * scale; lower = floor,
                                                               upper = min(ceil(in_y), in - 1); lerp = in_y - lower;
                                                               top = tl + (tr - tl) * xl; bottom = bl + (br - bl) * xl;
                                                               out = top + (bottom - top) * yl -- all float32]
  :191-192  random_flip_left_right, random_crop [height, width, 3]   (the draws are parameters here)
  :229      resize_image_with_crop_or_pad = central crop (eval)
  :198-199 / :233-234   (x - 0.5) * 2
The TF kernels are un-vendored (parity unpinned): their arithmetic is restated from TF r1.9.
"""
import numpy as np

RESIZE = 256


def convert_image_dtype_u8(img_u8):
    return img_u8.astype(np.float32) * np.float32(1.0 / 255)


def resize_bilinear(img, out_h=RESIZE, out_w=RESIZE):
    """float32 [H, W, C] -> [out_h, out_w, C], align_corners=False."""
    f = np.float32
    in_h, in_w = img.shape[:2]
    ys = np.arange(out_h, dtype=np.float32) * f(in_h / f(out_h))
    xs = np.arange(out_w, dtype=np.float32) * f(in_w / f(out_w))
    y0 = np.floor(ys).astype(np.int64)
    x0 = np.floor(xs).astype(np.int64)
    y1 = np.minimum(np.ceil(ys).astype(np.int64), in_h - 1)
    x1 = np.minimum(np.ceil(xs).astype(np.int64), in_w - 1)
    yl = (ys - y0.astype(np.float32))[:, None, None]
    xl = (xs - x0.astype(np.float32))[None, :, None]
    tl, tr = img[y0][:, x0], img[y0][:, x1]
    bl, br = img[y1][:, x0], img[y1][:, x1]
    top = tl + (tr - tl) * xl
    bottom = bl + (br - bl) * xl
    out = top + (bottom - top) * yl
    assert out.dtype == np.float32
    return out


def preprocess_image(img_u8, height, width, flip=False, oy=None, ox=None):
    """uint8 RGB [H, W, 3] -> float32 [height, width, 3] in [-1, 1].  oy / ox None: the central crop of the eval path."""
    img = resize_bilinear(convert_image_dtype_u8(np.asarray(img_u8)))
    if flip:
        img = img[:, ::-1]
    if oy is None:
        oy, ox = (RESIZE - height) // 2, (RESIZE - width) // 2
    img = img[oy:oy + height, ox:ox + width]
    return np.ascontiguousarray((img - np.float32(0.5)) * np.float32(2.0))
