"""TEST INFRASTRUCTURE (oracle): CPU restatement of the PIXEL half of a baseline JPEG decode -- quantised DCT coefficients
-> uint8 RGB -- with libjpeg's integer arithmetic, the checker of csrc/jpeg_pixels.hip (comic_jpeg_pixels).

What it restates (libjpeg / libjpeg-turbo 3.x as vendored by Pillow, the decoder behind PIL.Image.open and behind
tf.image.decode_jpeg in the reference's tf.data map, common/inputs/manager_image_caption.py:163-175; the library is a
third-party dependency absent from /root/reference, so the published algorithms are restated):
  * jidctint.c jpeg_idct_islow   -- "ISLOW" 8x8 inverse DCT, CONST_BITS 13 / PASS1_BITS 2, the JDCT default
  * jdsample.c h2v1_fancy_upsample / h2v2_fancy_upsample -- triangle-filter chroma upsampling (do_fancy_upsampling default)
  * jdcolor.c  ycc_rgb_convert   -- 16-bit fixed-point YCbCr -> RGB
Pinned: tests/test_jpeg_split.py compares idct + upsample + colour of the coefficients that libcomic_jpeg.so extracts with
PIL's own decode of the same file, bit for bit, over sizes, qualities, samplings, restart intervals and real photographs.
Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may import this module.
"""
import numpy as np

CONST_BITS, PASS1_BITS = 13, 2
FIX_0_298631336, FIX_0_390180644, FIX_0_541196100, FIX_0_765366865 = 2446, 3196, 4433, 6270
FIX_0_899976223, FIX_1_175875602, FIX_1_501321110, FIX_1_847759065 = 7373, 9633, 12299, 15137
FIX_1_961570560, FIX_2_053119869, FIX_2_562915447, FIX_3_072711026 = 16069, 16819, 20995, 25172


def _descale(x, n):
    return (x + (1 << (n - 1))) >> n


def _idct_1d(v, shift):
    """One pass of jpeg_idct_islow over axis 0 of v[8, ...] (int64), outputs descaled by `shift` bits."""
    z2, z3 = v[2], v[6]
    z1 = (z2 + z3) * FIX_0_541196100
    tmp2 = z1 - z3 * FIX_1_847759065
    tmp3 = z1 + z2 * FIX_0_765366865
    tmp0 = (v[0] + v[4]) << CONST_BITS
    tmp1 = (v[0] - v[4]) << CONST_BITS
    tmp10, tmp13, tmp11, tmp12 = tmp0 + tmp3, tmp0 - tmp3, tmp1 + tmp2, tmp1 - tmp2
    t0, t1, t2, t3 = v[7], v[5], v[3], v[1]
    z1, z2, z3, z4 = t0 + t3, t1 + t2, t0 + t2, t1 + t3
    z5 = (z3 + z4) * FIX_1_175875602
    t0 = t0 * FIX_0_298631336
    t1 = t1 * FIX_2_053119869
    t2 = t2 * FIX_3_072711026
    t3 = t3 * FIX_1_501321110
    z1 = -z1 * FIX_0_899976223
    z2 = -z2 * FIX_2_562915447
    z3 = -z3 * FIX_1_961570560 + z5
    z4 = -z4 * FIX_0_390180644 + z5
    t0 = t0 + z1 + z3
    t1 = t1 + z2 + z4
    t2 = t2 + z2 + z3
    t3 = t3 + z1 + z4
    return np.stack([_descale(tmp10 + t3, shift), _descale(tmp11 + t2, shift), _descale(tmp12 + t1, shift),
                     _descale(tmp13 + t0, shift), _descale(tmp13 - t0, shift), _descale(tmp12 - t1, shift),
                     _descale(tmp11 - t2, shift), _descale(tmp10 - t3, shift)])


def idct_plane(coef, quant, blocks_h, blocks_w):
    """coef [blocks_h * blocks_w, 64] int16 (natural order), quant [64] -> uint8 plane [blocks_h * 8, blocks_w * 8].
    Pass 1 over columns (descale CONST_BITS - PASS1_BITS), pass 2 over rows (CONST_BITS + PASS1_BITS + 3), + 128, clamped to
    0..255 (the range-limit table; the vector code of libjpeg-turbo saturates the same way)."""
    blk = coef.reshape(-1, 8, 8).astype(np.int64) * quant.reshape(1, 8, 8).astype(np.int64)
    ws = _idct_1d(blk.transpose(1, 0, 2), CONST_BITS - PASS1_BITS)              # [row u -> y, n, col]
    out = _idct_1d(ws.transpose(2, 1, 0), CONST_BITS + PASS1_BITS + 3)         # [col -> x, n, y]
    px = np.clip(out.transpose(1, 2, 0) + 128, 0, 255).astype(np.uint8)        # [n, y, x]
    return px.reshape(blocks_h, blocks_w, 8, 8).transpose(0, 2, 1, 3).reshape(blocks_h * 8, blocks_w * 8)


def upsample_h2v1(c, out_w):
    """h2v1_fancy_upsample over the real columns c[h, n]: (3 near + far + {1, 2}) >> 2, the ends copied."""
    c = c.astype(np.int32)
    n = c.shape[1]
    out = np.empty((c.shape[0], 2 * n), np.int32)
    left = np.concatenate([c[:, :1], c[:, :-1]], 1)
    right = np.concatenate([c[:, 1:], c[:, -1:]], 1)
    out[:, 0::2] = (3 * c + left + 1) >> 2
    out[:, 1::2] = (3 * c + right + 2) >> 2
    out[:, 0] = c[:, 0]
    out[:, -1] = c[:, -1]
    return out[:, :out_w].astype(np.uint8)


def upsample_h2v2(c, out_h, out_w):
    """h2v2_fancy_upsample over the real samples c[m, n]: vertically 3 near + far (the row above for even output rows, below
    for odd ones; the first / last real row stands in for the missing neighbour), horizontally (3 this + neighbour + {8, 7})
    >> 4, the first / last column (4 this + {8, 7}) >> 4."""
    c = c.astype(np.int32)
    m, n = c.shape
    up = np.concatenate([c[:1], c[:-1]], 0)
    down = np.concatenate([c[1:], c[-1:]], 0)
    rows = np.empty((2 * m, n), np.int32)
    rows[0::2] = 3 * c + up
    rows[1::2] = 3 * c + down
    left = np.concatenate([rows[:, :1], rows[:, :-1]], 1)
    right = np.concatenate([rows[:, 1:], rows[:, -1:]], 1)
    out = np.empty((2 * m, 2 * n), np.int32)
    out[:, 0::2] = (3 * rows + left + 8) >> 4
    out[:, 1::2] = (3 * rows + right + 7) >> 4
    out[:, 0] = (4 * rows[:, 0] + 8) >> 4
    out[:, -1] = (4 * rows[:, -1] + 7) >> 4
    return out[:out_h, :out_w].astype(np.uint8)


def ycc_to_rgb(y, cb, cr):
    """jdcolor.c: R = y + Cr_r[cr], G = y + ((Cb_g[cb] + Cr_g[cr]) >> 16), B = y + Cb_b[cb], clamped."""
    x = np.arange(256, dtype=np.int64) - 128
    half = 1 << 15

    def fix(v):
        return int(v * 65536 + 0.5)

    cr_r = (fix(1.40200) * x + half) >> 16
    cb_b = (fix(1.77200) * x + half) >> 16
    cr_g = -fix(0.71414) * x
    cb_g = -fix(0.34414) * x + half
    y = y.astype(np.int64)
    r = y + cr_r[cr]
    g = y + ((cb_g[cb] + cr_g[cr]) >> 16)
    b = y + cb_b[cb]
    return np.clip(np.stack([r, g, b], -1), 0, 255).astype(np.uint8)


def unpack(packed, blocks):
    """The packed form of a loader batch image (comic_jpeg_pool_submit_packed; 16-bit units: desc[blocks] as uint32 = (first entry
    << 7) | count, dc[blocks] int16, then the entries: (position << 10) | (value & 1023), or the pair (position, int16 value))
    -> dense int16 [blocks * 64] in natural order."""
    pk = np.ascontiguousarray(packed, np.uint16)
    desc = pk[:2 * blocks].view(np.uint32)
    dcs = pk[2 * blocks:3 * blocks].view(np.int16)
    ent = pk[3 * blocks:]
    out = np.zeros(blocks * 64, np.int16)
    out[0::64] = dcs
    for b in range(blocks):
        beg, n = int(desc[b] >> 7), int(desc[b] & 127)
        j = 0
        while j < n:
            e = int(ent[beg + j])
            pos = e >> 10
            if pos:
                v = e & 1023
                out[b * 64 + pos] = v - 1024 if v >= 512 else v
            else:
                out[b * 64 + (e & 63)] = int(ent[beg + j + 1:beg + j + 2].view(np.int16)[0])
                j += 1
            j += 1
    return out


def pixels(info, coef):
    """comic_jpeg_info (any object with its fields) + the image's coefficients (int16, info.coef_count) -> uint8 [H, W, 3]."""
    if isinstance(info, np.void):                        # a record of a JPEG_INFO_DTYPE array
        rec = info

        class _View(object):
            def __getattr__(self, name):
                return rec[name]
        info = _View()
    H, W = int(info.height), int(info.width)
    planes = []
    for c in range(int(info.ncomp)):
        bw, bh = int(info.blocks_w[c]), int(info.blocks_h[c])
        off = int(info.coef_off[c])
        q = np.array(list(info.quant[c]), np.int64)
        planes.append(idct_plane(np.asarray(coef[off:off + bw * bh * 64]).reshape(-1, 64), q, bh, bw))
    y = planes[0][:H, :W]
    if int(info.ncomp) == 1:
        return np.repeat(y[:, :, None], 3, 2)
    ch, cw = int(info.comp_h[1]), int(info.comp_w[1])
    cb, cr = planes[1][:ch, :cw], planes[2][:ch, :cw]
    if info.hmax == 2 and info.vmax == 2:
        cb, cr = upsample_h2v2(cb, H, W), upsample_h2v2(cr, H, W)
    elif info.hmax == 2:
        cb, cr = upsample_h2v1(cb, W), upsample_h2v1(cr, W)
    return ycc_to_rgb(y, cb, cr)
