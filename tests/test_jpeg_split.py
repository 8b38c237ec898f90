"""Split JPEG decoder of the input pipeline: libcomic_jpeg.so undoes the entropy coding on the host, the device
(comic_jpeg_pixels) does inverse DCT / upsampling / colour conversion in libjpeg's integer arithmetic.

CPU tests pin the host half + oracle/jpeg_ref.py (the restated pixel stage) against PIL's own decode and the committed golden
vectors; the `gpu` tests compare the device half and the loader path with PIL, bit for bit."""
import ctypes as C
import io
import os
import re

import numpy as np
import pytest

from comic_amd import _lib as L
from oracle import jpeg_ref

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _split(data):
    lib = L.load_jpeg()
    info = L.JpegInfo()
    buf = (C.c_uint8 * len(data)).from_buffer_copy(data)
    rc = lib.comic_jpeg_read_header(buf, len(data), C.byref(info))
    if rc:
        return rc, info, None
    coef = np.zeros(info.coef_count, np.int16)
    rc = lib.comic_jpeg_decode_coefficients(buf, len(data), C.byref(info), coef.ctypes.data)
    return rc, info, coef


def _encode(arr, **kw):
    from PIL import Image
    b = io.BytesIO()
    Image.fromarray(arr).save(b, 'JPEG', **kw)
    return b.getvalue()


def _pil(data):
    from PIL import Image
    return np.asarray(Image.open(io.BytesIO(data)).convert('RGB'))


def _photo(h, w, seed=0):
    """Photograph-like content: smooth structure + texture + a few hard edges."""
    rng = np.random.default_rng(seed)
    yy, xx = np.mgrid[0:h, 0:w]
    base = np.stack([127 + 110 * np.sin(xx / 17.0 + seed) * np.cos(yy / 13.0), 127 + 90 * np.cos((xx - yy) / 23.0),
                     127 + 100 * np.sin((xx + 2 * yy) / 31.0)], -1)
    base[h // 3:h // 2, w // 4:w // 2] = (250, 20, 30)
    return np.clip(base + rng.normal(0, 10, base.shape), 0, 255).astype(np.uint8)


def test_jpeg_library_exports_every_declared_symbol():
    header = open(os.path.join(ROOT, 'include', 'comic_jpeg.h')).read()
    declared = set(re.findall(r'\b(comic_jpeg_[a-z0-9_]+)\s*\(', header))
    lib = L.load_jpeg()
    for name in sorted(declared):
        assert hasattr(lib, name), name
    assert set(L.JPEG_EXPORTED_SYMBOLS) == declared, set(L.JPEG_EXPORTED_SYMBOLS) ^ declared
    assert C.sizeof(L.JpegInfo) == 512 == np.dtype(L.JPEG_INFO_DTYPE).itemsize


def test_golden_files_decode_to_the_committed_pixels(golden_dir):
    g = np.load(os.path.join(golden_dir, 'jpeg_split_golden.npz'))
    names = sorted(k[:-4] for k in g.files if k.endswith('.jpg'))
    assert len(names) == 6
    for name in names:
        rc, info, coef = _split(g[name + '.jpg'].tobytes())
        assert rc == 0, (name, rc)
        assert np.array_equal(jpeg_ref.pixels(info, coef), g[name + '.rgb']), name


@pytest.mark.parametrize('sub', [0, 1, 2])
def test_host_half_plus_oracle_give_pils_pixels(sub):
    """Samplings x sizes (whole / partial MCUs, one-row, one-column strips) x qualities, restart intervals, optimised tables."""
    n = 0
    for (w, h) in ((160, 120), (159, 107), (83, 125), (17, 9), (8, 8), (9, 33), (100, 1), (5, 200), (161, 3)):
        img = _photo(h, w, seed=w + h)
        for q in (35, 85, 98):
            data = _encode(img, quality=q, subsampling=sub)
            rc, info, coef = _split(data)
            if sub and (w + 1) // 2 <= 2:
                assert rc == L.JPEG_UNSUPPORTED          # libjpeg uses the box filter there: left to PIL
                continue
            assert rc == 0, (w, h, q, rc)
            assert np.array_equal(jpeg_ref.pixels(info, coef), _pil(data)), (w, h, q)
            n += 1
    img = _photo(96, 130, seed=5)
    for kw in (dict(restart_marker_blocks=7), dict(restart_marker_rows=1), dict(optimize=True), dict(restart_marker_blocks=1)):
        data = _encode(img, quality=88, subsampling=sub, **kw)
        rc, info, coef = _split(data)
        assert rc == 0, (kw, rc)
        if 'optimize' not in kw:
            assert info.restart_interval > 0
        assert np.array_equal(jpeg_ref.pixels(info, coef), _pil(data)), kw
    noise = np.random.default_rng(1).integers(0, 256, (41, 67, 3), dtype=np.uint8)          # every coefficient large
    for q in (30, 100):
        data = _encode(noise, quality=q, subsampling=sub)
        rc, info, coef = _split(data)
        assert rc == 0 and np.array_equal(jpeg_ref.pixels(info, coef), _pil(data)), q
    assert n >= 20


def test_greyscale_and_real_photographs():
    grey = _photo(70, 90)[:, :, 0]
    data = _encode(grey, quality=80)
    rc, info, coef = _split(data)
    assert rc == 0 and info.ncomp == 1
    assert np.array_equal(jpeg_ref.pixels(info, coef), _pil(data))
    import sklearn
    d = os.path.join(os.path.dirname(sklearn.__file__), 'datasets', 'images')
    for f in ('china.jpg', 'flower.jpg'):                 # camera files as they are (4:4:4, quantisation tables of their own)
        data = open(os.path.join(d, f), 'rb').read()
        rc, info, coef = _split(data)
        assert rc == 0, f
        assert np.array_equal(jpeg_ref.pixels(info, coef), _pil(data)), f


def test_files_the_split_decoder_does_not_take_are_reported():
    from PIL import Image
    img = _photo(64, 64)
    assert _split(_encode(img, progressive=True))[0] == L.JPEG_UNSUPPORTED
    b = io.BytesIO()
    Image.fromarray(img).convert('CMYK').save(b, 'JPEG')
    assert _split(b.getvalue())[0] == L.JPEG_UNSUPPORTED
    data = _encode(img, quality=90, subsampling=2)
    assert _split(data[:len(data) // 2])[0] == L.JPEG_CORRUPT          # truncated inside the scan (PIL refuses it too)
    assert _split(data[:100])[0] == L.JPEG_CORRUPT                     # truncated inside the headers
    assert _split(b'not a jpeg at all')[0] == L.JPEG_CORRUPT
    bad = bytearray(data)
    bad[len(bad) // 2:len(bad) // 2 + 64] = b'\xff' * 64               # garbage in the scan: an error code or pixels, no crash
    assert _split(bytes(bad))[0] in (L.JPEG_OK, L.JPEG_CORRUPT)


def test_mutated_files_give_a_code_never_a_crash():
    """Truncations, byte flips in headers and scan, runs of 0xFF: an error code or (garbage) coefficients, within the buffers.
    (tools/fuzz_jpeg.c is the same loop for the address / undefined-behaviour sanitizers.)"""
    rng = np.random.default_rng(11)
    seeds = [_encode(_photo(61, 83, seed=1), quality=88, subsampling=2),
             _encode(_photo(40, 57, seed=2), quality=70, subsampling=1, restart_marker_blocks=3),
             _encode(_photo(33, 33, seed=3)[:, :, 0], quality=80)]
    codes = set()
    for data in seeds:
        n = len(data)
        for it in range(150):
            d = bytearray(data)
            kind = it % 4
            if kind == 0:
                d = d[:1 + int(rng.integers(n - 1))]
            elif kind == 3:
                at = int(rng.integers(n))
                d[at:at + 40] = b'\xff' * len(d[at:at + 40])
            else:
                for _ in range(1 + int(rng.integers(8))):
                    d[int(rng.integers(min(n, 600) if kind == 1 else n))] = int(rng.integers(256))
            codes.add(_split(bytes(d))[0])
    assert codes <= {L.JPEG_OK, L.JPEG_UNSUPPORTED, L.JPEG_CORRUPT} and L.JPEG_CORRUPT in codes


def test_pool_decodes_queued_batches_back_to_back(tmp_path):
    lib = L.load_jpeg()
    sizes = [(64, 48), (33, 70), (120, 90), (16, 16), (50, 50), (71, 29), (90, 120)]
    paths, want, counts = [], [], {}
    for i, (w, h) in enumerate(sizes):
        data = _encode(_photo(h, w, seed=i), quality=70 + 3 * i, subsampling=i % 3)
        p = str(tmp_path / ('%d.jpg' % i))
        open(p, 'wb').write(data)
        paths.append(p)
        want.append(_pil(data))
        counts[p] = int(_split(data)[1].coef_count)
    prog = str(tmp_path / 'prog.jpg')
    open(prog, 'wb').write(_encode(_photo(40, 40), progressive=True))
    big = str(tmp_path / 'big.jpg')
    open(big, 'wb').write(_encode(_photo(200, 200), quality=90, subsampling=0))
    missing = str(tmp_path / 'missing.jpg')
    batch = paths + [prog, big, missing]
    capacity = sum(counts.values()) + 1000                  # every small file fits wherever the 200 x 200 one stands; it never does
    pool = lib.comic_jpeg_pool_create(3)
    assert pool
    runs = []
    for rep in range(4):                                    # several batches queued before the first wait
        order = batch[rep:] + batch[:rep]
        n = len(order)
        infos = np.zeros(n, L.JPEG_INFO_DTYPE)
        status = np.full(n, 99, np.int32)
        coef = np.full(capacity, 7, np.int16)
        arr = (C.c_char_p * n)(*[os.fsencode(p) for p in order])
        h = lib.comic_jpeg_pool_submit(pool, arr, n, infos.ctypes.data, status.ctypes.data, coef.ctypes.data, capacity)
        assert h
        runs.append((order, infos, status, coef, h))
    for order, infos, status, coef, h in runs:
        used, total = C.c_int64(-1), C.c_int64(-1)
        assert lib.comic_jpeg_pool_wait(pool, h, 60.0, C.byref(used), C.byref(total)) == 0
        off = base = 0
        for i, p in enumerate(order):
            if p == prog:
                assert status[i] == L.JPEG_UNSUPPORTED
            elif p == big:
                assert status[i] == L.JPEG_TOO_SMALL and infos['coef_count'][i] > 1000
            elif p == missing:
                assert status[i] == L.JPEG_IO
            else:
                assert status[i] == 0 and infos['coef_base'][i] == base and infos['pixel_off'][i] == off
                info = infos[i]
                img = jpeg_ref.pixels(info, coef[base:base + counts[p]])
                assert np.array_equal(img, want[paths.index(p)]), p
                off += (img.size + 15) // 16 * 16
                base += counts[p]
        assert total.value == off and used.value == base == sum(counts.values())
    lib.comic_jpeg_pool_destroy(pool)
    # destroy with work still queued: the pool finishes it first
    pool = lib.comic_jpeg_pool_create(2)
    n = len(paths)
    infos, status, coef = np.zeros(n, L.JPEG_INFO_DTYPE), np.full(n, 99, np.int32), np.zeros(capacity, np.int16)
    arr = (C.c_char_p * n)(*[os.fsencode(p) for p in paths])
    assert lib.comic_jpeg_pool_submit(pool, arr, n, infos.ctypes.data, status.ctypes.data, coef.ctypes.data, capacity)
    lib.comic_jpeg_pool_destroy(pool)
    assert (status == 0).all()


def test_packed_batches_hold_the_coefficients_of_the_dense_decode(tmp_path):
    """comic_jpeg_pool_submit_packed (the loader's form: a descriptor and the DC value per block, a 16-bit entry per non-zero AC coefficient) against the
    dense decode of the same files; with the cache on, a second pass returns the same words."""
    lib = L.load_jpeg()
    paths, dense = [], {}
    for i in range(8):
        p = str(tmp_path / ('p%d.jpg' % i))
        kw = dict(quality=55 + 6 * i, subsampling=i % 3)
        if i == 5:
            kw['restart_marker_blocks'] = 4
        data = _encode(_photo(33 + 9 * i, 70 + 3 * i, seed=i) if i != 6 else _photo(50, 50, seed=6)[:, :, 0], **kw)
        open(p, 'wb').write(data)
        paths.append(p)
        dense[p] = _split(data)
    noisy = str(tmp_path / 'noisy.jpg')                    # full-swing stripes at quality 100: AC coefficients beyond +-511 (two-word entries)
    stripes = np.zeros((40, 64), np.uint8)
    stripes[:, 0::2] = 255
    stripes[::3, :] = 255 - stripes[::3, :]
    data = _encode(stripes, quality=100)
    open(noisy, 'wb').write(data)
    paths.append(noisy)
    dense[noisy] = _split(data)
    assert np.abs(dense[noisy][2].reshape(-1, 64)[:, 1:]).max() > 511
    prog = str(tmp_path / 'prog.jpg')
    open(prog, 'wb').write(_encode(_photo(40, 40), progressive=True))
    order = paths[:3] + [prog, str(tmp_path / 'missing.jpg')] + paths[3:]
    pool = lib.comic_jpeg_pool_create(3)
    assert lib.comic_jpeg_pool_enable_cache(pool, 32 << 20) == 0
    cap = 800_000
    for rep in range(2):
        n = len(order)
        infos, status, blob = np.zeros(n, L.JPEG_INFO_DTYPE), np.full(n, 99, np.int32), np.full(cap, 0xdead, np.uint16)
        arr = (C.c_char_p * n)(*[os.fsencode(p) for p in order])
        h = lib.comic_jpeg_pool_submit_packed(pool, arr, n, infos.ctypes.data, status.ctypes.data, blob.ctypes.data, cap)
        used, planes = C.c_int64(), C.c_int64()
        assert h and lib.comic_jpeg_pool_wait(pool, h, 60.0, C.byref(used), C.byref(planes)) == 0
        spans, base = [], 0
        for i, p in enumerate(order):
            if p == prog:
                assert status[i] == L.JPEG_UNSUPPORTED
                continue
            if p.endswith('missing.jpg'):
                assert status[i] == L.JPEG_IO
                continue
            rc, dinfo, dcoef = dense[p]
            assert status[i] == 0 and infos['coef_count'][i] == dinfo.coef_count and infos['coef_base'][i] == base
            base += int(dinfo.coef_count)
            blocks = int(dinfo.coef_count) // 64
            off = int(infos['pixel_off'][i])
            ac = dcoef.reshape(-1, 64)[:, 1:]
            n16 = 3 * blocks + int(np.count_nonzero(ac)) + int(np.count_nonzero((ac < -512) | (ac > 511)))
            n16 += n16 & 1
            assert off % 2 == 0
            img = blob[off:off + n16]
            assert np.array_equal(jpeg_ref.unpack(img, blocks), dcoef), p
            spans.append((off, off + n16))
        spans.sort()
        assert spans[0][0] == 0 and all(a[1] == b[0] for a, b in zip(spans, spans[1:])) and spans[-1][1] == used.value
        assert planes.value == base
    b, e, hits = C.c_int64(), C.c_int64(), C.c_int64()
    lib.comic_jpeg_pool_cache_stats(pool, C.byref(b), C.byref(e), C.byref(hits))
    assert e.value == 9 and hits.value == 9
    # a blob with room for some of the images only: the others are reported, nothing is written past the end
    n = len(paths)
    small = 6000
    infos, status, blob = np.zeros(n, L.JPEG_INFO_DTYPE), np.full(n, 99, np.int32), np.full(small + 64, 0xdead, np.uint16)
    arr = (C.c_char_p * n)(*[os.fsencode(p) for p in paths])
    h = lib.comic_jpeg_pool_submit_packed(pool, arr, n, infos.ctypes.data, status.ctypes.data, blob.ctypes.data, small)
    used, planes = C.c_int64(), C.c_int64()
    assert lib.comic_jpeg_pool_wait(pool, h, 60.0, C.byref(used), C.byref(planes)) == 0
    assert set(status.tolist()) == {0, L.JPEG_TOO_SMALL} and used.value <= small and (blob[small:] == 0xdead).all()
    lib.comic_jpeg_pool_destroy(pool)


def test_coefficient_cache_serves_the_second_pass_from_memory(tmp_path):
    """comic_jpeg_pool_enable_cache: the batches of a second pass over the files come from the cache (hits counted) with the
    coefficients of the first pass, bit for bit; a file REWRITTEN since (other size / modification time) is decoded from disk
    again, not served stale; the byte limit stops insertion, nothing is evicted."""
    lib = L.load_jpeg()
    paths = []
    for i in range(9):
        p = str(tmp_path / ('c%d.jpg' % i))
        open(p, 'wb').write(_encode(_photo(40 + 7 * i, 60 + 5 * i, seed=i), quality=60 + 4 * i, subsampling=i % 3))
        paths.append(p)
    cap = 2_000_000

    def run(pool, order):
        n = len(order)
        infos, status, coef = np.zeros(n, L.JPEG_INFO_DTYPE), np.full(n, 99, np.int32), np.full(cap, 7, np.int16)
        arr = (C.c_char_p * n)(*[os.fsencode(p) for p in order])
        h = lib.comic_jpeg_pool_submit(pool, arr, n, infos.ctypes.data, status.ctypes.data, coef.ctypes.data, cap)
        used, total = C.c_int64(), C.c_int64()
        assert lib.comic_jpeg_pool_wait(pool, h, 60.0, C.byref(used), C.byref(total)) == 0
        return infos, status, coef[:used.value].copy()

    def stats(pool):
        b, e, h = C.c_int64(), C.c_int64(), C.c_int64()
        assert lib.comic_jpeg_pool_cache_stats(pool, C.byref(b), C.byref(e), C.byref(h)) == 0
        return b.value, e.value, h.value
    pool = lib.comic_jpeg_pool_create(3)
    assert lib.comic_jpeg_pool_enable_cache(pool, 64 << 20) == 0
    i1, s1, c1 = run(pool, paths)
    b, e, h = stats(pool)
    assert (s1 == 0).all() and e == 9 and h == 0 and 0 < b < c1.size * 2          # packed: smaller than the dense blocks
    i2, s2, c2 = run(pool, paths)
    assert (s2 == 0).all() and stats(pool)[2] == 9 and np.array_equal(c1, c2)
    for f in ('width', 'height', 'coef_count', 'coef_base', 'pixel_off', 'quant', 'comp_w', 'blocks_w'):
        assert np.array_equal(i1[f], i2[f]), f
    i3, s3, c3 = run(pool, paths[::-1])                                            # another order: other bases, same images
    assert (s3 == 0).all()
    for k, p in enumerate(paths[::-1]):
        j = paths.index(p)
        a = c3[int(i3['coef_base'][k]):int(i3['coef_base'][k]) + int(i3['coef_count'][k])]
        assert np.array_equal(a, c1[int(i1['coef_base'][j]):int(i1['coef_base'][j]) + int(i1['coef_count'][j])]), p
    # a file rewritten under the same path: its entry is not served any more, the eight others are
    hits0 = stats(pool)[2]
    open(paths[0], 'wb').write(_encode(_photo(33, 47, seed=77), quality=80, subsampling=0))
    i4, s4, c4 = run(pool, paths)
    assert (s4 == 0).all() and stats(pool)[2] == hits0 + 8
    assert (int(i4['width'][0]), int(i4['height'][0])) == (47, 33) and (int(i1['width'][0]), int(i1['height'][0])) == (60, 40)
    for f in ('width', 'height', 'coef_count'):
        assert np.array_equal(i1[f][1:], i4[f][1:]), f
    # ... and the fresh decode took its place in the cache (it used to be dropped as a duplicate of the stale entry, so the file
    # was decoded again in every later pass): the next pass is nine hits, with the rewritten image
    assert stats(pool)[1] == 10
    i5, s5, c5 = run(pool, paths)
    assert (s5 == 0).all() and stats(pool)[2] == hits0 + 8 + 9 and np.array_equal(c5, c4)
    assert (int(i5['width'][0]), int(i5['height'][0])) == (47, 33)
    lib.comic_jpeg_pool_destroy(pool)
    # a limit that holds about two images: insertion stops there
    pool = lib.comic_jpeg_pool_create(2)
    assert lib.comic_jpeg_pool_enable_cache(pool, 40_000) == 0
    run(pool, paths[4:])
    b, e, _ = stats(pool)
    assert 0 < e < 5 and b <= 40_000
    run(pool, paths[4:])
    assert stats(pool)[1] == e and stats(pool)[2] == e
    lib.comic_jpeg_pool_destroy(pool)


# ---- device half -----------------------------------------------------------------------------------------------------------
def _mixed_files(tmp_path, count=12):
    import sklearn
    from PIL import Image
    d = os.path.join(os.path.dirname(sklearn.__file__), 'datasets', 'images')
    photos = [np.asarray(Image.open(os.path.join(d, f)).convert('RGB')) for f in ('china.jpg', 'flower.jpg')]
    sizes = [(640, 480), (480, 640), (640, 427), (500, 375), (333, 500), (640, 640), (97, 131), (17, 9), (161, 3), (9, 200)]
    paths = []
    for i in range(count):
        w, h = sizes[i % len(sizes)]
        src = Image.fromarray(photos[i % 2]).resize((w, h)) if i % 4 != 3 else Image.fromarray(_photo(h, w, seed=i))
        if i % 6 == 5:
            src = src.convert('L')
        p = str(tmp_path / ('m%d.jpg' % i))
        kw = dict(quality=(60, 75, 90, 96)[i % 4], subsampling=(2, 2, 1, 0)[i % 4])
        if i % 5 == 4:
            kw['restart_marker_blocks'] = 11
        src.save(p, 'JPEG', **kw)
        paths.append(p)
    return paths


@pytest.mark.gpu
def test_device_pixels_are_pils_pixels(tmp_path):
    import torch
    lib, jl = L.load(), L.load_jpeg()
    paths = _mixed_files(tmp_path)
    n = len(paths)
    slot = 640 * 640 * 3
    infos = np.zeros(n, L.JPEG_INFO_DTYPE)
    coef = torch.zeros(n * slot, dtype=torch.int16)
    off = 0
    for i, p in enumerate(paths):
        info = L.JpegInfo()
        rc = jl.comic_jpeg_decode_file(os.fsencode(p), C.byref(info), coef.data_ptr() + 2 * i * slot, slot)
        assert rc == 0, (p, rc)
        info.coef_base, info.pixel_off = i * slot, off
        off += (info.width * info.height * 3 + 15) // 16 * 16
        infos[i] = np.frombuffer(bytes(info), L.JPEG_INFO_DTYPE)[0]
    dev_coef = coef.cuda()
    dev_infos = torch.from_numpy(infos.view(np.uint8)).cuda()
    planes = torch.empty(n * slot, dtype=torch.uint8, device='cuda')
    pixels = torch.zeros(off, dtype=torch.uint8, device='cuda')
    L.check(lib.comic_jpeg_pixels(dev_coef.data_ptr(), dev_infos.data_ptr(), n, int(infos['coef_count'].max()) // 64,
                                  int(infos['width'].max()), int(infos['height'].max()), planes.data_ptr(), pixels.data_ptr(),
                                  L.stream_ptr()), 'jpeg_pixels')
    got = pixels.cpu().numpy()
    for i, p in enumerate(paths):
        want = _pil(open(p, 'rb').read())
        o = int(infos['pixel_off'][i])
        img = got[o:o + want.size].reshape(want.shape)
        assert np.array_equal(img, want), (p, np.argwhere(img != want)[:4])
        # and the oracle agrees with both
        assert np.array_equal(jpeg_ref.pixels(infos[i], coef.numpy()[i * slot:i * slot + int(infos['coef_count'][i])]), want)
    # comic_jpeg_preprocess (dense blocks -> network input, the RGB image never written) == comic_image_preprocess on the RGB blob
    from comic_amd import inputs
    desc = np.zeros(n, inputs.DevicePreprocessor._DESC_DTYPE)
    desc['offset'], desc['in_h'], desc['in_w'] = infos['pixel_off'], infos['height'], infos['width']
    desc['flip'], desc['oy'], desc['ox'] = np.arange(n) % 2, (np.arange(n) * 7) % 33, (np.arange(n) * 5) % 33
    desc['sy'] = (infos['height'] / 256).astype(np.float32)
    desc['sx'] = (infos['width'] / 256).astype(np.float32)
    dev_desc = torch.from_numpy(desc.view(np.uint8)).cuda()
    a = torch.empty((n, 224, 224, 3), dtype=torch.float32, device='cuda')
    b = torch.empty_like(a)
    L.check(lib.comic_image_preprocess(pixels.data_ptr(), dev_desc.data_ptr(), n, a.data_ptr(), 224, 224, 256, L.stream_ptr()), 'pre')
    planes.zero_()
    L.check(lib.comic_jpeg_preprocess(dev_coef.data_ptr(), dev_infos.data_ptr(), n, int(infos['coef_count'].max()) // 64,
                                      planes.data_ptr(), None, dev_desc.data_ptr(), b.data_ptr(), 224, 224, 256, L.stream_ptr()),
            'jpeg_preprocess')
    assert torch.equal(a, b)


@pytest.mark.gpu
def test_split_loader_batches_are_bit_identical_to_the_thread_decode(tmp_path):
    """The loader path end to end (pool -> pinned slot -> strided copy -> comic_jpeg_pixels -> comic_image_preprocess) against
    PIL decode + the same device preprocessing; a progressive file and the files the staging slot has no room for take the
    PIL path inside."""
    import torch
    from comic_amd import inputs
    paths = _mixed_files(tmp_path, 10)
    prog = str(tmp_path / 'prog.jpg')
    open(prog, 'wb').write(_encode(_photo(300, 400), progressive=True, quality=85))
    paths.insert(3, prog)
    from PIL import Image
    png = str(tmp_path / 'lossless.png')                   # not a JPEG at all: PIL path too
    Image.fromarray(_photo(120, 90, seed=9)).save(png)
    paths.append(png)
    pre = inputs.DevicePreprocessor('cuda:0', 224, 224)
    jpool = inputs.JpegSplitPool(4, slot_elems=200000, max_batch=16)       # room for about half of the files: the rest -> PIL
    pre.enable_split(jpool, 3)
    params = [(bool(i % 2), (i * 7) % 33, (i * 5) % 33) for i in range(len(paths))]
    ref = pre(list(map(inputs.decode_image, paths)), params).cpu()
    inflight = [pre.pack_paths_split(paths, params) for _ in range(3)]
    for packed in inflight:
        got = pre.finish(packed).cpu()
        assert torch.equal(got, ref)
    torch.cuda.synchronize()
    pre._reap()
    assert pre._free_coef.qsize() == 3                     # every staging slot came back
    # files of the PIL path are remembered: later batches start their decode at pack time (same tensors)
    assert prog in pre._pil_known and png in pre._pil_known
    packed = pre.pack_paths_split(paths, params)
    assert sorted(packed.slot[6]) == sorted(paths.index(p) for p in pre._pil_known if p in paths)
    assert torch.equal(pre.finish(packed).cpu(), ref)
    torch.cuda.synchronize()
    pre._reap()
    # a file that does not exist: the error of the PIL path reaches the consumer, the slot comes back
    with pytest.raises(FileNotFoundError):
        pre.finish(pre.pack_paths_split(paths[:2] + [str(tmp_path / 'missing.jpg')], params[:3]))
    assert pre._free_coef.qsize() == 3
    jpool.close()
    with pytest.raises(RuntimeError, match='closed'):
        pre.pack_paths_split(paths[:2], params[:2])


@pytest.mark.gpu
def test_input_manager_with_split_decode_yields_the_batches_of_the_thread_decode(tmp_path):
    """config.loader_split_jpeg (train.py / infer.py --loader_split_jpeg): the managers' batches -- images on the device,
    captions -- are the ones of the PIL loader (decode threads + device preprocessing, --no-loader_split_jpeg), bit for bit."""
    import torch
    from tests import tiny_dataset
    from comic_amd import inputs, configuration as conf
    ds = tiny_dataset.make(str(tmp_path / 'mscoco'), n_train=24, n_valid=4, n_test=4)
    kw = dict(dataset_dir=ds, dataset_file_pattern='mscoco_{}_w5_s20_include_restval', cnn_name='inception_v3',
              cnn_input_size=[224, 224], cnn_input_augment=True, batch_size_train=8, batch_size_eval=2, max_epoch=3,
              rand_seed=7, token_type='radix', radix_base=256, loader_threads=4)
    a = inputs.InputManager_Radix(conf.Config(loader_split_jpeg=True, **kw))
    b = inputs.InputManager_Radix(conf.Config(loader_split_jpeg=False, **kw))
    try:
        a.enable_device_preprocess('cuda:0')
        b.enable_device_preprocess('cuda:0')
        assert a._jpeg_pool is not None and getattr(b, '_jpeg_pool', None) is None
        on_device = 0
        for it_a, it_b in ((a.batch_train, b.batch_train), (a.batch_eval, b.batch_eval)):
            for _ in range(8):
                (ia, ca), (ib, cb) = next(it_a), next(it_b)
                # (the batches the prefetch threads made before enable_device_preprocess are numpy arrays: same bits)
                on_device += int(torch.is_tensor(ia) and ia.is_cuda)
                ta, tb = torch.as_tensor(ia).cpu(), torch.as_tensor(ib).cpu()
                assert ta.dtype == torch.float32 and tuple(ta.shape[1:]) == (224, 224, 3)
                assert torch.equal(ta, tb)
                np.testing.assert_array_equal(ca, cb)
        assert on_device >= 4                          # the split decoder did serve batches
    finally:
        a.close()
        b.close()
