"""Op-level parity: every HIP kernel behind the C-ABI vs the CPU oracle on the same seeded
inputs.  Tolerances: fp32 1e-3 relative (max-norm), bf16 storage 2e-2 vs a bf16-emulating
oracle (inputs/weights rounded to bf16, fp32 accumulate), indices bit-exact."""
import ctypes as C
import math

import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu

import comic_amd._lib as L
from oracle import beam_ref, cnn_ref, decoder_ref as dr
from tests.gpu_util import DEV, F32_RTOL, P, assert_close, dev, lib, rel_err, stream, sync


# ------------------------------------------------------------------------ GEMM -----------
@pytest.mark.parametrize('M,N,K,ta,tb', [
    (64, 2048, 1280, 0, 0), (64, 512, 512, 0, 0), (64, 258, 512, 0, 0), (37, 258, 130, 0, 0),
    (64, 1280, 2048, 0, 1), (64, 512, 258, 0, 1), (1856, 512, 258, 0, 1),
    (1280, 2048, 1856, 1, 0), (512, 258, 1856, 1, 0), (2048, 768, 64, 1, 0), (33, 70, 19, 1, 1),
    (1600, 512, 2048, 0, 0), (2048, 512, 1600, 1, 0)])
def test_gemm_f32(M, N, K, ta, tb):
    rng = np.random.default_rng(M + N + K)
    A = rng.standard_normal((K, M) if ta else (M, K)).astype(np.float32)
    B = rng.standard_normal((N, K) if tb else (K, N)).astype(np.float32)
    bias = rng.standard_normal(N).astype(np.float32)
    C0 = rng.standard_normal((M, N)).astype(np.float32)
    ref = (A.T if ta else A).astype(np.float64) @ (B.T if tb else B).astype(np.float64)
    dA, dB, dC, db = dev(A), dev(B), dev(C0), dev(bias)
    L.check(lib().comic_gemm_f32(dA.data_ptr(), dB.data_ptr(), dC.data_ptr(), db.data_ptr(), M, N, K,
                                 A.shape[1], B.shape[1], N, ta, tb, 0.5, 2.0, stream()))
    sync()
    assert_close(dC.cpu().numpy(), 0.5 * ref + 2.0 * C0 + bias, 1e-5, 'gemm')
    # plain product, strided C (ldc > N)
    big = torch.zeros((M, N + 8), dtype=torch.float32, device=DEV)
    L.check(lib().comic_gemm_f32(dA.data_ptr(), dB.data_ptr(), big.data_ptr(), None, M, N, K, A.shape[1],
                                 B.shape[1], N + 8, ta, tb, 1.0, 0.0, stream()))
    sync()
    assert_close(big[:, :N].cpu().numpy(), ref, 1e-5, 'gemm strided')
    assert float(big[:, N:].abs().max()) == 0.0


@pytest.mark.parametrize('R,K,N', [(150, 512, 512), (33, 96, 70), (256, 1000, 258), (64, 2816, 2048), (224, 512, 258),
                                   (100, 40, 1), (35, 520, 130)])
def test_gemm_f32_stream(R, K, N):
    """Weight-streaming skinny product of the decode steps (csrc/lstm_stream.hip through its C-ABI entry): rows on one and
    two 16-row tiles per wave and with idle waves, K not a multiple of the 32-deep step, N not a multiple of the
    64-column chunk, the LSTM gate product's shape; bf16 hi/lo split products, bound 5e-5 of the result's max-norm."""
    rng = np.random.default_rng(R + K + N)
    x = rng.standard_normal((R, K)).astype(np.float32)
    W = rng.standard_normal((K, N)).astype(np.float32)
    b = rng.standard_normal(N).astype(np.float32)
    ref = x.astype(np.float64) @ W.astype(np.float64) + b
    nb = int(lib().comic_gemm_f32_stream_workspace(R, K, N))
    ws = torch.empty(nb, dtype=torch.uint8, device=DEV)
    dx, dW, db = dev(x), dev(W), dev(b)
    out = torch.full((R, N), float('nan'), dtype=torch.float32, device=DEV)
    for _ in range(3):
        L.check(lib().comic_gemm_f32_stream(dx.data_ptr(), dW.data_ptr(), db.data_ptr(), out.data_ptr(), R, K, N,
                                            ws.data_ptr(), nb, stream()))
    sync()
    assert_close(out.cpu().numpy(), ref, 5e-5, 'gemm stream')
    out2 = torch.empty_like(out)
    L.check(lib().comic_gemm_f32_stream(dx.data_ptr(), dW.data_ptr(), None, out2.data_ptr(), R, K, N, ws.data_ptr(), nb,
                                        stream()))
    sync()
    assert_close(out2.cpu().numpy(), ref - b, 5e-5, 'gemm stream, no bias')
    with pytest.raises(L.ComicHipError):
        L.check(lib().comic_gemm_f32_stream(dx.data_ptr(), dW.data_ptr(), None, out2.data_ptr(), 16, K, N, ws.data_ptr(), nb,
                                            stream()))


@pytest.mark.parametrize('M,N,K,ta,tb', [
    (1600, 512, 2048, 0, 0), (1856, 258, 512, 0, 0), (1856, 512, 258, 0, 1), (1280, 2048, 1856, 1, 0),
    (512, 258, 1856, 1, 0), (2048, 512, 1600, 1, 0), (33, 70, 19, 1, 1), (130, 131, 45, 0, 1), (257, 129, 64, 0, 0),
    # decode-step shapes: all rows in one 160- / 192-row tile over a wide (unaligned / aligned) vocabulary
    (150, 8195, 512, 0, 0), (150, 8196, 96, 0, 1), (190, 8200, 100, 0, 0), (161, 8300, 64, 1, 0)])
def test_gemm_f32_split3(M, N, K, ta, tb):
    """bf16 hi/lo split products (hi*hi + hi*lo + lo*hi, fp32 accumulate): every layout, ragged tiles,
    alpha/beta/bias and a strided C; error bound 5e-5 of the result's max-norm (measured ~2e-6)."""
    rng = np.random.default_rng(M + N + K)
    A = rng.standard_normal((K, M) if ta else (M, K)).astype(np.float32)
    B = rng.standard_normal((N, K) if tb else (K, N)).astype(np.float32)
    bias = rng.standard_normal(N).astype(np.float32)
    C0 = rng.standard_normal((M, N)).astype(np.float32)
    ref = (A.T if ta else A).astype(np.float64) @ (B.T if tb else B).astype(np.float64)
    dA, dB, dC, db = dev(A), dev(B), dev(C0), dev(bias)
    L.check(lib().comic_gemm_f32_split3(dA.data_ptr(), dB.data_ptr(), dC.data_ptr(), db.data_ptr(), M, N, K,
                                        A.shape[1], B.shape[1], N, ta, tb, 0.5, 2.0, None, 0, stream()))
    sync()
    assert_close(dC.cpu().numpy(), 0.5 * ref + 2.0 * C0 + bias, 5e-5, 'gemm split3')
    # with a workspace: split-K slabs for the shapes with few output tiles, same contract
    ws = torch.empty(8 << 20, dtype=torch.uint8, device=DEV)
    dC2 = dev(C0)
    L.check(lib().comic_gemm_f32_split3(dA.data_ptr(), dB.data_ptr(), dC2.data_ptr(), db.data_ptr(), M, N, K,
                                        A.shape[1], B.shape[1], N, ta, tb, 0.5, 2.0, ws.data_ptr(), 8 << 20, stream()))
    sync()
    assert_close(dC2.cpu().numpy(), 0.5 * ref + 2.0 * C0 + bias, 5e-5, 'gemm split3 + split-K')
    big = torch.zeros((M, N + 8), dtype=torch.float32, device=DEV)
    L.check(lib().comic_gemm_f32_split3(dA.data_ptr(), dB.data_ptr(), big.data_ptr(), None, M, N, K, A.shape[1],
                                        B.shape[1], N + 8, ta, tb, 1.0, 0.0, ws.data_ptr(), 8 << 20, stream()))
    sync()
    assert_close(big[:, :N].cpu().numpy(), ref, 5e-5, 'gemm split3 strided')
    assert float(big[:, N:].abs().max()) == 0.0


_GG_SPECS = {  # type, M, N, K, extras
    # any alignment: the four-wave kernel with scalar loads where rows are not 16-byte aligned
    'mixed': [
        (0, 1280, 2048, 1920, {}), (0, 2048, 512, 1600, {}), (2, 1856, 256, 2048, {'mask': 768}),
        (0, 512, 258, 1856, {}), (1, 1600, 512, 2048, {}), (1, 64, 768, 2048, {'mask': 768, 'ldc': 1280}),
        (0, 1, 2048, 1920, {'ones': True}), (0, 1, 258, 1856, {'ones': True}), (0, 1, 1, 256, {'ones': True, 'ldb': 1537}),
        (2, 64, 768, 2048, {'bias': True}), (0, 33, 70, 19, {'beta': 2.0}), (1, 130, 131, 45, {'bias': True, 'beta': -1.0}),
        (2, 257, 129, 64, {}), (1, 1856, 258, 512, {'bias': True})],
    # every operand row 16-byte aligned: the producer / consumer kernel (the training step's shapes, ragged tiles, short
    # and odd k-tile counts, N not a multiple of 4 inside a padded row)
    'aligned': [
        (0, 1280, 2048, 1920, {}), (0, 2048, 512, 1600, {}), (2, 1856, 256, 2048, {'mask': 768}),
        (0, 512, 258, 1856, {'ldb': 260}), (1, 1600, 512, 2048, {}), (1, 64, 768, 2048, {'mask': 768, 'ldc': 1280}),
        (0, 1, 2048, 1920, {'ones': True}), (0, 1, 258, 1856, {'ones': True, 'ldb': 260}), (0, 1, 1, 256, {'ones': True, 'ldb': 1537}),
        (2, 64, 768, 2048, {'bias': True}), (0, 36, 72, 20, {'beta': 2.0}), (1, 132, 130, 44, {'bias': True, 'beta': -1.0, 'ldb': 132}),
        (2, 257, 129, 64, {}), (1, 1856, 258, 512, {'bias': True, 'ldb': 260}), (2, 1856, 512, 260, {}), (1, 100, 100, 100, {}),
        (0, 100, 100, 300, {})],
}


@pytest.mark.parametrize('which', ['mixed', 'aligned'])
def test_gemm_group_matches_numpy_and_is_reproducible(which):
    """comic_gemm_group: the three operand layouts, bias / dropout-mask / beta epilogues, column sums as products, ragged
    tiles and split-K tiles combined inside the launch -- every problem against numpy in float64 (5e-5 of the
    max-norm, as comic_gemm_f32_split3), a second launch over the same buffers bit-equal to the first."""
    rng = np.random.default_rng(7)
    f = lambda *s: rng.standard_normal(s).astype(np.float32)
    keep = 0.75
    specs = _GG_SPECS[which]
    probs = (L.GemmProb * len(specs))()
    keepalive, refs, outs = [], [], []
    for i, (ty, M, N, K, ex) in enumerate(specs):
        ones = ex.get('ones', False)
        ldb = ex.get('ldb', K if ty == 2 else N)
        A = None if ones else (f(K, M) if ty == 0 else f(M, K))
        Bfull = f(N, ldb) if ty == 2 else f(K, ldb)
        Bm = Bfull[:, :K].T if ty == 2 else Bfull[:, :N]
        Am = np.ones((1, K), np.float32) if ones else (A.T if ty == 0 else A)
        ref = Am.astype(np.float64) @ Bm.astype(np.float64)
        ldc = ex.get('ldc', N)
        C0 = f(M, ldc)
        if ex.get('bias'):
            bias = f(N); ref = ref + bias
        else:
            bias = None
        if 'mask' in ex:
            ldm = ex['mask']
            mask = (rng.uniform(size=(M, ldm)) < keep).astype(np.float32)
            ref = ref / keep * mask[:, :N]
        else:
            mask, ldm = None, 0
        beta = ex.get('beta', 0.0)
        ref = ref + beta * C0[:, :N]
        dA = None if ones else dev(A)
        dB, dC = dev(Bfull), dev(C0)
        db = dev(bias) if bias is not None else None
        dm = dev(mask) if mask is not None else None
        keepalive += [dA, dB, dC, db, dm]
        q = probs[i]
        q.A = dA.data_ptr() if dA is not None else None
        q.B, q.C = dB.data_ptr(), dC.data_ptr()
        q.bias = db.data_ptr() if db is not None else None
        q.mask = dm.data_ptr() if dm is not None else None
        q.M, q.N, q.K, q.lda, q.ldb, q.ldc, q.ld_mask = M, N, K, (1 if ones else A.shape[1]), ldb, ldc, ldm
        q.alpha, q.beta, q.keep, q.type, q.ones_a = 1.0, beta, keep, ty, int(ones)
        refs.append(ref); outs.append((dC, C0, N))
    nbytes = lib().comic_gemm_group_workspace(probs, len(specs))
    assert nbytes > 0
    ws = torch.empty(int(nbytes), dtype=torch.uint8, device=DEV)
    L.check(lib().comic_gemm_group(probs, len(specs), ws.data_ptr(), nbytes, stream()), 'gemm_group')
    sync()
    first = []
    for (dC, C0, N), ref, sp in zip(outs, refs, specs):
        got = dC.cpu().numpy()
        assert_close(got[:, :N], ref, 5e-5, 'gemm_group %r' % (sp[:4],))
        assert np.array_equal(got[:, N:], C0[:, N:]), 'gemm_group wrote outside its columns'
        first.append(got.copy())
    # second launch into fresh outputs: same bits (fixed-order combine), tickets left clean by the first
    for (dC, C0, N) in outs:
        dC.copy_(torch.from_numpy(C0))
    L.check(lib().comic_gemm_group(probs, len(specs), ws.data_ptr(), nbytes, stream()), 'gemm_group')
    sync()
    for (dC, _, _), a in zip(outs, first):
        assert np.array_equal(dC.cpu().numpy(), a)


# ------------------------------------------------------------------------ conv / pool -----
def _run_conv(x, w, beta, mean, var, stride, padding, dtype, dst_channels=None, dst_coff=0, relu=1, out_f32=0, tile=0):
    B, H, W, Cin = x.shape
    kh, kw, _, Cout = w.shape
    Ho, pt, _ = cnn_ref.out_size(H, kh, stride, padding)
    Wo, pl, _ = cnn_ref.out_size(W, kw, stride, padding)
    tdt = torch.bfloat16 if dtype == 'bf16' else torch.float32
    code = 1 if dtype == 'bf16' else 0
    stem = Cin <= 4
    xd = dev(x) if stem else dev(x).to(tdt)
    wd = dev(w)
    scale = torch.empty(Cout, device=DEV); shift = torch.empty(Cout, device=DEV)
    L.check(lib().comic_fold_bn(P(beta), P(mean), P(var), 1e-3,
                                scale.data_ptr(), shift.data_ptr(), Cout, stream()))
    if stem:
        packed = wd.reshape(-1, Cout).contiguous()
    else:
        K = kh * kw * Cin
        packed = torch.empty(Cout * ((K + 63) // 64 * 64), dtype=tdt, device=DEV)
        L.check(lib().comic_pack_conv_weights(wd.data_ptr(), packed.data_ptr(), kh, kw, Cin, Cout, code, stream()))
    yc = dst_channels or Cout
    ydt = torch.float32 if out_f32 else tdt
    y = torch.full((B, Ho, Wo, yc), -7.0, dtype=ydt, device=DEV)
    op = L.CnnOp(kind=1 if stem else 0, src=0, dst=1, src_coff=0, dst_coff=dst_coff, H=H, W=W, Cin=Cin, Cout=Cout,
                 KH=kh, KW=kw, SH=stride, SW=stride, PT=pt, PL=pl, Ho=Ho, Wo=Wo, weight=0, relu=relu, out_f32=out_f32,
                 tile=tile)
    wt = L.ConvWeight(packed.data_ptr(), scale.data_ptr(), shift.data_ptr())
    L.check(lib().comic_conv2d_bn_relu(C.byref(op), xd.data_ptr(), Cin, y.data_ptr(), yc, C.byref(wt), B, code,
                                       stream()), 'conv')
    sync()
    return y.float().cpu().numpy()


def _ref_conv(x, w, beta, mean, var, stride, padding, dtype, relu=1, round_out=True):
    q = cnn_ref.bf16_round if dtype == 'bf16' else (lambda a: a)
    y = cnn_ref.conv2d(q(x), q(w), stride, padding)
    y = cnn_ref.batch_norm_inference(y, beta, mean, var)
    if relu:
        y = np.maximum(y, 0)
    return q(y) if round_out else y


CONV_CASES = [
    # B, H, W, Cin, Cout, (kh,kw), stride, padding
    (2, 17, 15, 32, 32, (3, 3), 1, 'VALID'), (2, 17, 15, 32, 64, (3, 3), 1, 'SAME'),
    (2, 21, 21, 64, 80, (1, 1), 1, 'VALID'), (1, 19, 19, 80, 192, (3, 3), 1, 'VALID'),
    (2, 25, 25, 48, 64, (5, 5), 1, 'SAME'), (3, 25, 25, 288, 384, (3, 3), 2, 'VALID'),
    (2, 12, 12, 128, 128, (1, 7), 1, 'SAME'), (2, 12, 12, 160, 192, (7, 1), 1, 'SAME'),
    (2, 5, 5, 448, 384, (3, 3), 1, 'SAME'), (2, 5, 5, 384, 384, (1, 3), 1, 'SAME'),
    (2, 5, 5, 2048, 320, (1, 1), 1, 'SAME'), (70, 12, 12, 768, 192, (1, 1), 1, 'SAME'),
    (9, 25, 25, 192, 48, (1, 1), 1, 'SAME'), (1, 1, 1, 32, 16, (1, 1), 1, 'SAME')]


@pytest.mark.parametrize('dtype', ['f32', 'bf16'])
@pytest.mark.parametrize('case', CONV_CASES)
def test_conv_bn_relu(case, dtype):
    B, H, W, Cin, Cout, k, s, pad = case
    rng = np.random.default_rng(B * 1000 + H * 7 + Cin + Cout + k[0])
    x = rng.standard_normal((B, H, W, Cin)).astype(np.float32)
    w = (rng.standard_normal((k[0], k[1], Cin, Cout)) / math.sqrt(k[0] * k[1] * Cin)).astype(np.float32)
    beta = 0.2 * rng.standard_normal(Cout).astype(np.float32)
    mean = 0.2 * rng.standard_normal(Cout).astype(np.float32)
    var = rng.uniform(0.5, 1.5, Cout).astype(np.float32)
    if dtype == 'bf16':
        x = cnn_ref.bf16_round(x)
    ref = _ref_conv(x, w, beta, mean, var, s, pad, dtype)
    got = _run_conv(x, w, beta, mean, var, s, pad, dtype)
    # bf16: one output ulp (2^-8) on top of accumulation-order noise
    assert_close(got, ref, 1e-4 if dtype == 'f32' else 1e-2, 'conv %s' % (case,))
    # concat slice: write at a channel offset of a wider buffer, neighbours untouched
    got2 = _run_conv(x, w, beta, mean, var, s, pad, dtype, dst_channels=Cout + 48, dst_coff=16)
    np.testing.assert_array_equal(got2[..., 16:16 + Cout], got)
    assert (got2[..., :16] == -7).all() and (got2[..., 16 + Cout:] == -7).all()


@pytest.mark.parametrize('case', [CONV_CASES[i] for i in (1, 3, 5, 7, 8, 11, 12)])
def test_conv_im2col_tile_variants_identical_bits(case):
    """Every im2col LDS-DMA variant (ids 1..12: tile shapes x pipeline depths; 26..47: wide tiles (35..47 with loader waves), 4 or 8 waves) is a
    different blocking of the same sums in the same k order: bit-identical outputs, ragged row / channel tiles
    included, and correct against the oracle."""
    B, H, W, Cin, Cout, k, s, pad = case
    rng = np.random.default_rng(B * 1000 + H * 7 + Cin + Cout + k[0] + 5)
    x = cnn_ref.bf16_round(rng.standard_normal((B, H, W, Cin)).astype(np.float32))
    w = (rng.standard_normal((k[0], k[1], Cin, Cout)) / math.sqrt(k[0] * k[1] * Cin)).astype(np.float32)
    beta = 0.2 * rng.standard_normal(Cout).astype(np.float32)
    mean = 0.2 * rng.standard_normal(Cout).astype(np.float32)
    var = rng.uniform(0.5, 1.5, Cout).astype(np.float32)
    base = _run_conv(x, w, beta, mean, var, s, pad, 'bf16', tile=3)
    assert_close(base, _ref_conv(x, w, beta, mean, var, s, pad, 'bf16'), 1e-2, 'conv %s' % (case,))
    for tile in list(range(1, 13)) + list(range(26, 48)):
        got = _run_conv(x, w, beta, mean, var, s, pad, 'bf16', tile=tile)
        np.testing.assert_array_equal(got, base, err_msg='tile %d %s' % (tile, case))


# patch-resident variants (tile ids 13..25, conv_patch.inc; 4, 8 and 12 waves per workgroup): stride 1, Cin >= 32.  The tile spans image
# boundaries (global output rows), ragged column tiles, SAME / VALID halos, the Kpad tail (K % 64 != 0),
# channel counts with and without the 32-byte pixel padding, Cout ragged against the channel tile.
PATCH_CASES = [
    # B, H, W, Cin, Cout, (kh,kw), padding
    (3, 17, 15, 32, 32, (3, 3), 'VALID'), (2, 37, 37, 32, 64, (3, 3), 'SAME'), (3, 19, 19, 80, 192, (3, 3), 'VALID'),
    (5, 25, 25, 48, 64, (5, 5), 'SAME'), (5, 25, 25, 96, 96, (3, 3), 'SAME'), (7, 12, 12, 128, 128, (1, 7), 'SAME'),
    (7, 12, 12, 160, 192, (7, 1), 'SAME'), (30, 5, 5, 448, 384, (3, 3), 'SAME'), (30, 5, 5, 384, 384, (1, 3), 'SAME'),
    (3, 21, 21, 64, 80, (1, 1), 'VALID'), (1, 1, 1, 32, 16, (1, 1), 'SAME'), (2, 40, 70, 32, 48, (3, 3), 'SAME')]


@pytest.mark.parametrize('tile', list(range(13, 26)) + list(range(48, 54)))
@pytest.mark.parametrize('case', PATCH_CASES)
def test_conv_patch_variants(case, tile):
    B, H, W, Cin, Cout, k, pad = case
    rng = np.random.default_rng(B * 1000 + H * 7 + Cin + Cout + k[0] + tile)
    x = cnn_ref.bf16_round(rng.standard_normal((B, H, W, Cin)).astype(np.float32))
    w = (rng.standard_normal((k[0], k[1], Cin, Cout)) / math.sqrt(k[0] * k[1] * Cin)).astype(np.float32)
    beta = 0.2 * rng.standard_normal(Cout).astype(np.float32)
    mean = 0.2 * rng.standard_normal(Cout).astype(np.float32)
    var = rng.uniform(0.5, 1.5, Cout).astype(np.float32)
    ref = _ref_conv(x, w, beta, mean, var, 1, pad, 'bf16')
    try:
        got = _run_conv(x, w, beta, mean, var, 1, pad, 'bf16', tile=tile)
    except L.ComicHipError as e:
        assert 'not eligible' in str(e) and Cin >= 128      # the input window of a fat-Cin layer does not fit the LDS
        pytest.skip(str(e))
    assert_close(got, ref, 1e-2, 'patch conv %s tile %d' % (case, tile))
    # same operands and k order per accumulator as the im2col kernel: identical bits
    np.testing.assert_array_equal(got, _run_conv(x, w, beta, mean, var, 1, pad, 'bf16', tile=3))
    got2 = _run_conv(x, w, beta, mean, var, 1, pad, 'bf16', dst_channels=Cout + 48, dst_coff=16, tile=tile)
    np.testing.assert_array_equal(got2[..., 16:16 + Cout], got)
    assert (got2[..., :16] == -7).all() and (got2[..., 16 + Cout:] == -7).all()


def test_conv_patch_random_sweep():
    """Random stride-1 layer shapes, source / destination channel slices and batch sizes: every eligible
    patch-resident variant gives the bits of the im2col kernel (same operands, same k order)."""
    rng = np.random.default_rng(2024)
    n_ok = 0
    for case in range(40):
        kh, kw = [(1, 1), (3, 3), (5, 5), (1, 7), (7, 1), (1, 3), (3, 1), (3, 5)][rng.integers(8)]
        Cin = int(rng.choice([32, 48, 64, 80, 96, 112, 128, 160, 192]))
        Cout = int(rng.choice([16, 32, 48, 64, 80, 96, 128, 192, 208]))
        H, W, B = int(rng.integers(max(kh, 2), 40)), int(rng.integers(max(kw, 2), 40)), int(rng.integers(1, 9))
        pad = ['SAME', 'VALID'][rng.integers(2)]
        Ho, pt, _ = cnn_ref.out_size(H, kh, 1, pad)
        Wo, pl, _ = cnn_ref.out_size(W, kw, 1, pad)
        xc = Cin + int(rng.choice([0, 8, 64])); xo = int(rng.choice([0, 8])) if xc > Cin else 0
        yc = Cout + int(rng.choice([0, 16, 48])); yo = int(rng.choice([0, 4, 16])) if yc >= Cout + 16 else 0
        relu = int(rng.integers(2))
        x = torch.randn(B, H, W, xc, device=DEV).to(torch.bfloat16)
        K = kh * kw * Cin
        wf = torch.randn(Cout, (K + 63) // 64 * 64, device=DEV) / K ** 0.5
        wf[:, K:] = 0
        w = wf.to(torch.bfloat16).contiguous()
        scale, shift = torch.rand(Cout, device=DEV) + 0.5, torch.randn(Cout, device=DEV) * 0.1
        wt = L.ConvWeight(w.data_ptr(), scale.data_ptr(), shift.data_ptr())
        ref = None
        for tile in [3] + list(range(13, 26)) + list(range(48, 54)):
            y = torch.full((B, Ho, Wo, yc), -7.0, dtype=torch.bfloat16, device=DEV)
            op = L.CnnOp(kind=0, src=0, dst=1, src_coff=xo, dst_coff=yo, H=H, W=W, Cin=Cin, Cout=Cout, KH=kh, KW=kw, SH=1,
                         SW=1, PT=pt, PL=pl, Ho=Ho, Wo=Wo, weight=0, relu=relu, out_f32=0, tile=tile)
            rc = lib().comic_conv2d_bn_relu(C.byref(op), x.data_ptr(), xc, y.data_ptr(), yc, C.byref(wt), B, 1, stream())
            sync()
            if rc != 0:
                assert tile != 3 and b'not eligible' in lib().comic_last_error()
                continue
            if tile == 3:
                ref = y
            else:
                assert torch.equal(y, ref), (case, tile, B, H, W, Cin, Cout, kh, kw, pad, xc, xo, yc, yo)
                n_ok += 1
    assert n_ok > 300


IMG_CASES = [   # (H, W, Cin, Cout, taps): every shape csrc/conv_img.hip is instantiated for
    (12, 12, 128, 192, (7, 1)), (12, 12, 128, 192, (1, 7)), (12, 12, 160, 192, (7, 1)), (12, 12, 192, 192, (1, 7)),
    (12, 12, 192, 192, (7, 1)), (12, 12, 128, 128, (1, 7)), (12, 12, 128, 128, (7, 1)), (12, 12, 160, 160, (1, 7)),
    (12, 12, 160, 160, (7, 1)), (25, 25, 64, 96, (3, 3)), (25, 25, 96, 96, (3, 3)), (5, 5, 448, 384, (3, 3)),
    (5, 5, 384, 384, (1, 3)), (5, 5, 384, 384, (3, 1))]


def _frag_weights(w, Cout, Kpad):
    """comic_cnn_pack_frag_weights on one [Cout][Kpad] bf16 record."""
    out = torch.zeros_like(w)
    table = torch.tensor([[0, Cout, Kpad]], dtype=torch.int64, device=DEV)
    L.check(lib().comic_cnn_pack_frag_weights(w.data_ptr(), out.data_ptr(), table.data_ptr(), 1, w.numel(), stream()), 'pack')
    return out


@pytest.mark.parametrize('case', IMG_CASES)
def test_conv_image_resident_identical_bits(case):
    """The image-resident kernel (tile id 55: whole images in the LDS without halo, out-of-image taps read a zero page,
    weights streamed in fragment order) against the im2col kernel: same operands, same k order per accumulator ->
    identical bits.  Batch sizes that leave the last workgroup with fewer images than it has room for, source and
    destination channel slices, with and without ReLU; one launch with two members (a grouped launch's form) too."""
    H, W, Cin, Cout, (kh, kw) = case
    rng = np.random.default_rng(H * 131 + Cin + Cout + 7 * kh + kw)
    K = kh * kw * Cin
    Kpad = (K + 63) // 64 * 64
    pt, pl = (kh - 1) // 2, (kw - 1) // 2
    for B, xc, xo, yc, yo, relu in ((1, Cin, 0, Cout, 0, 1), (5, Cin + 64, 8, Cout + 48, 16, 1), (13, Cin, 0, Cout + 16, 4, 0)):
        x = torch.randn(B, H, W, xc, device=DEV).to(torch.bfloat16)
        wf = torch.randn(Cout, Kpad, device=DEV) / K ** 0.5
        wf[:, K:] = 0
        w = wf.to(torch.bfloat16).contiguous()
        frag = _frag_weights(w, Cout, Kpad)
        # the packing itself: lane l of step s of tile t holds W[16 t + (l & 15)][32 s + 8 (l >> 4) .. + 8]
        f = frag.view(Cout // 16, Kpad // 32, 64, 8)
        for t, s_, l in ((0, 0, 0), (Cout // 16 - 1, Kpad // 32 - 1, 63), (1, 3, 37)):
            assert torch.equal(f[t, s_, l], w[16 * t + (l & 15), 32 * s_ + 8 * (l >> 4):32 * s_ + 8 * (l >> 4) + 8])
        scale, shift = torch.rand(Cout, device=DEV) + 0.5, torch.randn(Cout, device=DEV) * 0.1
        wt = L.ConvWeight(w.data_ptr(), scale.data_ptr(), shift.data_ptr(), frag.data_ptr())
        ys = {}
        for tile in (3, L.IMG_TILE):
            y = torch.full((B, H, W, yc), -7.0, dtype=torch.bfloat16, device=DEV)
            op = L.CnnOp(kind=0, src=0, dst=1, src_coff=xo, dst_coff=yo, H=H, W=W, Cin=Cin, Cout=Cout, KH=kh, KW=kw, SH=1,
                         SW=1, PT=pt, PL=pl, Ho=H, Wo=W, weight=0, relu=relu, out_f32=0, tile=tile)
            L.check(lib().comic_conv2d_bn_relu(C.byref(op), x.data_ptr(), xc, y.data_ptr(), yc, C.byref(wt), B, 1, stream()),
                    'conv tile %d' % tile)
            sync()
            ys[tile] = y
        assert torch.equal(ys[L.IMG_TILE], ys[3]), (case, B)
        assert bool((ys[3][..., yo:yo + Cout].float().abs().max() > 0.1))
    # without fragment-order weights the id is refused, like a patch id on an ineligible layer
    wt0 = L.ConvWeight(w.data_ptr(), scale.data_ptr(), shift.data_ptr())
    op = L.CnnOp(kind=0, src=0, dst=1, H=H, W=W, Cin=Cin, Cout=Cout, KH=kh, KW=kw, SH=1, SW=1, PT=pt, PL=pl, Ho=H, Wo=W,
                 weight=0, relu=1, tile=L.IMG_TILE)
    assert lib().comic_conv2d_bn_relu(C.byref(op), x.data_ptr(), xc, y.data_ptr(), yc, C.byref(wt0), B, 1, stream()) != 0
    assert b'not eligible' in lib().comic_last_error()


def test_conv_patch_rejects_strided():
    x = np.zeros((1, 9, 9, 32), np.float32)
    w = np.zeros((3, 3, 32, 32), np.float32)
    z, o = np.zeros(32, np.float32), np.ones(32, np.float32)
    with pytest.raises(L.ComicHipError):
        _run_conv(x, w, z, z, o, 2, 'VALID', 'bf16', tile=13)


@pytest.mark.parametrize('dtype', ['f32', 'bf16'])
def test_stem_conv(dtype):
    rng = np.random.default_rng(0)
    x = rng.uniform(-1, 1, (2, 37, 41, 3)).astype(np.float32)
    w = (rng.standard_normal((3, 3, 3, 32)) / 5).astype(np.float32)
    beta = 0.1 * rng.standard_normal(32).astype(np.float32)
    mean = 0.1 * rng.standard_normal(32).astype(np.float32)
    var = rng.uniform(0.5, 1.5, 32).astype(np.float32)
    # the stem reads fp32 images and fp32 weights in both modes; only the store is bf16
    ref = cnn_ref.batch_norm_inference(cnn_ref.conv2d(x, w, 2, 'VALID'), beta, mean, var)
    ref = np.maximum(ref, 0)
    got = _run_conv(x, w, beta, mean, var, 2, 'VALID', dtype)
    assert_close(got, cnn_ref.bf16_round(ref) if dtype == 'bf16' else ref, 1e-4 if dtype == 'f32' else 5e-3, 'stem')


def _run_pool(x, kind, k, s, pad, dtype, dst_channels=None, dst_coff=0):
    B, H, W, Cc = x.shape
    kh, kw = (k, k) if isinstance(k, int) else k
    Ho, pt, _ = cnn_ref.out_size(H, kh, s, pad)
    Wo, pl, _ = cnn_ref.out_size(W, kw, s, pad)
    tdt = torch.bfloat16 if dtype == 'bf16' else torch.float32
    xd = dev(x).to(tdt)
    yc = dst_channels or Cc
    y = torch.full((B, Ho, Wo, yc), -7.0, dtype=torch.float32 if kind == 4 else tdt, device=DEV)
    op = L.CnnOp(kind=kind, src=0, dst=1, src_coff=0, dst_coff=dst_coff, H=H, W=W, Cin=Cc, Cout=Cc, KH=kh, KW=kw,
                 SH=s, SW=s, PT=pt, PL=pl, Ho=Ho, Wo=Wo, weight=-1, relu=0, out_f32=int(kind == 4))
    L.check(lib().comic_conv2d_bn_relu(C.byref(op), xd.data_ptr(), Cc, y.data_ptr(), yc, None, B,
                                       1 if dtype == 'bf16' else 0, stream()), 'pool')
    sync()
    return y.float().cpu().numpy()


def test_device_image_preprocess_matches_numpy_pipeline():
    """comic_image_preprocess (uint8 -> [0,1] -> TF-1 bilinear 256x256 -> flip -> crop -> [-1,1]) against the ORACLE's
    restatement of inception_preprocessing_radix.py:158-278 (oracle/preprocess_ref.py), bit for bit: odd sizes, up- and
    down-scaling, flips, crop offsets; staging slots re-used across calls; and the product's own host path against it."""
    from comic_amd import inputs
    from oracle import preprocess_ref
    rng = np.random.default_rng(5)
    pre = inputs.DevicePreprocessor(DEV, 224, 224)
    for rep in range(3):
        sizes = [(480, 640), (333, 500), (100, 77), (256, 256), (1, 1), (257, 1023)][:6 - rep]
        ims = [rng.integers(0, 256, (h, w, 3), dtype=np.uint8) for h, w in sizes]
        params = [(bool(rng.integers(2)), int(rng.integers(0, 33)), int(rng.integers(0, 33))) for _ in ims]
        got = pre(ims, params).cpu().numpy()
        for i, (im, prm) in enumerate(zip(ims, params)):
            want = preprocess_ref.preprocess_image(im, 224, 224, *prm)
            np.testing.assert_array_equal(got[i], want, err_msg='image %d %s %s' % (i, im.shape, prm))
            np.testing.assert_array_equal(inputs.preprocess_image(im, 224, 224, True, None, prm), want)
    pre299 = inputs.DevicePreprocessor(DEV, 256, 256)          # no crop margin
    im = rng.integers(0, 256, (300, 300, 3), dtype=np.uint8)
    np.testing.assert_array_equal(pre299([im], [(True, 0, 0)]).cpu().numpy()[0],
                                  preprocess_ref.preprocess_image(im, 256, 256, True, 0, 0))


@pytest.mark.parametrize('dtype', ['f32', 'bf16'])
def test_pools(dtype):
    rng = np.random.default_rng(1)
    x = rng.standard_normal((3, 13, 11, 64)).astype(np.float32)
    if dtype == 'bf16':
        x = cnn_ref.bf16_round(x)
    q = cnn_ref.bf16_round if dtype == 'bf16' else (lambda a: a)
    np.testing.assert_array_equal(_run_pool(x, 2, 3, 2, 'VALID', dtype), cnn_ref.max_pool(x, 3, 2, 'VALID'))
    got = _run_pool(x, 3, 3, 1, 'SAME', dtype)
    assert_close(got, q(cnn_ref.avg_pool(x, 3, 1, 'SAME')), 1e-6 if dtype == 'f32' else 5e-3, 'avgpool')
    got = _run_pool(x, 2, 3, 2, 'VALID', dtype, dst_channels=160, dst_coff=96)
    np.testing.assert_array_equal(got[..., 96:], cnn_ref.max_pool(x, 3, 2, 'VALID'))
    x5 = x[:, :5, :5, :]
    got = _run_pool(x5, 4, 5, 1, 'VALID', dtype)
    assert_close(got, cnn_ref.avg_pool(x5, 5, 1, 'VALID'), 1e-5, 'global avgpool')
    got = _run_pool(x[:, :9, :9, :], 4, 8, 1, 'VALID', dtype)        # 9x9 map, 8x8 window -> 2x2
    assert_close(got, cnn_ref.avg_pool(x[:, :9, :9, :], 8, 1, 'VALID'), 1e-5, 'global avgpool 8x8')


# ------------------------------------------------------------------------ decoder kernels --
def test_embed_fwd_bwd():
    rng = np.random.default_rng(2)
    V, E, rows = 258, 256, 150
    table = rng.standard_normal((V, E)).astype(np.float32)
    ids = rng.integers(-1, V, rows).astype(np.int32)
    out = torch.empty((rows, E), device=DEV)
    L.check(lib().comic_embed_fwd(P(table), P(ids), out.data_ptr(), rows, E, V, stream()))
    np.testing.assert_array_equal(out.cpu().numpy(), dr.embed(table, ids))
    dout = rng.standard_normal((rows, E)).astype(np.float32)
    dt = torch.zeros((V, E), device=DEV)
    L.check(lib().comic_embed_bwd(P(ids), P(dout), dt.data_ptr(), rows, E, V, stream()))
    ref = np.zeros((V, E), np.float64)
    np.add.at(ref, ids[ids >= 0], dout[ids >= 0])
    assert_close(dt.cpu().numpy(), ref, 1e-6, 'embed_bwd')


@pytest.mark.parametrize('V,E,rows,hot', [(258, 256, 6048, 4800),      # a 224-hypothesis SCST step of long captions: radix digit 0 on 80 %
                                          (258, 256, 9000, 9000),      # one id on every row: the LDS list (4096 entries) runs full twice
                                          (40, 100, 5000, 4500),       # E % 64 != 0: a partly used column slice
                                          (3000, 64, 2000, 1500),      # many vocabulary rows: still the 1024-thread form (V * slices <= 4096)
                                          (25599, 256, 1280, 300),     # word vocabulary: the one-wave-per-stream form
                                          (300, 30, 5000, 4700)])      # E % 4 != 0: scalar columns
def test_embed_bwd_skewed_ids_and_list_overflow(V, E, rows, hot):
    """comic_embed_bwd on frequency-skewed ids (the reference vocabulary is frequency-sorted, prepro_base.py:186: one radix digit
    sits on most tokens): `hot` rows carry id 1 -- more than the kernel's LDS list holds in two of the cases, so the
    flush-and-refill path runs -- the rest are uniform with PAD (-1) among them.  Accumulating form (+=) against a float64
    scatter-add; two runs give the same bits."""
    rng = np.random.default_rng(V + rows)
    ids = rng.integers(-1, V, rows).astype(np.int32)
    ids[rng.permutation(rows)[:hot]] = 1
    dout = rng.standard_normal((rows, E)).astype(np.float32)
    base = rng.standard_normal((V, E)).astype(np.float32)
    d_ids, d_dout = dev(ids), dev(dout)
    runs = []
    for _ in range(2):
        dt = dev(base)
        L.check(lib().comic_embed_bwd(d_ids.data_ptr(), d_dout.data_ptr(), dt.data_ptr(), rows, E, V, stream()))
        runs.append(dt.cpu().numpy())
    ref = base.astype(np.float64)
    np.add.at(ref, ids[ids >= 0], dout[ids >= 0].astype(np.float64))
    assert_close(runs[0], ref, 2e-6, 'embed_bwd skewed')
    np.testing.assert_array_equal(runs[0], runs[1])
    untouched = np.setdiff1d(np.arange(V), ids[ids >= 0])
    np.testing.assert_array_equal(runs[0][untouched], base[untouched])


def test_dropout_mask_and_apply():
    n = 1 << 20
    m = torch.empty(n, device=DEV)
    L.check(lib().comic_dropout_mask(m.data_ptr(), n, 0.65, 1234, 0, stream()))
    m2 = torch.empty(n, device=DEV)
    L.check(lib().comic_dropout_mask(m2.data_ptr(), n, 0.65, 1234, 0, stream()))
    m3 = torch.empty(n, device=DEV)
    L.check(lib().comic_dropout_mask(m3.data_ptr(), n, 0.65, 1235, 0, stream()))
    mm = m.cpu().numpy()
    assert set(np.unique(mm)) <= {0.0, 1.0}
    assert abs(mm.mean() - 0.65) < 3e-3
    assert torch.equal(m, m2) and not torch.equal(m, m3)
    x = torch.randn(n, device=DEV)
    y = torch.empty_like(x)
    L.check(lib().comic_dropout_apply(x.data_ptr(), m.data_ptr(), 0.65, y.data_ptr(), n, stream()))
    np.testing.assert_array_equal(y.cpu().numpy(), (x.cpu().numpy() / np.float32(0.65)) * mm)


def test_dropout_masks4_equals_four_calls():
    """comic_dropout_masks4_dev (the four masks of a training step in one launch, keep probability per segment) gives
    the bits of four comic_dropout_mask calls with cumulative counter offsets."""
    n4 = [1000, 70001, 33333, 257]
    keep4 = [0.65, 0.65, 0.5, 0.9]
    seed = 0x9E3779B9 + 5
    buf = torch.empty(sum(n4), device=DEV)
    seed_dev = torch.tensor([seed], dtype=torch.int64, device=DEV)
    L.check(lib().comic_dropout_masks4_dev(buf.data_ptr(), (C.c_int64 * 4)(*n4), (C.c_float * 4)(*keep4),
                                           seed_dev.data_ptr(), stream()))
    off = 0
    for n, keep in zip(n4, keep4):
        m = torch.empty(n, device=DEV)
        L.check(lib().comic_dropout_mask(m.data_ptr(), n, keep, seed, off, stream()))
        assert torch.equal(m, buf[off:off + n]), (n, keep)
        off += n


def test_lstm_gates_fwd_bwd():
    rng = np.random.default_rng(3)
    B, D = 6, 128
    g = rng.standard_normal((B, 4 * D)).astype(np.float32)
    c = rng.standard_normal((B, D)).astype(np.float32)
    h = rng.standard_normal((B, D)).astype(np.float32)
    mask = (rng.random((B, D)) < 0.65).astype(np.float32)
    lens = np.array([3, 1, 5, 2, 9, 2], np.int32)
    t = 2
    fin = t >= lens
    i, j, f, o = g[:, :D], g[:, D:2 * D], g[:, 2 * D:3 * D], g[:, 3 * D:]
    si, tj, sf, so = dr.sigmoid(i), np.tanh(j), dr.sigmoid(f + 1), dr.sigmoid(o)
    c2 = c * sf + si * tj
    tc = np.tanh(c2)
    h2 = tc * so
    y = h2 / np.float32(0.65) * mask
    outs = {k: torch.empty((B, D), device=DEV) for k in ('c_new', 'h_new', 'y', 'cs', 'hs')}
    ga = torch.empty((B, 4 * D), device=DEV)
    L.check(lib().comic_lstm_gates_fwd(P(g), P(c), P(h), ga.data_ptr(),
                                       outs['c_new'].data_ptr(), outs['h_new'].data_ptr(), outs['y'].data_ptr(),
                                       P(mask), 0.65, P(lens), t, outs['cs'].data_ptr(),
                                       outs['hs'].data_ptr(), B, D, stream()))
    assert_close(outs['c_new'].cpu().numpy(), c2, 1e-5, 'c_new')
    assert_close(outs['y'].cpu().numpy(), y, 1e-5, 'y')
    assert_close(outs['cs'].cpu().numpy(), np.where(fin[:, None], c, c2), 1e-5, 'c_state')
    assert_close(outs['hs'].cpu().numpy(), np.where(fin[:, None], h, h2), 1e-5, 'h_state')
    assert_close(ga.cpu().numpy(), np.concatenate([si, tj, sf, so], 1), 1e-5, 'gates')
    # backward vs the oracle's cell backward
    dc = rng.standard_normal((B, D)).astype(np.float32)
    dh = rng.standard_normal((B, D)).astype(np.float32)
    dy = rng.standard_normal((B, D)).astype(np.float32)
    live = (~fin)[:, None].astype(np.float32)
    dh2 = dh * live + dy / np.float32(0.65) * mask
    dg_ref, dcp_ref = dr._lstm_backward(None, (si, tj, sf, so, tc), c, dc * live, dh2, D)
    d_dc, d_dh = dev(dc), dev(dh)
    dg = torch.empty((B, 4 * D), device=DEV)
    L.check(lib().comic_lstm_gates_bwd(ga.data_ptr(), P(c), outs['c_new'].data_ptr(), P(dy),
                                       P(mask), 0.65, P(lens), t, d_dc.data_ptr(),
                                       d_dh.data_ptr(), dg.data_ptr(), B, D, stream()))
    assert_close(dg.cpu().numpy(), dg_ref, 1e-5, 'dg')
    assert_close(d_dc.cpu().numpy(), dc * (1 - live) + dcp_ref, 1e-5, 'dc')
    assert_close(d_dh.cpu().numpy(), dh * (1 - live), 1e-6, 'dh')


ATTN_CASES = [
    # B, M, D, H, Cv, method, prob, tied
    (5, 25, 512, 8, 512, 'add_LN', 'softmax', True),
    (3, 64, 512, 8, 512, 'add_LN', 'softmax', False),
    (3, 196, 512, 8, 512, 'add_LN', 'sigmoid', True),
    (4, 25, 512, 1, 2048, 'add_LN', 'softmax', False),
    (4, 9, 128, 4, 128, 'dot', 'softmax', True),
    (2, 70, 64, 2, 192, 'dot', 'sigmoid', False),
    (3, 25, 1024, 16, 1024, 'add_LN', 'softmax', True)]


@pytest.mark.parametrize('case', ATTN_CASES)
@pytest.mark.parametrize('use_mask', [False, True])
def test_attn_step_fwd_bwd(case, use_mask):
    B, M, D, H, Cv, method, prob, tied = case
    rng = np.random.default_rng(B * M + D)
    cfg = dr.DecoderConfig(rnn_size=D, attn_num_heads=H, attn_alignment_method=method, attn_probability_fn=prob,
                           cnn_fm_projection='tied' if tied else 'independent')
    keys = rng.standard_normal((B, M, D)).astype(np.float32)
    values = keys if tied else rng.standard_normal((B, M, Cv)).astype(np.float32)
    q = rng.standard_normal((B, D)).astype(np.float32)
    p = dict(ln_g=(1 + 0.1 * rng.standard_normal(D)).astype(np.float32),
             ln_b=(0.1 * rng.standard_normal(D)).astype(np.float32),
             v=(rng.standard_normal(D) / math.sqrt(D / H) * 3).astype(np.float32), tau=np.array(2.5, np.float32))
    keep = 0.9
    mask = (rng.random((B, H, M)) < keep).astype(np.float32) if use_mask else None
    alpha_ref, ac = dr.attention_scores(p, cfg, keys, q)
    alpha_d_ref = dr.dropout(alpha_ref, mask, keep)
    ctx_ref = dr.context(cfg, alpha_d_ref, values)
    d = L.AttnDesc(B=B, M=M, D=D, H=H, Cv=Cv, method=0 if method == 'add_LN' else 1,
                   prob=0 if prob == 'softmax' else 1, tied=int(tied))
    dk = dev(keys)
    dvv = dk if tied else dev(values)
    dq_in = dev(q)
    dp = {k: dev(v.reshape(-1)) for k, v in p.items()}
    alpha = torch.empty((B, H, M), device=DEV); alpha_d = torch.empty((B, H, M), device=DEV)
    ctx = torch.empty((B, Cv), device=DEV)
    dmask = dev(mask) if use_mask else None
    L.check(lib().comic_attn_step_fwd(C.byref(d), dk.data_ptr(), dvv.data_ptr(), dq_in.data_ptr(),
                                      dp['ln_g'].data_ptr(), dp['ln_b'].data_ptr(), dp['v'].data_ptr(),
                                      dp['tau'].data_ptr(), L.ptr(dmask), keep, alpha.data_ptr(), alpha_d.data_ptr(),
                                      ctx.data_ptr(), stream()), 'attn_fwd')
    sync()
    assert_close(alpha.cpu().numpy(), alpha_ref, 1e-4, 'alpha')
    assert_close(alpha_d.cpu().numpy(), alpha_d_ref, 1e-4, 'alpha_d')
    assert_close(ctx.cpu().numpy(), ctx_ref, 1e-4, 'ctx')
    # ---- backward vs the oracle's analytic backward ----
    dctx = rng.standard_normal((B, Cv)).astype(np.float32)
    dmap = (0.01 * rng.standard_normal((B, M))).astype(np.float32)
    vs = values.reshape(B, M, H, Cv // H)
    dalpha_d = np.einsum('bhd,bmhd->bhm', dctx.reshape(B, H, Cv // H), vs) + dmap[:, None, :]
    dvalues_ref = np.einsum('bhm,bhd->bmhd', alpha_d_ref, dctx.reshape(B, H, Cv // H)).reshape(B, M, Cv)
    dalpha = dalpha_d / np.float32(keep) * mask if use_mask else dalpha_d
    grads = {k: np.zeros_like(v) for k, v in p.items()}
    dk_ref, dq_ref = dr._attention_backward(p, cfg, keys, q, ac, alpha_ref, dalpha, grads)
    dq = torch.empty((B, D), device=DEV)
    dkeys = torch.zeros((B, M, D), device=DEV)
    dvalues = dkeys if tied else torch.zeros((B, M, Cv), device=DEV)
    pgrad = torch.zeros((B, 3 * D + 1), device=DEV)
    L.check(lib().comic_attn_step_bwd(C.byref(d), dk.data_ptr(), dvv.data_ptr(), dq_in.data_ptr(),
                                      dp['ln_g'].data_ptr(), dp['ln_b'].data_ptr(), dp['v'].data_ptr(),
                                      dp['tau'].data_ptr(), alpha.data_ptr(), L.ptr(dmask), keep,
                                      P(dctx), P(dmap), dq.data_ptr(), dkeys.data_ptr(),
                                      dvalues.data_ptr(), pgrad.data_ptr(), stream()), 'attn_bwd')
    sync()
    assert_close(dq.cpu().numpy(), dq_ref, F32_RTOL, 'dq')
    if tied:
        assert_close(dkeys.cpu().numpy(), dk_ref + dvalues_ref, F32_RTOL, 'dkeys(tied)')
    else:
        assert_close(dkeys.cpu().numpy(), dk_ref, F32_RTOL, 'dkeys')
        assert_close(dvalues.cpu().numpy(), dvalues_ref, F32_RTOL, 'dvalues')
    if method == 'add_LN':
        pg = pgrad.cpu().numpy().sum(0)
        assert_close(pg[:D], grads['v'], F32_RTOL, 'dv')
        assert_close(pg[D:2 * D], grads['ln_g'], F32_RTOL, 'dln_g')
        assert_close(pg[2 * D:3 * D], grads['ln_b'], F32_RTOL, 'dln_b')
        assert abs(pg[3 * D] - grads['tau']) <= F32_RTOL * max(1e-3, abs(float(grads['tau']))) + 1e-6, 'dtau'


@pytest.mark.parametrize('V', [258, 25599])
def test_xent_fwd_bwd(V):
    rng = np.random.default_rng(V)
    T, B = 7, 5
    logits = (3 * rng.standard_normal((T, B, V))).astype(np.float32)
    lens = np.array([7, 3, 5, 1, 6], np.int32)
    targets = rng.integers(0, V, (B, T)).astype(np.int32)
    wmask = (np.arange(T)[None, :] < lens[:, None]).astype(np.float32)
    coef = (wmask / wmask.sum()).astype(np.float32)
    dl = dev(logits)
    loss_rows = torch.empty(T * B, device=DEV); dlog = torch.empty((T, B, V), device=DEV)
    ids = torch.empty((T, B), dtype=torch.int32, device=DEV)
    L.check(lib().comic_xent_fwd_bwd(dl.data_ptr(), P(targets), P(coef),
                                     P(wmask), P(lens), loss_rows.data_ptr(),
                                     dlog.data_ptr(), ids.data_ptr(), T, B, V, stream()))
    live = (np.arange(T)[:, None] < lens[None, :])
    lg = logits * live[..., None]
    lsm = dr.log_softmax(lg.astype(np.float64), -1)
    xent = -np.take_along_axis(lsm, targets.T[..., None], 2)[..., 0] * wmask.T
    np.testing.assert_array_equal(dl.cpu().numpy(), lg)                      # finished rows zeroed
    assert_close(loss_rows.cpu().numpy().reshape(T, B), xent, 1e-5, 'xent rows')
    oh = np.zeros_like(lsm); np.put_along_axis(oh, targets.T[..., None], 1.0, 2)
    assert_close(dlog.cpu().numpy(), (np.exp(lsm) - oh) * coef.T[..., None], 1e-5, 'dlogits')
    np.testing.assert_array_equal(ids.cpu().numpy(), lg.argmax(-1))


def test_argmax_ties_lowest_index():
    x = np.zeros((3, 300), np.float32)
    x[0, [7, 100, 299]] = 5; x[1, :] = -1; x[2, 299] = 1
    idx = torch.empty(3, dtype=torch.int32, device=DEV)
    L.check(lib().comic_argmax_rows(P(x), idx.data_ptr(), 3, 300, stream()))
    assert idx.cpu().tolist() == [7, 0, 299]


def test_beam_step_vs_oracle_step():
    rng = np.random.default_rng(5)
    B, W, V, end = 4, 3, 258, 257
    logits = (2 * rng.standard_normal((B, W, V))).astype(np.float32)
    logits[0, 0, 5] = logits[0, 0, 9] = 40.0            # exact tie -> lower flat index first
    lp = np.array([[0, -1.5, -2.0]] * B, np.float32); lp[1] = [0, -np.inf, -np.inf]
    fin = np.zeros((B, W), np.int32); fin[2, 1] = 1; fin[3] = 1
    lens = rng.integers(0, 5, (B, W)).astype(np.int64)
    step = dr.log_softmax(logits, -1)
    row = np.full(V, np.finfo(np.float32).min, np.float32); row[end] = 0
    step = np.where(fin[:, :, None].astype(bool), row[None, None, :], step)
    total = (lp[:, :, None] + step).reshape(B, W * V)
    order = np.argsort(-total, axis=1, kind='stable')[:, :W]
    d_lp, d_fin, d_len = dev(lp), dev(fin), dev(lens)
    word = torch.empty((B, W), dtype=torch.int32, device=DEV); par = torch.empty_like(word)
    sc = torch.empty((B, W), device=DEV)
    L.check(lib().comic_beam_step(P(logits), d_lp.data_ptr(), d_fin.data_ptr(), d_len.data_ptr(),
                                  word.data_ptr(), par.data_ptr(), sc.data_ptr(), B, W, V, end, stream()))
    np.testing.assert_array_equal(word.cpu().numpy(), order % V)
    np.testing.assert_array_equal(par.cpu().numpy(), order // V)
    ref_sc = np.take_along_axis(total, order, 1)
    assert_close(sc.cpu().numpy(), ref_sc, 1e-6, 'beam scores')
    bidx = np.arange(B)[:, None]
    pf = fin[bidx, order // V].astype(bool)
    np.testing.assert_array_equal(d_fin.cpu().numpy().astype(bool), pf | (order % V == end))
    np.testing.assert_array_equal(d_len.cpu().numpy(), lens[bidx, order // V] + (~pf))
    assert word.cpu().numpy()[0, 0] == 5 and word.cpu().numpy()[0, 1] == 9


@pytest.mark.parametrize('B,W,V,D', [(4, 3, 258, 128), (5, 8, 9000, 128), (6, 3, 8962, 256), (3, 5, 1000, 64), (4, 3, 2500, 96)])
def test_beam_step_dense_vs_oracle_step(B, W, V, D):
    """One beam step from the decoder outputs through the kernels the decode loop uses for the shape (streaming projection
    + chunk top-k + merge; register-resident small step; GEMM + comic_beam_step) against the [TF-1.9] step restated in
    numpy: exact ties between duplicate vocabulary columns (lower flat index first, also across chunks and beams), a
    finished beam, an entry whose other beams are at -inf, an entry with every beam finished."""
    rng = np.random.default_rng(B + W + V)
    end = V - 1
    y = rng.standard_normal((B, W, D)).astype(np.float32)
    Wo = (rng.standard_normal((D, V)) / np.sqrt(D)).astype(np.float32)
    bo = (0.1 * rng.standard_normal(V)).astype(np.float32)
    Wo[:, 7] = Wo[:, 3]; bo[7] = bo[3]                          # duplicate columns: exact ties, same chunk
    Wo[:, V - 5] = Wo[:, 3]; bo[V - 5] = bo[3]                  # ... and in the last chunk
    Wo[:, 3] *= 3.0; Wo[:, 7] *= 3.0; Wo[:, V - 5] *= 3.0       # make them likely winners for some rows
    y[0, 1] = y[0, 0]                                           # two beams of entry 0 with identical logits
    lp = np.tile(np.linspace(0, -2.0, W, dtype=np.float32), (B, 1))
    lp[0, 1] = lp[0, 0]                                         # ... and equal totals: ties across beams
    lp[1, 1:] = -np.inf
    fin = np.zeros((B, W), np.int32); fin[2, 1] = 1; fin[B - 1] = 1
    lens = rng.integers(0, 5, (B, W)).astype(np.int64)
    logits = (y.reshape(B * W, D).astype(np.float64) @ Wo.astype(np.float64) + bo).astype(np.float32).reshape(B, W, V)
    step = dr.log_softmax(logits, -1)
    row = np.full(V, np.finfo(np.float32).min, np.float32); row[end] = 0
    step = np.where(fin[:, :, None].astype(bool), row[None, None, :], step)
    total = (lp[:, :, None] + step).reshape(B, W * V)
    order = np.argsort(-total, axis=1, kind='stable')[:, :W]
    d_lp, d_fin, d_len = dev(lp), dev(fin), dev(lens)
    word = torch.empty((B, W), dtype=torch.int32, device=DEV); par = torch.empty_like(word)
    sc = torch.empty((B, W), device=DEV)
    nb = int(lib().comic_beam_step_dense_workspace(B, W, D, V))
    ws = torch.empty(nb, dtype=torch.uint8, device=DEV)
    L.check(lib().comic_beam_step_dense(P(y), P(Wo), P(bo), d_lp.data_ptr(), d_fin.data_ptr(), d_len.data_ptr(),
                                        word.data_ptr(), par.data_ptr(), sc.data_ptr(), B, W, D, V, end, ws.data_ptr(), nb,
                                        stream()))
    sync()
    ref_sc = np.take_along_axis(total, order, 1)
    got_w, got_p, got_sc = word.cpu().numpy(), par.cpu().numpy(), sc.cpu().numpy()
    # device logits carry ~1e-5 of split-product error: where the oracle's neighbours are closer than that the order may
    # legitimately differ; everything else must agree exactly, and exact ties (duplicate columns, equal beams) must
    # resolve to the lower flat index
    fl = got_p.astype(np.int64) * V + got_w
    for b in range(B):
        for k in range(W):
            if fl[b, k] != order[b, k]:
                assert abs(total[b, fl[b, k]] - ref_sc[b, k]) <= 2e-4 * max(1.0, abs(ref_sc[b, k])), (b, k, fl[b, k], order[b, k])
        assert len(set(fl[b].tolist())) == W
    fin_sc = np.isfinite(ref_sc) & (np.abs(ref_sc) < 1e30)
    assert_close(np.where(fin_sc, got_sc, 0), np.where(fin_sc, ref_sc, 0), 2e-4, 'beam scores')
    dup = {3, 7, V - 5}
    for b in range(B):
        ws_b = [int(x) for x in got_w[b]]
        for k in range(W - 1):
            if ws_b[k] in dup and ws_b[k + 1] in dup and got_p[b, k] == got_p[b, k + 1] and got_sc[b, k] == got_sc[b, k + 1]:
                assert ws_b[k] < ws_b[k + 1], 'tie between duplicate columns must keep the lower index first'
    bidx = np.arange(B)[:, None]
    pf = fin[bidx, got_p].astype(bool)
    np.testing.assert_array_equal(d_fin.cpu().numpy().astype(bool), pf | (got_w == end))
    np.testing.assert_array_equal(d_len.cpu().numpy(), lens[bidx, got_p] + (~pf))
    # an entry with every beam finished keeps its beams (EOS at score + 0), in beam order
    np.testing.assert_array_equal(got_w[B - 1], np.full(W, end))
    np.testing.assert_array_equal(got_p[B - 1], np.argsort(-lp[B - 1], kind='stable'))


def test_gather_tree_matches_oracle():
    rng = np.random.default_rng(6)
    T, B, W, end = 9, 5, 4, 17
    step = rng.integers(0, 18, (T, B, W)).astype(np.int32)
    par = rng.integers(0, W, (T, B, W)).astype(np.int32)
    ml = np.array([9, 4, 0, 12, 1], np.int32)
    out = torch.empty((T, B, W), dtype=torch.int32, device=DEV)
    L.check(lib().comic_gather_tree(P(step), P(par), P(ml), out.data_ptr(),
                                    T, B, W, end, stream()))
    np.testing.assert_array_equal(out.cpu().numpy(), beam_ref.gather_tree(step, par, ml, end))


def test_adam_tf_matches_oracle():
    rng = np.random.default_rng(7)
    n = 100003
    w = rng.standard_normal(n).astype(np.float32); g = rng.standard_normal(n).astype(np.float32)
    m = (0.1 * rng.standard_normal(n)).astype(np.float32); v = (0.1 * rng.random(n)).astype(np.float32)
    dw, dg, dm, dv = dev(w), dev(g), dev(m), dev(v)
    t, lr, l2 = 3, 1e-2, 1e-5
    lr_t = lr * math.sqrt(1 - 0.999 ** t) / (1 - 0.9 ** t)
    L.check(lib().comic_adam_tf(dw.data_ptr(), dg.data_ptr(), dm.data_ptr(), dv.data_ptr(), n, lr_t, 0.9, 0.999,
                                1e-2, l2, 1.0, stream()))
    g_eff = g + np.float32(l2) * w
    dr.adam_tf_update(w, g_eff, m, v, t, lr, eps=1e-2)
    assert_close(dw.cpu().numpy(), w, 1e-5, 'adam w')
    assert_close(dm.cpu().numpy(), m, 1e-5, 'adam m')
    assert_close(dv.cpu().numpy(), v, 1e-5, 'adam v')


@pytest.mark.parametrize('opt_name', ['adam', 'sgd'])
def test_optimiser_update_by_ranges_equals_one_launch(opt_name):
    """The chunked gradient exchange (trainer.DataParallel.exchange_and_step) updates the flat buffer range by range, each
    range behind its own all-reduce, the status word's range first: the same bits as the one-launch update, `before_range`
    called once per range in order, and a voided step (status word set) skips every range."""
    from comic_amd import decoder as cdec, optim
    from comic_amd.trainer import DataParallel
    spec = cdec.DecoderSpec(M=25, C=2048, Cg=2048)
    outs = []
    for mode in ('flat', 'ranges', 'ranges_void'):
        g = torch.Generator(device='cpu').manual_seed(3)
        PP = cdec.FlatParams(spec.param_shapes(), DEV, status_tail=True)
        GG = cdec.FlatParams(spec.param_shapes(), DEV, status_tail=True)
        PP.data[:PP.numel].copy_(torch.randn(PP.numel, generator=g))
        GG.data[:GG.numel].copy_(torch.randn(GG.numel, generator=g))
        opt = optim.make_optimiser(opt_name, PP)
        seen = []
        for step in range(2):
            if mode == 'flat':
                opt.step(GG, 1e-2, grad_scale=0.5)
            else:
                if mode == 'ranges_void' and step == 1:
                    GG.data[GG.numel] = 1.0
                rng_ = DataParallel.chunk_bounds(GG, 4)[::-1]
                opt.step(GG, 1e-2, grad_scale=0.5, ranges=rng_, before_range=seen.append)
        sync()
        assert mode == 'flat' or seen == [0, 1, 2, 3] * 2
        outs.append((PP.data.clone(), opt.m.data.clone(), opt.v.data.clone(), opt.t))
    assert all(torch.equal(a, b) for a, b in zip(outs[0][:3], outs[1][:3])) and outs[0][3] == outs[1][3] == 2
    opt1 = outs[2]
    assert not torch.equal(opt1[0], outs[0][0])        # the voided second step left the parameters of step 1 ...
    PP = cdec.FlatParams(spec.param_shapes(), DEV, status_tail=True)
    GG = cdec.FlatParams(spec.param_shapes(), DEV, status_tail=True)
    g = torch.Generator(device='cpu').manual_seed(3)
    PP.data[:PP.numel].copy_(torch.randn(PP.numel, generator=g))
    GG.data[:GG.numel].copy_(torch.randn(GG.numel, generator=g))
    o1 = optim.make_optimiser(opt_name, PP)
    o1.step(GG, 1e-2, grad_scale=0.5)
    sync()
    assert torch.equal(opt1[0][:PP.numel], PP.data[:PP.numel])      # ... in every range


def test_gradient_clipping_matches_oracle():
    """clip_gradient_norm (model_base.py:394-401): tf.clip_by_norm per variable on g*gscale + l2*w, then the TF-Adam update;
    variables of one element, of less than a chunk and of several chunks, norms above and below the threshold, and the
    voided-step flag."""
    from comic_amd import decoder as cdec, optim
    rng = np.random.default_rng(11)
    shapes = {'a': (3, 5000), 'b': (700,), 'c': (), 'd': (40000,), 'e': (128, 64)}
    scale_of = {'a': 1.0, 'b': 1e-3, 'c': 50.0, 'd': 0.2, 'e': 1e-2}
    PP = cdec.FlatParams(shapes, DEV, status_tail=True)
    G = PP.like()
    w = {k: rng.standard_normal(shapes[k]).astype(np.float32) for k in shapes}
    g = {k: (scale_of[k] * rng.standard_normal(shapes[k])).astype(np.float32) for k in shapes}
    PP.load(w); G.load(g)
    clip, l2, gs, lr = 2.5, 1e-3, 0.5, 1e-2
    opt = optim.AdamTF(PP, l2_decay=l2, clip_norm=clip)
    opt.step(G, lr, grad_scale=gs)
    got = PP.to_numpy()
    n_clipped = 0
    for k in shapes:
        ge = np.float32(gs) * g[k] + np.float32(l2) * w[k]
        n_clipped += float(np.sqrt((ge.astype(np.float64) ** 2).sum())) > clip
        ge = dr.clip_by_norm(ge, clip)
        wk, m, v = w[k].copy(), np.zeros_like(w[k]), np.zeros_like(w[k])
        dr.adam_tf_update(wk, ge, m, v, 1, lr, eps=1e-2)
        assert_close(got[k], wk, 1e-5, 'clipped adam ' + k)
    assert 2 <= n_clipped < len(shapes)                      # both sides of the threshold are exercised
    # a voided step: neither the clip nor the update touches anything
    G.status.fill_(1.0)
    before, gbefore = PP.data.clone(), G.data.clone()
    opt.step(G, lr, grad_scale=gs)
    assert torch.equal(PP.data, before) and torch.equal(G.data, gbefore)


def test_momentum_tf_matches_oracle():
    """--optimiser sgd: tf.train.MomentumOptimizer(0.9, use_nesterov=False) (model_base.py:867-880), two updates."""
    rng = np.random.default_rng(8)
    n = 70001
    w = rng.standard_normal(n).astype(np.float32)
    acc = np.zeros(n, np.float32)
    dw, dacc = dev(w), dev(acc)
    lr, l2 = 1e-2, 1e-5
    for it in range(2):
        g = rng.standard_normal(n).astype(np.float32)
        L.check(lib().comic_momentum_tf(dw.data_ptr(), dev(g).data_ptr(), dacc.data_ptr(), n, lr, 0.9, l2, 0.5, stream()))
        dr.momentum_tf_update(w, np.float32(0.5) * g + np.float32(l2) * w, acc, lr)
    assert_close(dw.cpu().numpy(), w, 1e-6, 'momentum w')
    assert_close(dacc.cpu().numpy(), acc, 1e-6, 'momentum accum')


def test_legacy_encoder_head_matches_oracle():
    """--legacy image embedding (model_base.py:80-91): LN_tanh + linear(1024) forward and parameter gradients."""
    from comic_amd import encoder_head
    rng = np.random.default_rng(9)
    B, C_ = 6, 1024
    net = (rng.standard_normal((B, C_)) * 1.5 + 0.2).astype(np.float32)
    p = encoder_head.init_params(C_, seed=1)
    p['ln_gamma'] = rng.uniform(0.5, 1.5, C_).astype(np.float32)
    p['ln_beta'] = (0.1 * rng.standard_normal(C_)).astype(np.float32)
    head = encoder_head.LegacyEncoderHead(C_, p, DEV)
    out = head.forward(dev(net))
    ref, cache = dr.legacy_head_forward(p, net)
    assert_close(out.cpu().numpy(), ref, 1e-3, 'legacy im_embed')
    d = rng.standard_normal(ref.shape).astype(np.float32)
    g = head.backward(dev(d)).to_numpy()
    gref = dr.legacy_head_backward(p, cache, d)
    for k in gref:
        assert_close(g[k], gref[k], 1e-3, 'legacy d ' + k)
    assert set(head.export_params()) == {'Model/encoder/LN_tanh/beta', 'Model/encoder/LN_tanh/gamma',
                                         'Model/encoder/im_embed/weight'}


def test_process_decode_pool_equals_thread_decode(tmp_path):
    """inputs.DecodePool (spawned workers decoding JPEGs into shared-memory staging, registered as pinned memory) feeds
    the device preprocessing the same bytes as the in-process PIL decode: identical output tensors, blocks recycled
    over several batches, an image larger than its slot is refused with a clear error."""
    from PIL import Image
    from comic_amd import inputs
    rng = np.random.default_rng(4)
    paths = []
    for i, (h, w) in enumerate([(48, 64), (33, 50), (64, 64), (20, 31), (57, 40)]):
        p = str(tmp_path / ('%d.jpg' % i))
        Image.fromarray(rng.integers(0, 256, (h, w, 3), dtype=np.uint8)).save(p, quality=92)
        paths.append(p)
    params = [(bool(i & 1), i, 2 * i) for i in range(len(paths))]
    pre = inputs.DevicePreprocessor(DEV, 224, 224)
    want = pre([inputs.decode_image(p) for p in paths], params).cpu()
    pool = inputs.DecodePool(2, blocks=2, slot_bytes=64 * 64 * 3, max_batch=8)
    try:
        for _ in range(5):                                   # more batches than blocks: they are recycled
            got = pre.finish(pre.pack_paths(pool, paths, params)).cpu()
            assert torch.equal(got, want)
        big = str(tmp_path / 'big.jpg')
        Image.fromarray(rng.integers(0, 256, (80, 80, 3), dtype=np.uint8)).save(big)
        with pytest.raises(ValueError, match='loader slot'):
            pre.finish(pre.pack_paths(pool, [big], [(False, 0, 0)]))
    finally:
        pool.close()
