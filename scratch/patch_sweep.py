"""Randomised sweep: every patch-resident variant against the im2col kernel (bit-exact) on random layer shapes."""
import sys, ctypes as C
sys.path.insert(0, '.')
import numpy as np, torch
from comic_amd import _lib as L
lib = L.load(); dev = 'cuda:0'; st = L.stream_ptr()
rng = np.random.default_rng(int(sys.argv[1]) if len(sys.argv) > 1 else 0)
def out(size, k, pad):
    if pad == 'SAME': return size, (k - 1) // 2
    return size - k + 1, 0
n_ok = n_skip = 0
for case in range(int(sys.argv[2]) if len(sys.argv) > 2 else 150):
    kh, kw = [(1, 1), (3, 3), (5, 5), (1, 7), (7, 1), (1, 3), (3, 1), (3, 5)][rng.integers(8)]
    Cin = int(rng.choice([32, 48, 64, 80, 96, 112, 128, 160, 192])); Cout = int(rng.choice([16, 32, 48, 64, 80, 96, 128, 192, 208]))
    H = int(rng.integers(max(kh, 2), 40)); W = int(rng.integers(max(kw, 2), 40)); B = int(rng.integers(1, 9))
    pad = ['SAME', 'VALID'][rng.integers(2)]
    Ho, pt = out(H, kh, pad); Wo, pl = out(W, kw, pad)
    if Ho < 1 or Wo < 1: continue
    xc = Cin + int(rng.choice([0, 8, 64])); xo = int(rng.choice([0, 8])) if xc > Cin else 0
    yc = Cout + int(rng.choice([0, 16, 48])); yo = int(rng.choice([0, 4, 16])) if yc >= Cout + 16 else 0
    x = torch.randn(B, H, W, xc, device=dev).to(torch.bfloat16)
    K = kh * kw * Cin; Kpad = (K + 63) // 64 * 64
    wf = (torch.randn(Cout, Kpad, device=dev) / K ** 0.5); wf[:, K:] = 0
    w = wf.to(torch.bfloat16).contiguous()
    scale = torch.rand(Cout, device=dev) + 0.5; shift = torch.randn(Cout, device=dev) * 0.1
    wt = L.ConvWeight(w.data_ptr(), scale.data_ptr(), shift.data_ptr())
    res = {}
    for tile in [3] + list(range(13, 26)):
        y = torch.full((B, Ho, Wo, yc), -7.0, dtype=torch.bfloat16, device=dev)
        op = L.CnnOp(kind=0, src=0, dst=1, src_coff=xo, dst_coff=yo, H=H, W=W, Cin=Cin, Cout=Cout, KH=kh, KW=kw, SH=1, SW=1,
                     PT=pt, PL=pl, Ho=Ho, Wo=Wo, weight=0, relu=int(rng.integers(2)) if tile == 3 else res['relu'], out_f32=0, tile=tile)
        if tile == 3: res['relu'] = op.relu
        rc = lib.comic_conv2d_bn_relu(C.byref(op), x.data_ptr(), xc, y.data_ptr(), yc, C.byref(wt), B, 1, st)
        torch.cuda.synchronize()
        if rc != 0:
            assert tile != 3 and b'not eligible' in lib.comic_last_error(), lib.comic_last_error()
            n_skip += 1; continue
        if tile == 3: res['ref'] = y
        else:
            if not torch.equal(y, res['ref']):
                bad = (y != res['ref']).nonzero()
                print('MISMATCH case', case, dict(B=B, H=H, W=W, Cin=Cin, Cout=Cout, k=(kh, kw), pad=pad, xc=xc, xo=xo, yc=yc, yo=yo, tile=tile), 'n bad', len(bad), bad[:4].tolist())
                sys.exit(1)
            n_ok += 1
print('sweep ok: %d variant runs bit-identical to the im2col kernel, %d not eligible' % (n_ok, n_skip))
