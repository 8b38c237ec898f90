"""One conv layer (bf16) under several explicit tile ids: bit-compare every variant's output with the first id's
and time it.  CASES='B,H,W,Cin,Cout,kh,kw,stride,PAD;...'  TILES=1,44,54,...  REPS=20"""
import os, sys, ctypes as C
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), '..'))
import torch
from comic_amd import _lib as L
lib = L.load()
tiles = [int(t) for t in os.environ.get('TILES', '1,44,54,55,56,57').split(',')]
reps = int(os.environ.get('REPS', '20'))
dev = 'cuda:0'


def out(size, k, s, pad):
    if pad == 'SAME':
        o = -(-size // s); tot = max((o - 1) * s + k - size, 0); return o, tot // 2
    return (size - k) // s + 1, 0


for case in os.environ.get('CASES', '640,12,12,768,192,1,1,1,SAME;640,12,12,192,192,7,1,1,SAME').split(';'):
    f = case.split(',')
    B, H, W, Cin, Cout, kh, kw, s = [int(v) for v in f[:8]]
    pad = f[8] if len(f) > 8 else 'SAME'
    Ho, pt = out(H, kh, s, pad); Wo, pl = out(W, kw, s, pad)
    torch.manual_seed(0)
    x = torch.randn(B, H, W, Cin, device=dev).to(torch.bfloat16)
    K = kh * kw * Cin
    Kpad = (K + 63) // 64 * 64
    w = torch.zeros(Cout, Kpad, device=dev)
    w[:, :K] = torch.randn(Cout, K, device=dev) / K ** 0.5
    w = w.to(torch.bfloat16).reshape(-1)
    scale = torch.rand(Cout, device=dev) + 0.5; shift = torch.randn(Cout, device=dev) * 0.1
    wt = L.ConvWeight(w.data_ptr(), scale.data_ptr(), shift.data_ptr())
    st = L.stream_ptr()
    fl = 2.0 * B * Ho * Wo * K * Cout
    ref = None
    print('case', case, 'M=%d K=%d' % (B * Ho * Wo, K))
    for tile in tiles:
        y = torch.full((B, Ho, Wo, Cout), float('nan'), dtype=torch.bfloat16, device=dev)
        op = L.CnnOp(kind=0, src=0, dst=1, src_coff=0, dst_coff=0, H=H, W=W, Cin=Cin, Cout=Cout, KH=kh, KW=kw, SH=s, SW=s,
                     PT=pt, PL=pl, Ho=Ho, Wo=Wo, weight=0, relu=1, out_f32=0, tile=tile)

        def run():
            L.check(lib.comic_conv2d_bn_relu(C.byref(op), x.data_ptr(), Cin, y.data_ptr(), Cout, C.byref(wt), B, 1, st), 'conv')
        try:
            run(); run()
        except L.ComicHipError as e:
            print('  tile %2d n/a (%s)' % (tile, str(e)[:60])); continue
        torch.cuda.synchronize()
        if ref is None:
            ref = y.clone()
            same = 'ref'
        else:
            same = 'identical' if torch.equal(y.view(torch.int16), ref.view(torch.int16)) else 'DIFFERENT (%d elems)' % int(
                (y.view(torch.int16) != ref.view(torch.int16)).sum())
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        best = 1e9
        for _ in range(3):
            e0.record()
            for _ in range(reps):
                run()
            e1.record(); e1.synchronize()
            best = min(best, e0.elapsed_time(e1) / reps * 1e3)
        print('  tile %2d  %8.1f us  %7.1f TF/s  %s' % (tile, best, fl / best / 1e6, same))
