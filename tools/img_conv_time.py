"""Image-resident conv kernel (tile 55) against the autotuned best of the other variants, per layer shape, at B images.
   B=1280 python tools/img_conv_time.py"""
import os, sys, ctypes as C
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), '..'))
import torch
from comic_amd import _lib as L
lib = L.load()
B = int(os.environ.get('B', '1280'))
CASES = [(12, 12, 128, 192, 7, 1), (12, 12, 160, 192, 1, 7), (12, 12, 192, 192, 7, 1), (12, 12, 192, 192, 1, 7),
         (12, 12, 128, 128, 1, 7), (12, 12, 160, 160, 7, 1), (25, 25, 64, 96, 3, 3), (25, 25, 96, 96, 3, 3),
         (5, 5, 448, 384, 3, 3), (5, 5, 384, 384, 1, 3)]
OTHERS = [int(t) for t in os.environ.get('OTHERS', '13,18,19,22,23,26,27,28,32,34,35,36,38,39,40,41,42,43,44').split(',')]
dev = 'cuda:0'
st = L.stream_ptr()
for (H, W, Cin, Cout, kh, kw) in CASES:
    K = kh * kw * Cin; Kpad = (K + 63) // 64 * 64
    x = torch.randn(B, H, W, Cin, device=dev).to(torch.bfloat16)
    w = (torch.randn(Cout, Kpad, device=dev) / K ** 0.5).to(torch.bfloat16)
    frag = torch.zeros_like(w)
    table = torch.tensor([[0, Cout, Kpad]], dtype=torch.int64, device=dev)
    L.check(lib.comic_cnn_pack_frag_weights(w.data_ptr(), frag.data_ptr(), table.data_ptr(), 1, w.numel(), st), 'pack')
    scale = torch.ones(Cout, device=dev); shift = torch.zeros(Cout, device=dev)
    y = torch.empty(B, H, W, Cout, dtype=torch.bfloat16, device=dev)
    wt = L.ConvWeight(w.data_ptr(), scale.data_ptr(), shift.data_ptr(), frag.data_ptr())
    fl = 2.0 * B * H * W * K * Cout
    res = {}
    for tile in [L.IMG_TILE] + OTHERS:
        op = L.CnnOp(kind=0, src=0, dst=1, H=H, W=W, Cin=Cin, Cout=Cout, KH=kh, KW=kw, SH=1, SW=1, PT=(kh - 1) // 2,
                     PL=(kw - 1) // 2, Ho=H, Wo=W, weight=0, relu=1, tile=tile)
        def run():
            return lib.comic_conv2d_bn_relu(C.byref(op), x.data_ptr(), Cin, y.data_ptr(), Cout, C.byref(wt), B, 1, st)
        if run() != 0:
            continue
        run()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        best = 1e9
        for _ in range(3):
            e0.record()
            for _ in range(10):
                run()
            e1.record(); e1.synchronize()
            best = min(best, e0.elapsed_time(e1) / 10 * 1e3)
        res[tile] = best
    img = res.get(L.IMG_TILE)
    other = min((v, t) for t, v in res.items() if t != L.IMG_TILE)
    print('%2dx%-2d %dx%d %3d->%3d  img %7.1f us %7.1f TF/s | best other: tile %2d %7.1f us %7.1f TF/s | x%.2f' % (
        H, W, kh, kw, Cin, Cout, img, fl / img / 1e6, other[1], other[0], fl / other[0] / 1e6, other[0] / img))
