"""One conv layer alone (bf16), explicit tile id: timing + a target for rocprofv3 --pmc.
   CASE=B,H,W,Cin,Cout,kh,kw,stride,pad  TILE=13  REPS=20"""
import os, sys, ctypes as C
sys.path.insert(0, '.')
import numpy as np, torch
from comic_amd import _lib as L
lib = L.load()
B, H, W, Cin, Cout, kh, kw, s = [int(v) for v in os.environ.get('CASE', '64,54,54,80,192,3,3,1').split(',')]
pad = os.environ.get('PAD', 'VALID')
tiles = [int(t) for t in os.environ.get('TILE', '13').split(',')]
reps = int(os.environ.get('REPS', '20'))


def out(size, k, s, pad):
    if pad == 'SAME':
        o = -(-size // s); tot = max((o - 1) * s + k - size, 0); return o, tot // 2
    return (size - k) // s + 1, 0


Ho, pt = out(H, kh, s, pad); Wo, pl = out(W, kw, s, pad)
dev = 'cuda:0'
x = (torch.randn(B, H, W, Cin, device=dev)).to(torch.bfloat16)
K = kh * kw * Cin
w = (torch.randn(Cout * ((K + 63) // 64 * 64), device=dev) / K ** 0.5).to(torch.bfloat16)
scale = torch.ones(Cout, device=dev); shift = torch.zeros(Cout, device=dev)
y = torch.empty(B, Ho, Wo, Cout, dtype=torch.bfloat16, device=dev)
wt = L.ConvWeight(w.data_ptr(), scale.data_ptr(), shift.data_ptr())
st = L.stream_ptr()
fl = 2.0 * B * Ho * Wo * K * Cout
for tile in tiles:
    op = L.CnnOp(kind=0, src=0, dst=1, src_coff=0, dst_coff=0, H=H, W=W, Cin=Cin, Cout=Cout, KH=kh, KW=kw, SH=s, SW=s,
                 PT=pt, PL=pl, Ho=Ho, Wo=Wo, weight=0, relu=1, out_f32=0, tile=tile)
    def run():
        L.check(lib.comic_conv2d_bn_relu(C.byref(op), x.data_ptr(), Cin, y.data_ptr(), Cout, C.byref(wt), B, 1, st), 'conv')
    try:
        run(); run()
    except L.ComicHipError as e:
        print('tile', tile, 'n/a'); continue
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps):
        run()
    e1.record(); e1.synchronize()
    us = e0.elapsed_time(e1) / reps * 1e3
    print('tile %2d  %7.1f us  %7.1f TF/s' % (tile, us, fl / us / 1e6))
