"""Fixed vs per-k-iteration cost of the DMA conv kernel: one conv, KH swept, tile fixed."""
import sys, ctypes as C
sys.path.insert(0, '.')
import torch
from comic_amd import _lib as L
lib = L.load()
dev = 'cuda:0'
B = 64
def bench(H, Cin, Cout, KH, KW, tile, reps=50):
    x = torch.randn(B, H, H, Cin, device=dev).bfloat16()
    y = torch.empty(B, H, H, Cout, device=dev, dtype=torch.bfloat16)
    K = KH * KW * Cin; kpad = (K + 63) // 64 * 64
    w = torch.randn(Cout * kpad, device=dev).bfloat16()
    sc = torch.ones(Cout, device=dev); sh = torch.zeros(Cout, device=dev)
    op = L.CnnOp()
    for k, v in dict(kind=0, src=0, dst=1, src_coff=0, dst_coff=0, H=H, W=H, Cin=Cin, Cout=Cout, KH=KH, KW=KW, SH=1, SW=1,
                     PT=(KH - 1) // 2, PL=(KW - 1) // 2, Ho=H, Wo=H, weight=0, relu=1, out_f32=0, src_f32=0, lane=0, tile=tile, group=0).items():
        setattr(op, k, v)
    wt = L.ConvWeight(); wt.w, wt.scale, wt.shift = w.data_ptr(), sc.data_ptr(), sh.data_ptr()
    st = L.stream_ptr()
    def run():
        L.check(lib.comic_conv2d_bn_relu(C.byref(op), x.data_ptr(), Cin, y.data_ptr(), Cout, C.byref(wt), B, 1, st), 'conv')
    for _ in range(5): run()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps): run()
    e1.record(); e1.synchronize()
    return e0.elapsed_time(e1) / reps * 1e3
for (H, Cin, Cout) in [(12, 192, 192)]:
    for tile in (3, 1):
        row = []
        for KH in (1, 3, 5, 7, 9, 15):
            t = bench(H, Cin, Cout, KH, 1, tile)
            nk = (KH * Cin + 63) // 64
            row.append('KH%d nk%3d %6.1fus' % (KH, nk, t))
        print('H%d Cin%d Cout%d tile%d | ' % (H, Cin, Cout, tile) + ' | '.join(row))
