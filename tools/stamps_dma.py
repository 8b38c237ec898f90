"""Phase stamps of the im2col DMA conv kernel (lib built with -DCOMIC_STAMPS)."""
import os, sys, ctypes as C
sys.path.insert(0, '.')
import numpy as np, torch
from comic_amd import _lib as L
L.LIB_PATH = L.LIB_PATH.replace('libcomic_hip.so', 'libcomic_hip_dbg.so')
lib = L.load()
exec(open('tools/one_conv.py').read().split("for tile in tiles:")[0].split("lib = L.load()")[1])
for tile in tiles:
    op = L.CnnOp(kind=0, src=0, dst=1, src_coff=0, dst_coff=0, H=H, W=W, Cin=Cin, Cout=Cout, KH=kh, KW=kw, SH=s, SW=s,
                 PT=pt, PL=pl, Ho=Ho, Wo=Wo, weight=0, relu=1, out_f32=0, tile=tile)
    for _ in range(3):
        L.check(lib.comic_conv2d_bn_relu(C.byref(op), x.data_ptr(), Cin, y.data_ptr(), Cout, C.byref(wt), B, 1, st), 'conv')
    torch.cuda.synchronize()
    n = 16384
    buf = np.zeros(n * 8, np.uint64)
    lib.comic_debug_read_stamps.argtypes = [C.c_void_p, C.c_int]
    assert lib.comic_debug_read_stamps(buf.ctypes.data, n * 8) == 0
    t = buf.reshape(n, 8).astype(np.int64)
    t = t[t[:, 3] > 0]
    nk = np.median(t[:, 6])
    print('tile', tile, 'workgroups', len(t), 'k-tiles', nk)
    print('   prologue issue  median %8.0f' % np.median(t[:, 1] - t[:, 0]))
    print('   k loop          median %8.0f  per k-tile %6.0f' % (np.median(t[:, 2] - t[:, 1]), np.median(t[:, 2] - t[:, 1]) / nk))
    print('      of which wait+barrier %8.0f  per k-tile %6.0f' % (np.median(t[:, 5]), np.median(t[:, 5]) / nk))
    print('      of which DMA issue    %8.0f  per k-tile %6.0f' % (np.median(t[:, 4]), np.median(t[:, 4]) / nk))
    print('      first k-tile wait     %8.0f' % np.median(t[:, 7]))
    print('   epilogue        median %8.0f' % np.median(t[:, 3] - t[:, 2]))
    print('   workgroup total median %8.0f   (s_memtime ticks)' % np.median(t[:, 3] - t[:, 0]))
