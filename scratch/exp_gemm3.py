import sys; sys.path.insert(0, '.')
import numpy as np, torch
from comic_amd import _lib as L
lib = L.load(); st = L.stream_ptr()
def t(fn, n=20):
    for _ in range(3): fn()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n): fn()
    e1.record(); e1.synchronize(); return e0.elapsed_time(e1) / n * 1e3
for (M, N, K, ta, tb) in [(1600, 512, 2048, 0, 0), (1856, 258, 512, 0, 0), (1856, 512, 258, 0, 1), (1280, 2048, 1856, 1, 0), (512, 258, 1856, 1, 0), (512, 512, 1856, 1, 0), (2048, 512, 1600, 1, 0), (2048, 768, 64, 1, 0)]:
    A = torch.randn((K, M) if ta else (M, K), device='cuda'); B = torch.randn((N, K) if tb else (K, N), device='cuda'); Cc = torch.empty(M, N, device='cuda')
    ws = torch.empty(8 << 20, dtype=torch.uint8, device='cuda')
    f1 = lambda: L.check(lib.comic_gemm_f32_splitk(A.data_ptr(), B.data_ptr(), Cc.data_ptr(), None, M, N, K, A.shape[1], B.shape[1], N, ta, tb, 1.0, 0.0, ws.data_ptr(), 8 << 20, st))
    f3 = lambda: L.check(lib.comic_gemm_f32_split3(A.data_ptr(), B.data_ptr(), Cc.data_ptr(), None, M, N, K, A.shape[1], B.shape[1], N, ta, tb, 1.0, 0.0, ws.data_ptr(), 8 << 20, st))
    print('%5dx%5dx%5d ta%d tb%d  exact %7.1f us   split3 %7.1f us   (%.0f TF/s)' % (M, N, K, ta, tb, t(f1), t(f3), 2 * M * N * K / t(f3) / 1e6))
