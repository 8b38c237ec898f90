"""Grouped GEMM micro-benchmark: the weight-gradient group of the decoder step at the bench geometry (and the keys / logits
groups), through comic_gemm_group, for several planner settings."""
import os, sys
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), '..'))
import numpy as np, torch
import comic_amd._lib as L
lib = L.load()
dev = 'cuda:0'
T, B, D, E, A, V, C, M = 29, 64, 512, 256, 512, 258, 2048, 25
Wd, EA, TB, R = E + A + D, E + A, T * B, (T + 1) * B
f = lambda *s: torch.randn(*s, device=dev)
xh, dg, fm, dkeys, y, dq, dlog = f(R, Wd), f(R, 4 * D), f(B * M, C), f(B * M, D), f(TB, D), f(TB, D), f(TB, 260)
K, Wm, Wo, pg, im, Winit = f(Wd, 4 * D), f(C, D), f(D, 260), f(4 * B, 3 * D + 1), f(B, C), f(C, EA)
mask = (torch.rand(TB, EA, device=dev) < 0.75).float()
outs = dict(gK=f(Wd, 4 * D), gWm=f(C, D), demb=f(TB, E), gWq=f(D, D), gWo=f(D, V), gb=f(4 * D), gbo=f(V), gv=f(D), gg=f(D), gbt=f(D),
            gtau=f(1), dxi=f(B, EA), keys=f(B * M, D), xi=f(B, Wd), logits=f(TB, V), dy=f(TB, D), gates=f(B, 4 * D))
def prob(ty, A_, B_, C_, M_, N_, K_, lda, ldb, ldc, ones=0, mask_=None, ldm=0, bias=None):
    q = L.GemmProb()
    q.A = A_.data_ptr() if A_ is not None else None
    q.B, q.C = B_.data_ptr(), C_.data_ptr()
    q.bias = bias.data_ptr() if bias is not None else None
    q.mask = mask_.data_ptr() if mask_ is not None else None
    q.M, q.N, q.K, q.lda, q.ldb, q.ldc, q.ld_mask = M_, N_, K_, lda, ldb, ldc, ldm
    q.alpha, q.beta, q.keep, q.type, q.ones_a = 1.0, 0.0, 0.75, ty, ones
    return q
o = outs
groups = {
 'post': [prob(0, xh, dg, o['gK'], Wd, 4 * D, R, Wd, 4 * D, 4 * D), prob(0, fm, dkeys, o['gWm'], C, D, B * M, C, D, D),
          prob(2, dg, K, o['demb'], TB, E, 4 * D, 4 * D, 4 * D, E, 0, mask, EA), prob(0, y, dq, o['gWq'], D, D, TB, D, D, D),
          prob(0, y, dlog, o['gWo'], D, V, TB, D, 260, V), prob(0, None, dg, o['gb'], 1, 4 * D, R, 1, 4 * D, 4 * D, 1),
          prob(0, None, dlog, o['gbo'], 1, V, TB, 1, 260, V, 1), prob(0, None, pg, o['gv'], 1, D, 4 * B, 1, 3 * D + 1, D, 1),
          prob(0, None, pg[:, D:], o['gg'], 1, D, 4 * B, 1, 3 * D + 1, D, 1), prob(0, None, pg[:, 2 * D:], o['gbt'], 1, D, 4 * B, 1, 3 * D + 1, D, 1),
          prob(0, None, pg[:, 3 * D:], o['gtau'], 1, 1, 4 * B, 1, 3 * D + 1, 1, 1), prob(2, dg[TB:], K, o['dxi'], B, EA, 4 * D, 4 * D, 4 * D, EA, 0, mask, EA)],
 'dK': [prob(0, xh, dg, o['gK'], Wd, 4 * D, R, Wd, 4 * D, 4 * D)],
 'dWm': [prob(0, fm, dkeys, o['gWm'], C, D, B * M, C, D, D)],
 'pre': [prob(1, fm, Wm, o['keys'], B * M, D, C, C, D, D), prob(1, im, Winit, o['xi'], B, EA, C, C, EA, Wd, 0, mask, EA)],
 'gates': [prob(1, o['xi'], K, o['gates'], B, 4 * D, EA, Wd, 4 * D, 4 * D)],
 'logits': [prob(1, y, Wo, o['logits'], TB, V, D, D, 260, V)],
 'dy': [prob(2, dlog, Wo, o['dy'], TB, D, 260, 260, 260, D)],
}
ws = torch.empty(256 << 20, dtype=torch.uint8, device=dev)
def run(name, n=20):
    arr = (L.GemmProb * len(groups[name]))(*groups[name])
    st = torch.cuda.current_stream().cuda_stream
    for _ in range(3):
        L.check(lib.comic_gemm_group(arr, len(arr), ws.data_ptr(), ws.numel(), st))
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n):
        L.check(lib.comic_gemm_group(arr, len(arr), ws.data_ptr(), ws.numel(), st))
    e1.record(); e1.synchronize()
    return e0.elapsed_time(e1) / n * 1e3
only = os.environ.get('GROUPS')
if only:
    groups = {k: v for k, v in groups.items() if k in only.split(',')}
for tgt in (int(x) for x in os.environ.get('TARGETS', '480,720,960,1440').split(',')):
    for xcd in (1, 3):      # 1: producer / consumer kernel, 3: four-wave kernel (both with the XCD-contiguous item order)
        lib.comic_debug_gemm_group_tuning(tgt, xcd)
        print('target %4d xcd %d: ' % (tgt, xcd) + '  '.join('%s %.1f' % (k, run(k)) for k in groups), flush=True)
