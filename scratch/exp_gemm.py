import sys, time, numpy as np, torch
sys.path.insert(0,'.')
import comic_amd._lib as L
lib=L.load(); dev='cuda:0'
def bench(M,N,K,tb,ws_mb):
    A=torch.randn(M,K,device=dev); B=torch.randn((N,K) if tb else (K,N),device=dev); C=torch.empty(M,N,device=dev)
    ws=torch.empty(ws_mb<<20,dtype=torch.uint8,device=dev) if ws_mb else None
    st=torch.cuda.current_stream().cuda_stream
    def run():
        L.check(lib.comic_gemm_f32_splitk(A.data_ptr(),B.data_ptr(),C.data_ptr(),None,M,N,K,K,B.shape[1],N,0,tb,1.0,0.0,ws.data_ptr() if ws is not None else None, ws_mb<<20, st))
    for _ in range(5): run()
    e0=torch.cuda.Event(enable_timing=True); e1=torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(50): run()
    e1.record(); e1.synchronize()
    ref=(A@(B.t() if tb else B))
    err=float((C-ref).abs().max()/ref.abs().max())
    return e0.elapsed_time(e1)/50*1e3, err
for (M,N,K,tb) in [(64,2048,1280,0),(64,512,512,0),(64,1280,2048,1),(64,512,512,1),(1600,512,2048,0),(1856,258,512,0)]:
    for ws in (0,8):
        t,err=bench(M,N,K,tb,ws)
        print('M%d N%d K%d tb%d ws%dMB: %.1f us  %.1f TF/s  err %.1e'%(M,N,K,tb,ws,t,2*M*N*K/t/1e6,err))
