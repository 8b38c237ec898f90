import sys, numpy as np, torch, math
sys.path.insert(0, '.')
from tests.test_gpu_ops import _run_conv, _ref_conv
from tests.gpu_util import rel_err
rng = np.random.default_rng(0)
def run(B,H,W,Cin,Cout,k,s,pad,dtype):
    x = rng.standard_normal((B,H,W,Cin)).astype(np.float32)
    w = (rng.standard_normal((k[0],k[1],Cin,Cout))/math.sqrt(k[0]*k[1]*Cin)).astype(np.float32)
    beta = np.zeros(Cout,np.float32); mean=np.zeros(Cout,np.float32); var=np.ones(Cout,np.float32)
    from oracle import cnn_ref
    if dtype=='bf16': x = cnn_ref.bf16_round(x)
    ref = _ref_conv(x,w,beta,mean,var,s,pad,dtype,relu=0)
    got = _run_conv(x,w,beta,mean,var,s,pad,dtype,relu=0)
    e = rel_err(got,ref)
    bad = np.abs(got-ref) > 1e-2*np.abs(ref).max()
    print(dtype,(B,H,W,Cin,Cout,k,s,pad),'err %.3e'%e, 'bad frac %.3f'%bad.mean(), 'bad per-channel', bad.reshape(-1,Cout).mean(0)[:8].round(2), 'bad per row m', bad.reshape(-1,Cout).mean(1)[:20].round(2))
    return got, ref
for dt in ('f32','bf16'):
    run(1,8,8,32,32,(1,1),1,'VALID',dt)
    run(1,8,8,64,32,(1,1),1,'VALID',dt)
    run(1,8,8,32,64,(1,1),1,'VALID',dt)
    run(2,8,8,32,32,(3,3),1,'VALID',dt)
    run(2,8,8,32,32,(3,3),1,'SAME',dt)
    run(2,17,15,32,32,(3,3),1,'VALID',dt)
    run(2,25,25,288,384,(3,3),2,'VALID',dt)
