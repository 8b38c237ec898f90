import os, sys, time, numpy as np, torch
sys.path.insert(0, '.')
from comic_amd import decoder as cdec, nets, trainer, optim
import bench
dev='cuda:0'
rng=np.random.default_rng(0)
images=torch.from_numpy(rng.uniform(-1,1,(64,224,224,3)).astype(np.float32)).to(dev)
caps=bench.synth_captions(rng,64)
spec=cdec.DecoderSpec()
dec=cdec.Decoder(spec,None,dev,1)
opt=optim.AdamTF(dec.params)
for lanes in (0,1,0,1):
    plan=nets.CnnPlan('inception_v3',(224,224),branch_streams=bool(lanes))
    enc=nets.CnnEncoder(plan,plan.init_params(0),64,'bf16',dev)
    im,fm=enc.forward(images)           # eager warmup
    dec.train_step(fm,im,caps,training=True)   # eager warm-up creates ctx
    ctx=list(dec._ctx.values())[-1]
    torch.cuda.synchronize()
    g=torch.cuda.CUDAGraph()
    with torch.cuda.graph(g):
        enc._run()
        im,fm=enc.bufs[plan.pooled].reshape(64,-1), enc.bufs[plan.fm].reshape(64,25,2048)
        ctx.fm.copy_(fm); ctx.im.copy_(im)
        dec._train_device(ctx)
        opt.step(dec.grads,1e-3)
    for _ in range(3): g.replay()
    torch.cuda.synchronize()
    t0=time.perf_counter()
    for _ in range(30): g.replay()
    torch.cuda.synchronize(); t_all=(time.perf_counter()-t0)/30*1e3
    print('lanes',lanes,'one graph per step: %.3f ms'%t_all, float(ctx.loss[0]))
