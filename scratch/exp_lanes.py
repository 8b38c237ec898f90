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
res={}
for lanes in (0,1,0,1):
    plan=nets.CnnPlan('inception_v3',(224,224),branch_streams=bool(lanes))
    enc=nets.CnnEncoder(plan,plan.init_params(0),64,'bf16',dev)
    for _ in range(3): im,fm=enc.forward(images,use_graph=True)
    torch.cuda.synchronize()
    t0=time.perf_counter()
    for _ in range(50): im,fm=enc.forward(images,use_graph=True)
    torch.cuda.synchronize(); t_cnn=(time.perf_counter()-t0)/50*1e3
    for _ in range(3): dec.train_step(fm,im,caps,training=True,use_graph=True)
    torch.cuda.synchronize()
    t0=time.perf_counter()
    for _ in range(30): dec.train_step(fm,im,caps,training=True,use_graph=True)
    torch.cuda.synchronize(); t_dec=(time.perf_counter()-t0)/30*1e3
    t0=time.perf_counter()
    for _ in range(30):
        im,fm=enc.forward(images,use_graph=True)
        dec.train_step(fm,im,caps,training=True,use_graph=True)
        opt.step(dec.grads,1e-3)
    torch.cuda.synchronize(); t_all=(time.perf_counter()-t0)/30*1e3
    print('lanes',lanes,'cnn %.3f dec %.3f all %.3f ms'%(t_cnn,t_dec,t_all))
