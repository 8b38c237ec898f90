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
plan=nets.CnnPlan('inception_v3',(224,224))
enc=nets.CnnEncoder(plan,plan.init_params(0),64,'bf16',dev)
sA=torch.cuda.Stream(); sB=torch.cuda.current_stream()
# warmup + graph capture on respective streams
with torch.cuda.stream(sA):
    for _ in range(3): im,fm=enc.forward(images,use_graph=True)
torch.cuda.synchronize()
for _ in range(3): dec.train_step(fm,im,caps,training=True,use_graph=True)
torch.cuda.synchronize()
ev_cnn=torch.cuda.Event(); ev_used=torch.cuda.Event()
def step_seq():
    im,fm=enc.forward(images,use_graph=True)
    dec.train_step(fm,im,caps,training=True,use_graph=True)
    opt.step(dec.grads,1e-3)
for _ in range(3): step_seq()
torch.cuda.synchronize(); t0=time.perf_counter()
for _ in range(30): step_seq()
torch.cuda.synchronize(); print('sequential %.3f ms'%((time.perf_counter()-t0)/30*1e3))
# pipelined: CNN(i+1) on sA overlaps decoder(i) on sB
with torch.cuda.stream(sA):
    im,fm=enc.forward(images,use_graph=True); ev_cnn.record(sA)
torch.cuda.synchronize(); t0=time.perf_counter()
ctx=list(dec._ctx.values())[-1]
for i in range(30):
    sB.wait_event(ev_cnn)
    # decoder step copies fm/im into its ctx first; then record "used"
    ctx.fm.copy_(fm); ctx.im.copy_(im); ev_used.record(sB)
    with torch.cuda.stream(sA):
        sA.wait_event(ev_used)
        im,fm=enc.forward(images,use_graph=True); ev_cnn.record(sA)
    dec.train_step(ctx.fm,ctx.im,caps,training=True,use_graph=True)
    opt.step(dec.grads,1e-3)
torch.cuda.synchronize(); print('pipelined %.3f ms'%((time.perf_counter()-t0)/30*1e3), float(ctx.loss[0]))
