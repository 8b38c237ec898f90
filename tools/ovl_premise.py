"""Premise of the half-chip persistent loops: does a decoder step whose loops hold 128 CUs (batch 32 = two 16-row groups)
keep its speed while the InceptionV3 forward runs beside it on a CU-masked stream (the other 128 CUs)?
   MASK=lo128|hi128|even|none  GRAPH=0|1  python tools/ovl_premise.py"""
import ctypes, os, sys, time
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), '..'))
import numpy as np, torch
import bench
from comic_amd import decoder as cdec, nets, optim

dev = 'cuda:0'
B = int(os.environ.get('B', '32'))
NIMG = int(os.environ.get('NIMG', '1280'))
NSTEP = int(os.environ.get('NSTEP', '20'))
MASKS = {'lo128': [0xffffffff] * 4 + [0] * 4, 'hi128': [0] * 4 + [0xffffffff] * 4, 'even': [0x55555555] * 8,
         'lo16': [0x0000ffff] * 8, 'hi16': [0xffff0000] * 8}


def masked_stream(name):
    if name == 'none':
        return torch.cuda.Stream(device=dev)
    hip = ctypes.CDLL('libamdhip64.so')
    st = ctypes.c_void_p()
    words = (ctypes.c_uint32 * 8)(*MASKS[name])
    rc = hip.hipExtStreamCreateWithCUMask(ctypes.byref(st), 8, words)
    assert rc == 0, rc
    return torch.cuda.ExternalStream(st.value, device=dev)


plan = nets.CnnPlan('inception_v3', (224, 224), group_branches=True, pool_after_projection=True, fuse_pools=True)
enc = nets.CnnEncoder(plan, plan.init_params(0), NIMG, 'bf16', dev)
enc.autotune(cache=os.environ.get('COMIC_TUNE_CACHE') or None)
x = torch.rand(NIMG, 224, 224, 3, device=dev) * 2 - 1
spec = cdec.DecoderSpec()
dec = cdec.Decoder(spec, None, dev, seed=1)
opt = optim.AdamTF(dec.params)
rng = np.random.default_rng(0)
fm = torch.randn(B, spec.M, spec.C, device=dev)
im = torch.randn(B, spec.Cg, device=dev)
caps = [bench.synth_captions(rng, B) for _ in range(4)]
G = os.environ.get('GRAPH', '1') == '1'


def dec_steps(n):
    for i in range(n):
        r = dec.train_step(fm, im, caps[i % 4], training=True, use_graph=G); opt.step(dec.grads, 1e-3)
    return r


for _ in range(3): enc.forward(x, use_graph=G)
dec_steps(4)
torch.cuda.synchronize()
main = torch.cuda.Stream(device=dev)      # non-blocking: a CU-masked stream is a blocking one and would serialise against stream 0
torch.cuda.set_stream(main)
dec_steps(2)
torch.cuda.synchronize()


def ev(): return torch.cuda.Event(enable_timing=True)


# decoder alone
a, b = ev(), ev()
a.record(); r = dec_steps(NSTEP); b.record(); torch.cuda.synchronize()
print('decoder alone B=%d: %.3f ms/step (path %d, loss %.3f)' % (B, a.elapsed_time(b) / NSTEP, dec.lib.comic_decoder_train_path(), float(r['loss'])))
for name in os.environ.get('MASK', 'none,lo128,hi128').split(','):
    side = masked_stream(name)
    enc._drop_graphs()
    with torch.cuda.stream(side):
        for _ in range(3): enc.forward(x, use_graph=G)
        a, b = ev(), ev()
        a.record(side)
        for _ in range(3): enc.forward(x, use_graph=G)
        b.record(side)
    torch.cuda.synchronize()
    t_enc = a.elapsed_time(b) / 3
    # both: the forward on the side stream, NSTEP decoder steps on the main stream
    res = []
    for rep in range(3):
        a, b, c, d = ev(), ev(), ev(), ev()
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        with torch.cuda.stream(side):
            a.record(side); enc.forward(x, use_graph=G); b.record(side)
        c.record(main); r = dec_steps(NSTEP); d.record(main)
        torch.cuda.synchronize()
        res.append((a.elapsed_time(b), c.elapsed_time(d) / NSTEP, (time.perf_counter() - t0) * 1e3, float(r['loss'])))
    print('mask %-6s: forward of %d alone on its stream %.2f ms; beside %d decoder steps: forward %s ms, decoder %s ms/step, wall %s ms, loss %.3f' % (
        name, NIMG, t_enc, NSTEP, '/'.join('%.2f' % q[0] for q in res), '/'.join('%.3f' % q[1] for q in res), '/'.join('%.1f' % q[2] for q in res), res[-1][3]))
