"""Diagnostic: gradients of a small decoder step with the grouped GEMM path vs the separate-launch path."""
import os, sys
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), '..'))
import numpy as np, torch
from comic_amd import decoder as cdec
import comic_amd._lib as L
from oracle import decoder_ref as dr
D = int(os.environ.get('D', '128'))
def run(flag, spec, p, fm, im, caps):
    os.environ['COMIC_GROUP_GEMM'] = flag
    dec = cdec.Decoder(spec, p, 'cuda:0')
    res = dec.train_step(fm, im, caps, training=False, want_input_grads=True)
    torch.cuda.synchronize()
    g = dec.grads.to_numpy()
    g['dfm'] = res['dfm'].cpu().numpy(); g['dim'] = res['dim_embed'].cpu().numpy()
    return g, float(res['loss'])
for (B, M, C, Lc) in ((3, 9, 48, 9), (64, 25, 256, 20)):
    spec = cdec.DecoderSpec(D=D, E=64, C=C, Cg=C, M=M, H=4)
    cfg = dr.DecoderConfig(rnn_size=D, word_size=64, C=C, Cg=C, M=M, H=4) if hasattr(dr, 'DecoderConfig') else None
    rng = np.random.default_rng(5)
    p = cdec.init_params(spec, 3)
    fm = torch.from_numpy(rng.standard_normal((B, M, C)).astype(np.float32)).cuda()
    im = torch.from_numpy(rng.standard_normal((B, C)).astype(np.float32)).cuda()
    caps = np.full((B, Lc + 2), -1, np.int64)
    for b in range(B):
        n = int(rng.integers(3, Lc + 1))
        caps[b, 0] = spec.start_id; caps[b, 1:1 + n] = rng.integers(0, 256, n); caps[b, 1 + n] = spec.end_id
    g0, l0 = run('0', spec, p, fm, im, caps)
    g1, l1 = run('1', spec, p, fm, im, caps)
    print('B %d M %d C %d: loss %.6f %.6f' % (B, M, C, l0, l1))
    for k in g0:
        a, b = g1[k].astype(np.float64), g0[k].astype(np.float64)
        if not a.size: continue
        d = np.abs(a - b)
        rms = np.sqrt(np.mean(b * b)) + 1e-30
        print('  %-8s max|b| %.3e rms %.3e  max|a-b| %.3e  /max %.2e  /rms %.2e  elementwise %.3f' % (
            k, np.abs(b).max(), rms, d.max(), d.max() / (np.abs(b).max() + 1e-30), d.max() / rms,
            (d / (1e-3 * np.abs(b) + 1e-3 * rms)).max()))
