"""Beam-3 decode of the word baseline (bench.py extras: V = 25 599, 1 head, no projection, batch 50, 30 steps) alone:
timing line + a target for rocprofv3 --kernel-trace (GRAPH=0: eager launches, visible to the trace)."""
import os, sys, time
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), '..'))
import numpy as np, torch
from comic_amd import decoder as cdec
dev = 'cuda:0'
B, V, W = int(os.environ.get('B', '50')), 25599, int(os.environ.get('W', '3'))
spec = (cdec.DecoderSpec() if os.environ.get('SPEC') == 'radix' else
        cdec.DecoderSpec(V=V, H=1, fm_projection=None, token_type='word', start_id=V - 2, end_id=V - 1))   # SPEC=radix: COMIC-256 (SCST rollouts: B=32 W=7)
dec = cdec.Decoder(spec, None, dev, seed=3)
if os.environ.get('SPEC') == 'radix':
    dec.params.view('b_o')[spec.end_id] = -9.0      # rollouts run all steps
fm = torch.randn(B, spec.M, spec.C, device=dev)
im = torch.randn(B, spec.Cg, device=dev)
graph = os.environ.get('GRAPH', '1') == '1'
for _ in range(3):
    r = dec.beam_search(fm, im, W, 30, want_attention=False, use_graph=graph) if 'use_graph' in dec.beam_search.__code__.co_varnames else dec.beam_search(fm, im, W, 30, want_attention=False)
torch.cuda.synchronize()
n, t0 = 10, time.perf_counter()
for _ in range(n):
    r = dec.beam_search(fm, im, W, 30, want_attention=False, use_graph=graph) if 'use_graph' in dec.beam_search.__code__.co_varnames else dec.beam_search(fm, im, W, 30, want_attention=False)
torch.cuda.synchronize()
steps = int(r['predicted_ids'].shape[0])
print('beam-%d batch %d: %.3f ms per call, %d steps, %.1f us per step' % (W, B, (time.perf_counter() - t0) / n * 1e3, steps, (time.perf_counter() - t0) / n / steps * 1e6))
