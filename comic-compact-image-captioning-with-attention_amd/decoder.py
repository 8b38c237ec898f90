"""Host side of the attention-LSTM decoder: parameter storage, input processing and the
calls into the native executors (`comic_decoder_train_step / _greedy / _beam`).

Counterpart of the decoder half of `ModelBase` (reference src/model_base.py:109-314,
:501-757) and of `common/ops_rnn.py`'s three dynamic-decode builders.  All arithmetic is in
the HIP library; this module only prepares integer/mask tables on the host (the reference
does the same work in `_process_inputs`, model_base.py:501-528) and post-processes ids.

Parameters live in ONE flat fp32 device buffer (views per variable) so that the optimiser
is a single fused kernel and the data-parallel gradient exchange is a single all-reduce.
"""
from __future__ import annotations

import ctypes as C
import math
from dataclasses import dataclass

import numpy as np

from . import _lib as L
from .ops import number_to_base

# TF variable names (scope Model/decoder/rnn_decoder/), SURVEY Appendix C
# TensorFlow variable names under `Model/decoder/rnn_decoder/` (checkpoint.decoder_var_names composes them; derived from
# the reference's scopes by oracle/ref_var_names.py and pinned by tests/golden/ref_var_names.json):
#   * created when the graph is CONSTRUCTED (model_base.py:147-176): memory / value layers, output projection, embedding
#     map, the rnn-init projection;
#   * STEP_VARS are created at the first decoder step, i.e. inside dynamic_decode's 'decoder' scope and the attention
#     wrapper's own layer scope (ops_rnn.py:543-562, :735);
#   * CELL_VARS follow the cell's FIRST call: 'rnn_init_input/' with rnn_init_method='first_input' (model_base.py:677-686),
#     the step scope with 'project_hidden'.
STEP_SCOPE = 'decoder/multi_head_attention_wrapper_v3/'
TF_NAMES = {
    'W_init': {'first_input': 'rnn_init_input/projection/weight', 'project_hidden': 'rnn_initial_state/weight'},
    'K': 'basic_lstm_cell/kernel', 'b': 'basic_lstm_cell/bias',
    'W_m': 'memory_layer/kernel', 'W_v': 'value_layer/kernel',
    'W_q': 'multi_add_attention/query_layer/kernel', 'v': 'multi_add_attention/attention_v',
    'ln_g': 'multi_add_attention/LN_tanh/gamma', 'ln_b': 'multi_add_attention/LN_tanh/beta',
    'tau': 'softmax_temperature', 'W_a': 'a_layer/kernel',
    'W_o': 'output_projection/kernel', 'b_o': 'output_projection/bias', 'emb': 'embedding_map',
    # --rnn_name GRU: tf.contrib.rnn.GRUCell's candidate pair (its gates pair takes K / b)
    'K_c': 'gru_cell/candidate/kernel', 'b_c': 'gru_cell/candidate/bias',
}
STEP_VARS = ('W_q', 'v', 'ln_g', 'ln_b', 'tau', 'W_a')
# per cell: TF scope of the cell's variables and the names of K / b inside it (model_base.py:606-632)
CELL_SCOPES = {'LSTM': ('basic_lstm_cell', 'kernel', 'bias'), 'LN_LSTM': ('layer_norm_basic_lstm_cell', 'kernel', None),
               'GRU': ('gru_cell', 'gates/kernel', 'gates/bias')}
# --rnn_name LN_LSTM: LayerNormBasicLSTMCell's five layer_norm scopes, in the order of comic_decoder_params::cell_ln
LN_LSTM_NORMS = (('i', 'input'), ('j', 'transform'), ('f', 'forget'), ('o', 'output'), ('c', 'state'))
for _k, _scope in LN_LSTM_NORMS:
    TF_NAMES['cln_%sg' % _k] = 'layer_norm_basic_lstm_cell/%s/gamma' % _scope
    TF_NAMES['cln_%sb' % _k] = 'layer_norm_basic_lstm_cell/%s/beta' % _scope
CELL_VARS = ('K', 'b', 'K_c', 'b_c') + tuple('cln_%s%s' % (k, s) for k, _ in LN_LSTM_NORMS for s in 'gb')


@dataclass
class DecoderSpec:
    """Static decoder geometry derived from a reference `Config` (src/train.py:29-162)."""
    D: int = 512
    E: int = 256
    V: int = 258
    C: int = 2048
    Cg: int = 2048
    H: int = 8
    M: int = 25
    fm_projection: str | None = 'tied'
    method: str = 'add_LN'
    prob: str = 'softmax'
    context_layer: bool = False
    init_method: str = 'first_input'
    token_type: str = 'radix'
    start_id: int = 256
    end_id: int = 257
    dropout_rnn_in: float = 0.35
    dropout_rnn_out: float = 0.35
    attn_keep_prob: float = 0.9
    map_loss_scale: float = 1.0
    l2_decay: float = 1e-5
    recurrent_dropout: bool = False      # DropoutWrapper(variational_recurrent=True): ONE input / output mask row
                                         # for all batch rows and time steps of a run [TF-1.9: noise shape [1, size]]
    rnn_name: str = 'LSTM'               # 'LSTM' | 'LN_LSTM' | 'GRU' (model_base.py:606-632); the persistent / fused /
                                         # streaming kernels are BasicLSTMCell's, the other two run per-step launches

    @property
    def A(self):                     # model_base.py:611-615
        return self.C if (self.fm_projection is None and not self.context_layer) else self.D

    @property
    def Cv(self):
        return self.C if self.fm_projection is None else self.D

    @classmethod
    def from_config(cls, c, fm_shape, im_embed_size):
        """c: reference-style Config (token_type, radix_base, rnn_size, ... itow/wtoi)."""
        if c.rnn_name not in L.CELLS:
            raise ValueError('Only `LSTM`, `LN_LSTM` and `GRU` are accepted.')      # model_base.py:631
        if c.attn_alignment_method not in ('add_LN', 'dot'):
            raise ValueError('Invalid alignment method.')          # model_base.py:133-138
        if c.attn_probability_fn not in ('softmax', 'sigmoid'):
            raise ValueError('Invalid alignment method.')
        if c.token_type == 'radix':
            V, start, end = c.radix_base + 2, c.radix_base, c.radix_base + 1     # model_base.py:42-43,701-703
        else:
            V, start, end = len(c.itow), c.wtoi['<GO>'], c.wtoi['<EOS>']
        return cls(D=c.rnn_size, E=c.rnn_word_size, V=V, C=fm_shape[-1], Cg=im_embed_size, H=c.attn_num_heads,
                   M=fm_shape[-2], fm_projection=c.cnn_fm_projection, method=c.attn_alignment_method,
                   prob=c.attn_probability_fn, context_layer=bool(c.attn_context_layer),
                   init_method=c.rnn_init_method, token_type=c.token_type, start_id=start, end_id=end,
                   dropout_rnn_in=getattr(c, 'dropout_rnn_in', 0.35), dropout_rnn_out=getattr(c, 'dropout_rnn_out', 0.35),
                   attn_keep_prob=c.attn_keep_prob, map_loss_scale=getattr(c, 'rnn_map_loss_scale', 1.0),
                   l2_decay=getattr(c, 'l2_decay', 1e-5), recurrent_dropout=bool(getattr(c, 'rnn_recurr_dropout', False)),
                   rnn_name=c.rnn_name)

    def param_shapes(self):
        D, E, A, V, C_, Cg = self.D, self.E, self.A, self.V, self.C, self.Cg
        s = {}
        s['W_init'] = (Cg, E + A) if self.init_method == 'first_input' else (Cg, D)
        if self.rnn_name == 'GRU':
            s['K'] = (E + A + D, 2 * D); s['b'] = (2 * D,)
            s['K_c'] = (E + A + D, D); s['b_c'] = (D,)
        elif self.rnn_name == 'LN_LSTM':       # no bias; the ten LayerNorm vectors are consecutive views (cell_ln)
            s['K'] = (E + A + D, 4 * D)
            for k, _ in LN_LSTM_NORMS:
                s['cln_%sg' % k] = (D,); s['cln_%sb' % k] = (D,)
        else:
            s['K'] = (E + A + D, 4 * D)
            s['b'] = (4 * D,)
        s['W_m'] = (C_, D)
        if self.fm_projection == 'independent':
            s['W_v'] = (C_, D)
        s['W_q'] = (D, D)
        if self.method == 'add_LN':
            s['v'] = (D,); s['ln_g'] = (D,); s['ln_b'] = (D,); s['tau'] = ()
        if self.context_layer:
            s['W_a'] = (self.Cv, D)
        s['W_o'] = (D, V)
        s['b_o'] = (V,)
        s['emb'] = (V, E)
        return s

    def desc(self, training):
        d = L.DecoderDesc()
        d.D, d.E, d.A, d.V, d.C, d.Cg, d.H, d.M, d.Cv = (self.D, self.E, self.A, self.V, self.C, self.Cg, self.H,
                                                       self.M, self.Cv)
        d.fm_projection = {None: 0, 'independent': 1, 'tied': 2}[self.fm_projection]
        d.method = {'add_LN': 0, 'dot': 1}[self.method]
        d.prob = {'softmax': 0, 'sigmoid': 1}[self.prob]
        d.context_layer = int(self.context_layer)
        d.init_method = {'first_input': 0, 'project_hidden': 1}[self.init_method]
        d.start_id, d.end_id = self.start_id, self.end_id
        d.keep_in = 1.0 - self.dropout_rnn_in if training else 1.0
        d.keep_out = 1.0 - self.dropout_rnn_out if training else 1.0
        d.keep_alpha = self.attn_keep_prob if training else 1.0
        d.map_loss_scale = self.map_loss_scale
        d.flags = L.decoder_flags_from_env()
        d.cell = L.CELLS[self.rnn_name]
        return d


def xavier_uniform(rng, shape):
    """slim.xavier_initializer() [TF-1.9] (model_base.py:823-831)."""
    if len(shape) > 1:
        fi, fo = shape[-2], shape[-1]
    else:
        fi = fo = shape[-1]
    lim = math.sqrt(6.0 / (fi + fo))
    return rng.uniform(-lim, lim, shape).astype(np.float32)


def init_params(spec: DecoderSpec, seed=0):
    rng = np.random.default_rng(seed)
    p = {}
    for k, shp in spec.param_shapes().items():
        if k == 'b' and spec.rnn_name == 'GRU':
            p[k] = np.ones(shp, np.float32)                        # [TF-1.9] GRUCell: gates bias starts at 1.0
        elif k in ('b', 'b_o', 'ln_b', 'b_c') or (k.startswith('cln_') and k.endswith('b')):
            p[k] = np.zeros(shp, np.float32)
        elif k == 'ln_g' or (k.startswith('cln_') and k.endswith('g')):
            p[k] = np.ones(shp, np.float32)
        elif k == 'tau':
            p[k] = np.array(5.0, np.float32)                       # ops_rnn.py:559
        else:
            p[k] = xavier_uniform(rng, shp)
    return p


class FlatParams:
    """One flat fp32 device buffer with named views (+ identical layouts for grads/Adam slots)."""
    ALIGN = 64      # floats: keeps every view 256-byte aligned (16-byte vector loads in the GEMM)

    def __init__(self, shapes: dict, device='cuda:0', status_tail=False):
        import torch
        self.torch = torch
        self.shapes = dict(shapes)
        self.offsets = {}
        off = 0
        for k, shp in shapes.items():
            self.offsets[k] = off
            n = int(np.prod(shp)) if len(shp) else 1
            off += (n + self.ALIGN - 1) // self.ALIGN * self.ALIGN
        self.numel = off
        self.device = device
        # status_tail: one aligned block behind the variables whose first float is the buffer's status word
        # (comic_decoder_params::status: voided-step flag of a gradient buffer, sticky count of a parameter buffer)
        self.tail = self.ALIGN if status_tail else 0
        self.data = torch.zeros(off + self.tail, dtype=torch.float32, device=device)

    def like(self):
        o = FlatParams.__new__(FlatParams)
        o.torch, o.shapes, o.offsets, o.numel, o.device = self.torch, self.shapes, self.offsets, self.numel, self.device
        o.tail = self.tail
        o.data = self.torch.zeros(self.numel + self.tail, dtype=self.torch.float32, device=self.device)
        return o

    @property
    def status(self):
        """The status word (a 1-element view) or None."""
        return self.data[self.numel:self.numel + 1] if self.tail else None

    @property
    def flat(self):
        """The variables without the status tail (what checkpoints hold)."""
        return self.data[:self.numel]

    def view(self, k):
        shp = self.shapes[k]
        n = int(np.prod(shp)) if len(shp) else 1
        return self.data[self.offsets[k]:self.offsets[k] + n].view(*shp) if len(shp) else \
            self.data[self.offsets[k]:self.offsets[k] + 1]

    def load(self, values: dict):
        for k in self.shapes:
            v = self.torch.from_numpy(np.ascontiguousarray(values[k], np.float32).reshape(-1))
            self.view(k).reshape(-1).copy_(v)

    def to_numpy(self):
        return {k: self.view(k).detach().cpu().numpy().reshape(self.shapes[k]) for k in self.shapes}

    def table(self):
        t = L.DecoderParams()
        base = self.data.data_ptr()
        for k in L.PARAM_NAMES:
            setattr(t, k, base + 4 * self.offsets[k] if k in self.offsets else None)
        t.status = base + 4 * self.numel if self.tail else None
        if 'cln_ig' in self.offsets:          # LN_LSTM: the ten vectors are consecutive views, ALIGN-padded (= the C stride)
            t.cell_ln = base + 4 * self.offsets['cln_ig']
        return t

    def n_params(self):
        return int(sum(int(np.prod(s)) if len(s) else 1 for s in self.shapes.values()))


def process_inputs(captions, token_type):
    """ModelBase._process_inputs (model_base.py:501-528) on the host.
    -> inputs [B,T] int32, targets [B,T] int32, masks [B,T] fp32, lens [B] int32."""
    sent = np.asarray(captions, np.int64)
    masks = np.sign((sent[:, 1:] + 1).astype(np.float32))
    lens = masks.sum(axis=1).astype(np.int32)
    if token_type == 'word':
        sent = np.maximum(sent, 0)
        inputs = sent[:, :-1]
    else:
        inputs = sent[:, :-1]
        sent = np.maximum(sent, 0)
    targets = sent[:, 1:]
    return (np.ascontiguousarray(inputs, np.int32), np.ascontiguousarray(targets, np.int32),
            np.ascontiguousarray(masks, np.float32), lens)


class Decoder:
    """Device decoder: parameters, gradients and native step calls."""

    def __init__(self, spec: DecoderSpec, params: dict | None = None, device='cuda:0', seed=0):
        import torch
        self.torch = torch
        self.lib = L.load()
        self.spec, self.device = spec, device
        self.params = FlatParams(spec.param_shapes(), device, status_tail=True)
        self.params.load(params if params is not None else init_params(spec, seed))
        self.grads = self.params.like()
        self._ws = None
        self._ws_bytes = 0
        self._dropout_calls = 0
        self.set_dropout_stream(seed, 0)
        self._ctx = {}

    # ------------------------------------------------------------------ helpers --------
    def _workspace(self, nbytes):
        if self._ws is None or self._ws_bytes < nbytes:
            self._ws = self.torch.empty(int(nbytes), dtype=self.torch.uint8, device=self.device)
            self._ws_bytes = int(nbytes)
        return self._ws

    def _dev(self, a, dtype=None):
        t = self.torch.from_numpy(np.ascontiguousarray(a))
        if dtype is not None:
            t = t.to(dtype)
        return t.to(self.device, non_blocking=False)

    def make_masks(self, B, Tp, seed):
        """Bernoulli keep masks generated on the device (counter-based; TF's RNG stream is
        not reproducible, SURVEY §7 'Dropout parity')."""
        torch, s = self.torch, self.spec
        EA = s.E + s.A
        out = {}
        off = 0
        for name, shape, keep in (('init_in', (B, EA), 1 - s.dropout_rnn_in), ('inp', (Tp, B, EA), 1 - s.dropout_rnn_in),
                                  ('out', (Tp, B, s.D), 1 - s.dropout_rnn_out),
                                  ('alpha', (Tp, B, s.H, s.M), s.attn_keep_prob)):
            t = torch.empty(shape, dtype=torch.float32, device=self.device)
            L.check(self.lib.comic_dropout_mask(t.data_ptr(), t.numel(), keep, int(seed), off, L.stream_ptr()),
                    'dropout_mask')
            off += t.numel()
            out[name] = t
        return out

    # ------------------------------------------------------------------ training -------
    def _train_ctx(self, B, T, Tp, training, gen_masks, want_input_grads):
        """Persistent device buffers (+ hipGraph) of one training-step shape."""
        key = (B, T, Tp, bool(training), bool(gen_masks), bool(want_input_grads))
        ctx = self._ctx.get(key)
        if ctx is not None:
            return ctx
        torch, s, dev = self.torch, self.spec, self.device
        from types import SimpleNamespace
        ctx = SimpleNamespace(key=key, calls=0, graph=None)
        f32 = dict(dtype=torch.float32, device=dev)
        # ONE staging block per step: [seed int64 | inputs, targets, lens int32 | wmask, coef, row scale fp32]
        n_i32, n_f32 = 2 * B * T + B, 3 * B * T
        o_i32, o_f32 = 8, 8 + 4 * ((n_i32 + 1) // 2 * 2)
        nbytes = o_f32 + 4 * n_f32

        def views(buf):
            return (buf[:8].view(torch.int64), buf[o_i32:o_i32 + 4 * n_i32].view(torch.int32),
                    buf[o_f32:o_f32 + 4 * n_f32].view(torch.float32))
        ctx.stage_dev = torch.zeros(nbytes, dtype=torch.uint8, device=dev)
        ctx.seed, ctx.i32, ctx.f32 = views(ctx.stage_dev)
        # pinned staging, two slots used alternately: the host may run ahead of the GPU, and an async H2D copy
        # reads its pinned source when the STREAM gets there, so a slot is rewritten only after the copy that last
        # used it has executed (event per slot)
        ctx.stage = []
        for _ in range(2):
            buf = torch.zeros(nbytes, dtype=torch.uint8).pin_memory()
            sd, i32, f32v = views(buf)
            ctx.stage.append(SimpleNamespace(buf=buf, seed=sd, i32=i32, f32=f32v, copied=None))
        ctx.fm = torch.empty((B, s.M, s.C), **f32)
        ctx.im = torch.empty((B, s.Cg), **f32)
        EA = s.E + s.A
        ctx.masks = None
        if training:
            # one buffer, four views (generated by ONE launch: comic_dropout_masks4_dev)
            shapes = dict(init_in=(B, EA), inp=(Tp, B, EA), out=(Tp, B, s.D), alpha=(Tp, B, s.H, s.M))
            sizes = [int(np.prod(v)) for v in shapes.values()]
            ctx.mask_buf = torch.empty(sum(sizes), **f32)
            ctx.masks, off = {}, 0
            for (k, shp), n in zip(shapes.items(), sizes):
                ctx.masks[k] = ctx.mask_buf[off:off + n].view(shp)
                off += n
            ctx.mask_n4 = (C.c_int64 * 4)(*sizes)
            ctx.mask_keep4 = (C.c_float * 4)(1 - s.dropout_rnn_in, 1 - s.dropout_rnn_in, 1 - s.dropout_rnn_out,
                                             s.attn_keep_prob)
        ctx.logits = torch.empty((T, B, s.V), **f32)
        ctx.ids = torch.empty((T, B), dtype=torch.int32, device=dev)
        ctx.hist = torch.empty((Tp, B, s.H, s.M), **f32)
        ctx.loss_rows = torch.empty(T * B, **f32)
        ctx.map_loss = torch.zeros(1, **f32)
        ctx.loss = torch.zeros(1, **f32)
        ctx.dfm = torch.empty((B, s.M, s.C), **f32) if want_input_grads else None
        ctx.dim = torch.empty((B, s.Cg), **f32) if want_input_grads else None
        ctx.desc = s.desc(training)
        ctx.nbytes = self.lib.comic_decoder_train_workspace(C.byref(ctx.desc), B, T)
        ctx.ws = torch.empty(int(ctx.nbytes), dtype=torch.uint8, device=dev)
        self._ctx[key] = ctx
        return ctx

    def voided_steps(self):
        """Training steps the device voided so far (a bounded wait of a persistent loop expired: NaN loss, no update;
        comic_decoder_params::status of the parameter table).  Synchronises: for log points, not for every step."""
        return int(float(self.params.status))

    def set_dropout_stream(self, seed, rank=0):
        """Base of the per-step dropout seeds: a function of the run's seed (tf.set_random_seed(rand_seed), train_fn.py:35)
        and of the data-parallel rank, so that runs 1/2/3 and the ranks of one run draw different masks."""
        x = (int(seed) & 0xFFFFFFFF) * 0x9E3779B97F4A7C15 + (int(rank) + 1) * 0xBF58476D1CE4E5B9
        x ^= x >> 31
        self._dropout_base = (x * 0x94D049BB133111EB) & 0x3FFFFFFFFFFFFFFF

    def _train_device(self, ctx, phase=None):
        """Device-only part of a training step (no host sync, no host memcpy): capturable.
        phase 'fwd' / 'bwd': the forward to the logits / the loss and the backward (COMIC_DEC_PHASE_*)."""
        torch, s = self.torch, self.spec
        B, T, Tp, training, gen_masks, _ = ctx.key
        st = L.stream_ptr()
        m = ctx.masks or {}
        if training and gen_masks and phase != 'bwd':
            L.check(self.lib.comic_dropout_masks4_dev(ctx.mask_buf.data_ptr(), ctx.mask_n4, ctx.mask_keep4,
                                                      ctx.seed.data_ptr(), st), 'dropout_masks4')
            if s.recurrent_dropout:
                # variational recurrent dropout (model_base.py:645; [TF-1.9] DropoutWrapper draws its input / output
                # noise once per run with a batch dimension of 1): the first row of the drawn masks serves every batch
                # row and time step, the init call through the same wrapper included; the attention dropout is not
                # part of the wrapper and stays per step
                row_in = m['inp'][0, 0].clone()
                row_out = m['out'][0, 0].clone()
                m['inp'].copy_(row_in.expand_as(m['inp']))
                m['init_in'].copy_(row_in.expand_as(m['init_in']))
                m['out'].copy_(row_out.expand_as(m['out']))
        BT = B * T
        i32, f32 = ctx.i32, ctx.f32
        ptab, gtab = self.params.table(), self.grads.table()
        ctx.desc.flags = (L.decoder_flags_from_env() | {None: 0, 'fwd': L.DEC_PHASE_FWD, 'bwd': L.DEC_PHASE_BWD}[phase]
                          | (L.DEC_INJECT_TIMEOUT if ctx.__dict__.pop('inject_timeout', False) else 0))
        L.check(self.lib.comic_decoder_train_step(
            C.byref(ctx.desc), C.byref(ptab), C.byref(gtab), ctx.fm_in.data_ptr(), ctx.im_in.data_ptr(),
            i32.data_ptr(), i32.data_ptr() + 4 * BT, f32.data_ptr(), f32.data_ptr() + 4 * BT,
            i32.data_ptr() + 8 * BT, B, T, Tp,
            L.ptr(m.get('init_in')), L.ptr(m.get('inp')), L.ptr(m.get('out')), L.ptr(m.get('alpha')),
            ctx.logits.data_ptr(), ctx.ids.data_ptr(), ctx.hist.data_ptr(), ctx.loss_rows.data_ptr(),
            ctx.map_loss.data_ptr(), L.ptr(ctx.dfm), L.ptr(ctx.dim), ctx.ws.data_ptr(), ctx.nbytes, st),
            'decoder_train_step')
        if phase == 'fwd':
            return
        # sequence_loss reduction (model_base.py:337-347): rows carry xent*w; rs = 1/denominator (* reward/B)
        L.check(self.lib.comic_weighted_sum_tb(ctx.loss_rows.data_ptr(), f32.data_ptr() + 4 * 2 * BT, T, B,
                                               ctx.loss.data_ptr(), st), 'weighted_sum_tb')

    def train_step(self, fm, im_embed, captions, masks=None, rewards=None, training=True, seed=None,
                   want_input_grads=False, xe_denom=None, use_graph=False, on_inputs_consumed=None, dp=None,
                   copy_inputs=True, phase=None, inject_timeout=False):
        """One teacher-forced forward + backward.  `captions` [B,L] int (PAD = -1).
        phase: None = the whole step.  'fwd' = everything no loss coefficient enters (forward to the logits; `rewards` is
        ignored) and 'bwd' = the rest, with the SAME captions and now the rewards: the SCST step enqueues 'fwd' as soon as
        the rollouts are back and scores them on the host while it runs (the same kernels in the same order: same bits).
        masks: None -> generated on device when training; dict(init_in, inp, out, alpha) of
        device tensors or numpy arrays -> injected (parity tests).  rewards [B] -> SCST loss
        mean_b(xent_b * reward_b) (model_base.py:342-347).
        xe_denom: override of the XE normaliser sum(w)+1e-12 (data parallel: global token count / world size,
        so that the rank-mean of the gradients equals the single-process gradient of the global batch).
        dp: a trainer.DataParallel with world > 1 -> the same normaliser formed ON THE DEVICE: the token count of
        this rank's batch is summed from the staged mask, all-reduced on the stream, and the per-token coefficients
        the kernels read are rescaled in place -- no host synchronisation in the step.
        copy_inputs=False: the step reads `fm` / `im_embed` in place instead of from its own copies (13 MB less
        device-to-device traffic per step at batch 64).  The caller then keeps both tensors unchanged until the step has
        EXECUTED (stream order: anything it enqueues on this stream afterwards is safe) and `on_inputs_consumed` is
        called after the step's launches rather than before them.  Ignored with use_graph (a graph holds addresses).
        use_graph: replay the step from a hipGraph captured per (B, T, T') shape (the second call with
        a shape captures it) -- removes the ~450 host launches of a step from the critical path.
        Returns dict(loss, map_loss, logits [B,T,V], ids [B,T], attn_maps [B,H,T',M]) (device views
        of persistent buffers: valid until the next call with the same shape)."""
        torch, s = self.torch, self.spec
        inputs, targets, wmask, lens = process_inputs(captions, s.token_type)
        B, T = inputs.shape
        Tp = int(lens.max())
        if rewards is None:
            den_xe = np.float32(xe_denom) if xe_denom is not None else np.float32(wmask.sum() + np.float32(1e-12))
            rs = np.full((B, T), np.float32(1.0) / den_xe, np.float32)
            coef = wmask / den_xe
        else:
            den = wmask.sum(axis=1, keepdims=True) + np.float32(1e-12)
            rb = (np.asarray(rewards, np.float32)[:, None] / np.float32(B))
            rs = np.broadcast_to(rb / den, (B, T)).astype(np.float32)
            coef = wmask / den * rb
        gen_masks = bool(training and masks is None)
        use_masks = bool(training or masks is not None)
        ctx = self._train_ctx(B, T, Tp, use_masks, gen_masks, want_input_grads)
        BT = B * T
        if phase == 'bwd':          # second call of a split step: only the coefficients are new
            assert getattr(ctx, 'split_open', False), 'train_step(phase="bwd") without its phase="fwd" call'
            ctx.split_open = False
            slot = ctx.stage[(ctx.calls - 1) % 2]                 # the slot its forward call staged: same inputs
            slot.copied.synchronize()
            fh = slot.f32.numpy()
            fh[BT:2 * BT] = coef.reshape(-1); fh[2 * BT:] = rs.reshape(-1)
            ctx.stage_dev.copy_(slot.buf, non_blocking=True)
            slot.copied.record(torch.cuda.current_stream())
            self._run_train_phase(ctx, 'bwd', use_graph)
            return dict(loss=ctx.loss[0], map_loss=ctx.map_loss[0], logits=ctx.logits.permute(1, 0, 2), ids=ctx.ids.t(),
                        attn_maps=ctx.hist.permute(1, 2, 0, 3), dfm=ctx.dfm, dim_embed=ctx.dim, Tp=Tp)
        slot = ctx.stage[ctx.calls % 2]
        if slot.copied is not None:
            slot.copied.synchronize()
        ih, fh = slot.i32.numpy(), slot.f32.numpy()
        ih[:BT] = inputs.reshape(-1); ih[BT:2 * BT] = targets.reshape(-1); ih[2 * BT:] = lens
        fh[:BT] = wmask.reshape(-1); fh[BT:2 * BT] = coef.reshape(-1); fh[2 * BT:] = rs.reshape(-1)
        if gen_masks:
            if seed is None:
                self._dropout_calls += 1
                seed = (self._dropout_base + self._dropout_calls) & 0x7FFFFFFFFFFFFFFF
            slot.seed[0] = int(seed)
        ctx.stage_dev.copy_(slot.buf, non_blocking=True)
        if masks is not None:
            for k, v in masks.items():
                ctx.masks[k].copy_(v if torch.is_tensor(v) else torch.from_numpy(np.ascontiguousarray(v, np.float32)))
        if dp is not None and dp.world > 1 and rewards is None:
            f = ctx.f32                                  # device views [wmask | coef | row scale]
            inv = 1.0 / dp.global_xe_denominator(f[:BT])
            torch.mul(f[:BT], inv, out=f[BT:2 * BT])
            f[2 * BT:3 * BT] = inv
        if slot.copied is None:
            slot.copied = torch.cuda.Event()
        slot.copied.record(torch.cuda.current_stream())
        assert fm.shape == (B, s.M, s.C) and im_embed.shape == (B, s.Cg), (fm.shape, im_embed.shape)
        assert fm.dtype == torch.float32 and im_embed.dtype == torch.float32
        direct = (not copy_inputs and not use_graph and fm.is_contiguous() and im_embed.is_contiguous())
        if direct:
            ctx.fm_in, ctx.im_in = fm, im_embed
        else:
            ctx.fm.copy_(fm)
            ctx.im.copy_(im_embed)
            ctx.fm_in, ctx.im_in = ctx.fm, ctx.im
            if on_inputs_consumed is not None:      # the encoder buffers may be overwritten from here on
                on_inputs_consumed()
        if phase == 'fwd':
            ctx.split_open = True
            self._run_train_phase(ctx, 'fwd', use_graph)
            if direct and on_inputs_consumed is not None:
                on_inputs_consumed()
            ctx.calls += 1
            return None
        if inject_timeout:          # fault injection of THIS call (COMIC_DEC_INJECT_TIMEOUT, tests): eager launches only
            assert not use_graph and phase is None
            ctx.inject_timeout = True
        if use_graph and ctx.graph is None and ctx.calls >= 1:
            g = torch.cuda.CUDAGraph()
            with torch.cuda.graph(g, capture_error_mode='thread_local'):
                self._train_device(ctx)
            ctx.graph = g
        if use_graph and ctx.graph is not None:
            ctx.graph.replay()
        else:
            self._train_device(ctx)
        if direct and on_inputs_consumed is not None:   # stream order: whatever the caller enqueues now runs after the step
            on_inputs_consumed()
        ctx.calls += 1
        return dict(loss=ctx.loss[0], map_loss=ctx.map_loss[0], logits=ctx.logits.permute(1, 0, 2), ids=ctx.ids.t(),
                    attn_maps=ctx.hist.permute(1, 2, 0, 3), dfm=ctx.dfm, dim_embed=ctx.dim, Tp=Tp)

    def _run_train_phase(self, ctx, phase, use_graph):
        """One half of a split step, eagerly or from its own hipGraph (captured at the second use of the shape)."""
        torch = self.torch
        graphs = ctx.__dict__.setdefault('phase_graphs', {})
        used = ctx.__dict__.setdefault('phase_calls', {})
        used[phase] = used.get(phase, 0) + 1
        if use_graph and phase not in graphs and used[phase] >= 2:
            g = torch.cuda.CUDAGraph()
            with torch.cuda.graph(g, capture_error_mode='thread_local'):
                self._train_device(ctx, phase)
            graphs[phase] = g
        if use_graph and phase in graphs:
            graphs[phase].replay()
        else:
            self._train_device(ctx, phase)

    # ------------------------------------------------------------------ decoding -------
    def max_iterations(self, infer_max_length, vocab_len):
        """model_base.py:708-714."""
        it = infer_max_length
        if self.spec.token_type == 'radix':
            it *= len(number_to_base(vocab_len, self.spec.start_id))    # start_id == radix_base
        elif self.spec.token_type == 'char':
            it *= 5
        return it

    def _infer_ctx(self, kind, B, W, max_steps, want_logits, fm, im_embed, slot=0):
        """Persistent buffers (+ a hipGraph of the whole decode loop, captured on the second call
        with the same shape) of greedy / beam decoding: the executors run all `max_steps` steps on
        the device without host synchronisation, so one graph launch replaces ~8 kernel launches
        per step of host work."""
        torch, s = self.torch, self.spec
        key = (kind, B, W, max_steps, bool(want_logits), int(slot))      # slot: independent buffer sets (decodes in flight on several streams)
        ctxs = self.__dict__.setdefault('_infer_ctxs', {})
        ctx = ctxs.get(key)
        if ctx is None:
            ctx = type('InferCtx', (), {})()
            R = B * W
            i32 = dict(dtype=torch.int32, device=self.device)
            f32 = dict(dtype=torch.float32, device=self.device)
            ctx.fm = torch.empty((B, s.M, s.C), **f32)
            ctx.im = torch.empty((B, s.Cg), **f32)
            ctx.hist = torch.empty((max_steps, R, s.H * s.M), **f32)
            if kind == 'greedy':
                ctx.ids = torch.empty((max_steps, B), **i32)
                ctx.logits = torch.empty((max_steps, B, s.V), **f32) if want_logits else None
                ctx.first_eos = torch.empty(B, **i32)
            else:
                # rows past the executed steps are never written (device-side early exit): poison them so a
                # kernel that indexes through them fails the same way every time
                ctx.step_ids = torch.full((max_steps, B, W), 0x7f7f7f7f, **i32)
                ctx.parent_ids = torch.full((max_steps, B, W), 0x7f7f7f7f, **i32)
                ctx.scores = torch.empty((max_steps, B, W), **f32)
                ctx.lengths = torch.empty((B, W), dtype=torch.int64, device=self.device)
                ctx.finished = torch.empty((B, W), **i32)
                ctx.steps = torch.empty(1, **i32)
            ctx.desc = s.desc(False)
            ctx.nbytes = int(self.lib.comic_decoder_infer_workspace(C.byref(ctx.desc), R, max_steps))
            ctx.ws = torch.empty(ctx.nbytes, dtype=torch.uint8, device=self.device)   # own workspace: graph-stable
            ctx.ptab = self.params.table()
            ctx.graph, ctx.calls = None, 0
            ctxs[key] = ctx
        ctx.fm.copy_(fm.reshape(ctx.fm.shape))
        ctx.im.copy_(im_embed.reshape(ctx.im.shape))
        return ctx

    def _run_infer(self, ctx, launch, use_graph):
        torch = self.torch
        if use_graph and ctx.graph is None and ctx.calls >= 1:
            g = torch.cuda.CUDAGraph()
            with torch.cuda.graph(g, capture_error_mode='thread_local'):
                launch()
            ctx.graph = g
        if use_graph and ctx.graph is not None:
            ctx.graph.replay()
        else:
            launch()
        ctx.calls += 1

    def greedy(self, fm, im_embed, max_steps, want_logits=False, use_graph=True, defer=False):
        """rnn_decoder_search(greedy) (ops_rnn.py:115-180).  -> ids [B,T_exec] (numpy int32),
        attn_maps [B,H,T_exec,M] (device), logits [B,T_exec,V] or None.
        defer: only enqueue the loop and return a function that fetches the result (the SCST step converts the beam
        rollouts to text on the host while this loop runs)."""
        torch, s = self.torch, self.spec
        B = fm.shape[0]
        ctx = self._infer_ctx('greedy', B, 1, max_steps, want_logits, fm, im_embed)

        def launch():
            ctx.desc.flags = L.decoder_flags_from_env()
            L.check(self.lib.comic_decoder_greedy(C.byref(ctx.desc), C.byref(ctx.ptab), ctx.fm.data_ptr(),
                                                  ctx.im.data_ptr(), B, max_steps, ctx.ids.data_ptr(),
                                                  L.ptr(ctx.logits), ctx.hist.data_ptr(), ctx.first_eos.data_ptr(),
                                                  ctx.ws.data_ptr(), ctx.nbytes, L.stream_ptr()), 'decoder_greedy')
        self._run_infer(ctx, launch, use_graph)
        if defer:
            # the ids leave through pinned memory behind an event of their own (as beam_search_ids): the fetch then waits for
            # the rollout, not for whatever the caller has enqueued behind it in the meantime (the SCST step enqueues the
            # update's forward pass before it looks at the greedy captions)
            if getattr(ctx, 'ids_host', None) is None:
                ctx.ids_host = torch.empty((max_steps, B), dtype=torch.int32).pin_memory()
                ctx.eos_host = torch.empty(B, dtype=torch.int32).pin_memory()
                ctx.fetched = torch.cuda.Event()
            ctx.ids_host.copy_(ctx.ids, non_blocking=True)
            ctx.eos_host.copy_(ctx.first_eos, non_blocking=True)
            ctx.fetched.record(torch.cuda.current_stream())

        def fetch():
            if defer:
                ctx.fetched.synchronize()
                fe = ctx.eos_host.numpy()
            else:
                fe = ctx.first_eos.cpu().numpy()
            if fe.min() < 0:                              # comic_persist_check_greedy: a bounded wait of the loop expired
                raise L.ComicHipError('greedy: the persistent decode loop did not complete (a wait on another workgroup timed out)')
            t_exec = int(min(max_steps, fe.max() + 1))   # loop ends when every row has emitted EOS
            if defer:
                out_ids = np.ascontiguousarray(ctx.ids_host[:t_exec].numpy().T)
            else:
                out_ids = ctx.ids[:t_exec].t().contiguous().cpu().numpy()
            hist = ctx.hist[:t_exec].reshape(t_exec, B, s.H, s.M).permute(1, 2, 0, 3).clone()
            return out_ids, hist, (ctx.logits[:t_exec].permute(1, 0, 2).clone() if want_logits else None)
        return fetch if defer else fetch()

    def sample(self, fm, im_embed, max_steps, seed=0, noise=None, want_logits=False):
        """rnn_decoder_search(greedy_search=False) (ops_rnn.py:158-166; ModelBase._rnn_dynamic_decoder(sample=True),
        model_base.py:716-726): SampleEmbeddingHelper draws every next token from Categorical(logits).  The draw is
        argmax(logits + Gumbel noise); `noise` [max_steps,B,V] (device fp32) may be supplied, else it is generated from
        `seed` (torch's generator: the stream is not TensorFlow's Philox, only the distribution is the reference's).
        -> ids [B,T_exec], attn_maps [B,H,T_exec,M], logits [B,T_exec,V] or None, as greedy()."""
        torch, s = self.torch, self.spec
        B = fm.shape[0]
        ctx = self._infer_ctx('greedy', B, 1, max_steps, want_logits, fm, im_embed)
        if noise is None:
            gen = torch.Generator(device=self.device)
            gen.manual_seed(int(seed))
            u = torch.rand((max_steps, B, s.V), generator=gen, device=self.device, dtype=torch.float32)
            noise = -torch.log(-torch.log(u.clamp_(1e-20, 1.0 - 1e-7)))
        assert tuple(noise.shape) == (max_steps, B, s.V) and noise.dtype == torch.float32 and noise.is_contiguous()
        ctx.desc.flags = L.decoder_flags_from_env()
        L.check(self.lib.comic_decoder_sample(C.byref(ctx.desc), C.byref(ctx.ptab), ctx.fm.data_ptr(), ctx.im.data_ptr(), B,
                                              max_steps, noise.data_ptr(), ctx.ids.data_ptr(), L.ptr(ctx.logits),
                                              ctx.hist.data_ptr(), ctx.first_eos.data_ptr(), ctx.ws.data_ptr(), ctx.nbytes,
                                              L.stream_ptr()), 'decoder_sample')
        fe = ctx.first_eos.cpu().numpy()
        t_exec = int(min(max_steps, fe.max() + 1))
        out_ids = ctx.ids[:t_exec].t().contiguous().cpu().numpy()
        hist = ctx.hist[:t_exec].reshape(t_exec, B, s.H, s.M).permute(1, 2, 0, 3).clone()
        return out_ids, hist, (ctx.logits[:t_exec].permute(1, 0, 2).clone() if want_logits else None)

    def beam_search_ids(self, fm, im_embed, beam, max_steps, use_graph=True, slot=0):
        """Beam search for its predicted ids alone, fetched WITHOUT draining the stream: the loop, gather_tree over all
        max_steps rows (rows past the executed steps come out as end_id) and the copies to pinned memory are enqueued, an
        event marks their end, and the returned function waits for that event only -- work enqueued behind it (the greedy
        rollout of the SCST step) keeps running while the host turns these ids into text.  -> fetch() -> [T,B,W] int32,
        the same values as beam_search()['predicted_ids']."""
        torch, s = self.torch, self.spec
        B, W = fm.shape[0], beam
        ctx = self._infer_ctx('beam', B, W, max_steps, False, fm, im_embed, slot=slot)
        ctx.desc.length_penalty_weight = 0.0

        def launch():
            ctx.desc.flags = L.decoder_flags_from_env()
            L.check(self.lib.comic_decoder_beam(C.byref(ctx.desc), C.byref(ctx.ptab), ctx.fm.data_ptr(),
                                                ctx.im.data_ptr(), B, W, max_steps, ctx.step_ids.data_ptr(),
                                                ctx.parent_ids.data_ptr(), ctx.scores.data_ptr(),
                                                ctx.lengths.data_ptr(), ctx.finished.data_ptr(), ctx.hist.data_ptr(),
                                                ctx.steps.data_ptr(), ctx.ws.data_ptr(), ctx.nbytes, L.stream_ptr()),
                    'decoder_beam')
        self._run_infer(ctx, launch, use_graph)
        if getattr(ctx, 'pred', None) is None:
            ctx.pred = torch.empty((max_steps, B, W), dtype=torch.int32, device=self.device)
            ctx.pred_host = torch.empty((max_steps, B, W), dtype=torch.int32).pin_memory()
            ctx.steps_host = torch.empty(1, dtype=torch.int32).pin_memory()
            ctx.fetched = torch.cuda.Event()
        max_len = ctx.lengths.max(dim=1).values.to(torch.int32).contiguous()
        L.check(self.lib.comic_gather_tree(ctx.step_ids.data_ptr(), ctx.parent_ids.data_ptr(), max_len.data_ptr(),
                                           ctx.pred.data_ptr(), max_steps, B, W, s.end_id, L.stream_ptr()), 'gather_tree')
        ctx.pred_host.copy_(ctx.pred, non_blocking=True)
        ctx.steps_host.copy_(ctx.steps, non_blocking=True)
        ctx.fetched.record(torch.cuda.current_stream())
        ctx.keep = max_len                                # (alive until the copies have run)

        def fetch():
            ctx.fetched.synchronize()
            return ctx.pred_host[:int(ctx.steps_host[0])].numpy().copy()
        return fetch

    def beam_search(self, fm, im_embed, beam, max_steps, want_attention=True, use_graph=True, length_penalty_weight=0.0):
        """rnn_decoder_beam_search (ops_rnn.py:49-112).  Returns predicted_ids [T,B,W] (after
        gather_tree), scores [T,B,W] (with length_penalty_weight != 0: the penalised scores the beams were ranked by,
        BeamSearchDecoder's `scores` output), the raw step/parent ids and, unless want_attention=False, the
        beam-sorted alignment history [T,B*W,H*M] (numpy; BeamSearchDecoderMultiHead,
        ops_rnn.py:807-845 -- host post-processing that only visualisation needs)."""
        torch, s = self.torch, self.spec
        B, W = fm.shape[0], beam
        # (a non-zero length penalty is another captured graph: the weight is baked into the step kernel's arguments)
        ctx = self._infer_ctx('beam' if not length_penalty_weight else 'beam_lp%r' % float(length_penalty_weight), B, W,
                              max_steps, False, fm, im_embed)
        ctx.desc.length_penalty_weight = float(length_penalty_weight)

        def launch():
            ctx.desc.flags = L.decoder_flags_from_env()
            L.check(self.lib.comic_decoder_beam(C.byref(ctx.desc), C.byref(ctx.ptab), ctx.fm.data_ptr(),
                                                ctx.im.data_ptr(), B, W, max_steps, ctx.step_ids.data_ptr(),
                                                ctx.parent_ids.data_ptr(), ctx.scores.data_ptr(),
                                                ctx.lengths.data_ptr(), ctx.finished.data_ptr(), ctx.hist.data_ptr(),
                                                ctx.steps.data_ptr(), ctx.ws.data_ptr(), ctx.nbytes, L.stream_ptr()),
                    'decoder_beam')
        self._run_infer(ctx, launch, use_graph)
        T = int(ctx.steps.item())
        max_len = ctx.lengths.max(dim=1).values.to(torch.int32).contiguous()
        pred = torch.empty((T, B, W), dtype=torch.int32, device=self.device)
        L.check(self.lib.comic_gather_tree(ctx.step_ids[:T].data_ptr(), ctx.parent_ids[:T].data_ptr(),
                                           max_len.data_ptr(), pred.data_ptr(), T, B, W, s.end_id, L.stream_ptr()),
                'gather_tree')
        par = ctx.parent_ids[:T].cpu().numpy()
        ln = ctx.lengths.cpu().numpy()
        out = dict(predicted_ids=pred.cpu().numpy(), scores=ctx.scores[:T].cpu().numpy(),
                   step_ids=ctx.step_ids[:T].cpu().numpy(), parent_ids=par, lengths=ln)
        if want_attention:
            out['attn_hist'] = gather_tree_from_array(ctx.hist[:T].cpu().numpy(), par, ln, s.end_id)
        return out


def _gather_tree_host(step_ids, parent_ids, max_len, end_token):
    T, B, W = parent_ids.shape
    beams = np.full((T, B, W), end_token, np.int32)
    for b in range(B):
        Lb = min(T, int(max_len[b]))
        if Lb <= 0:
            continue
        for w in range(W):
            beams[Lb - 1, b, w] = step_ids[Lb - 1, b, w]
            parent = parent_ids[Lb - 1, b, w]
            for level in range(Lb - 2, -1, -1):
                beams[level, b, w] = step_ids[level, b, parent]
                parent = parent_ids[level, b, parent]
            fin = False
            for t in range(Lb):
                if fin:
                    beams[t, b, w] = end_token
                elif beams[t, b, w] == end_token:
                    fin = True
    return beams


def gather_tree_from_array(t, parent_ids, sequence_length, _unused_end=None):
    """Beam-sort a per-step state array [T, B*W, S] (BeamSearchDecoderMultiHead.
    _maybe_sort_array_beams, ops_rnn.py:807-845 -> [TF-1.9] gather_tree_from_array).
    Host post-processing of the alignment history (small, index-only)."""
    T, B, W = parent_ids.shape
    beam_ids = np.tile(np.arange(W, dtype=np.int32)[None, None, :], (T, B, 1))
    mask = (np.arange(T)[None, None, :] < np.asarray(sequence_length)[:, :, None]).transpose(2, 0, 1)
    masked = np.where(mask, beam_ids, W + 1).astype(np.int32)
    max_len = np.asarray(sequence_length).max(axis=1)
    sorted_ids = _gather_tree_host(masked, parent_ids, max_len, W + 1)
    sorted_ids = np.where(mask, sorted_ids, beam_ids)
    src = np.asarray(t).reshape(T, B, W, -1)
    return src[np.arange(T)[:, None, None], np.arange(B)[None, :, None], sorted_ids].reshape(np.asarray(t).shape)
