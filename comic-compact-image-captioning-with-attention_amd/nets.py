"""CNN encoder plans (table-driven) and the device-side encoder object.

Counterpart of `nets_factory.get_network_fn(cnn_name, num_classes=None, is_training=False)`
(reference common/nets/nets_factory.py:116-159) + `ModelBase._encoder`
(src/model_base.py:56-104) for the backbone on the hot path: InceptionV3
(common/nets/inception_v3.py:100-415, head :520-532).  The network is described as a flat
list of `comic_cnn_op` records over a buffer table; branch outputs are written straight
into channel slices of the block's output buffer (tf.concat without a copy) and the whole
plan is executed by one native call (`comic_cnn_forward`).

Variable names follow the slim checkpoint (`InceptionV3/<scope>/weights`,
`.../BatchNorm/{beta,moving_mean,moving_variance}`) so checkpoints map 1:1.
"""
from __future__ import annotations

import ctypes as C
import json
import math
import os

import numpy as np

from . import _lib as L
from . import streams

BN_EPS = 1e-3            # inception_utils.py:36

# --- InceptionV3 topology as data ------------------------------------------------------
# op forms: ('c', scope, cout, (kh,kw), stride, 'SAME'|'VALID') | ('avg',) | ('max',)
#           | ('fork', [op, op])  -> two convs on the same input, outputs concatenated
_STEM = [('c', 'Conv2d_1a_3x3', 32, (3, 3), 2, 'VALID'), ('c', 'Conv2d_2a_3x3', 32, (3, 3), 1, 'VALID'),
         ('c', 'Conv2d_2b_3x3', 64, (3, 3), 1, 'SAME'), ('max', 'MaxPool_3a_3x3'),
         ('c', 'Conv2d_3b_1x1', 80, (1, 1), 1, 'VALID'), ('c', 'Conv2d_4a_3x3', 192, (3, 3), 1, 'VALID'),
         ('max', 'MaxPool_5a_3x3')]


def _c(scope, cout, k, stride=1, pad='SAME'):
    return ('c', scope, cout, k if isinstance(k, tuple) else (k, k), stride, pad)


def _mixed5(pool_c, b1):
    return [[_c('Conv2d_0a_1x1', 64, 1)],
            [_c(b1[0], 48, 1), _c(b1[1], 64, 5)],
            [_c('Conv2d_0a_1x1', 64, 1), _c('Conv2d_0b_3x3', 96, 3), _c('Conv2d_0c_3x3', 96, 3)],
            [('avg',), _c('Conv2d_0b_1x1', pool_c, 1)]]


def _mixed6(d):
    return [[_c('Conv2d_0a_1x1', 192, 1)],
            [_c('Conv2d_0a_1x1', d, 1), _c('Conv2d_0b_1x7', d, (1, 7)), _c('Conv2d_0c_7x1', 192, (7, 1))],
            [_c('Conv2d_0a_1x1', d, 1), _c('Conv2d_0b_7x1', d, (7, 1)), _c('Conv2d_0c_1x7', d, (1, 7)),
             _c('Conv2d_0d_7x1', d, (7, 1)), _c('Conv2d_0e_1x7', 192, (1, 7))],
            [('avg',), _c('Conv2d_0b_1x1', 192, 1)]]


def _mixed7(b1b):
    return [[_c('Conv2d_0a_1x1', 320, 1)],
            [_c('Conv2d_0a_1x1', 384, 1), ('fork', [_c('Conv2d_0b_1x3', 384, (1, 3)), _c(b1b, 384, (3, 1))])],
            [_c('Conv2d_0a_1x1', 448, 1), _c('Conv2d_0b_3x3', 384, 3),
             ('fork', [_c('Conv2d_0c_1x3', 384, (1, 3)), _c('Conv2d_0d_3x1', 384, (3, 1))])],
            [('avg',), _c('Conv2d_0b_1x1', 192, 1)]]


INCEPTION_V3_BLOCKS = [
    ('Mixed_5b', _mixed5(32, ('Conv2d_0a_1x1', 'Conv2d_0b_5x5'))),
    ('Mixed_5c', _mixed5(64, ('Conv2d_0b_1x1', 'Conv_1_0c_5x5'))),      # reference scope-name quirk
    ('Mixed_5d', _mixed5(64, ('Conv2d_0a_1x1', 'Conv2d_0b_5x5'))),
    ('Mixed_6a', [[_c('Conv2d_1a_1x1', 384, 3, 2, 'VALID')],
                  [_c('Conv2d_0a_1x1', 64, 1), _c('Conv2d_0b_3x3', 96, 3), _c('Conv2d_1a_1x1', 96, 3, 2, 'VALID')],
                  [('max',)]]),
    ('Mixed_6b', _mixed6(128)), ('Mixed_6c', _mixed6(160)), ('Mixed_6d', _mixed6(160)), ('Mixed_6e', _mixed6(192)),
    ('Mixed_7a', [[_c('Conv2d_0a_1x1', 192, 1), _c('Conv2d_1a_3x3', 320, 3, 2, 'VALID')],
                  [_c('Conv2d_0a_1x1', 192, 1), _c('Conv2d_0b_1x7', 192, (1, 7)), _c('Conv2d_0c_7x1', 192, (7, 1)),
                   _c('Conv2d_1a_3x3', 192, 3, 2, 'VALID')],
                  [('max',)]]),
    ('Mixed_7b', _mixed7('Conv2d_0b_3x1')), ('Mixed_7c', _mixed7('Conv2d_0c_3x1')),
]


def _v1(c0, c1a, c1b, c2a, c2b, c3, quirk=False):
    """Inception-V1 block (inception_v1.py:95-265): 1x1 | 1x1->3x3 | 1x1->3x3 | maxpool(3x3,s1,SAME)->1x1."""
    return [[_c('Conv2d_0a_1x1', c0, 1)],
            [_c('Conv2d_0a_1x1', c1a, 1), _c('Conv2d_0b_3x3', c1b, 3)],
            [_c('Conv2d_0a_1x1', c2a, 1), _c('Conv2d_0a_3x3' if quirk else 'Conv2d_0b_3x3', c2b, 3)],
            [('maxs1',), _c('Conv2d_0b_1x1', c3, 1)]]


# ('pool', name, k) entries between blocks are SAME-padded stride-2 max pools
INCEPTION_V1_STEM = [('c', 'Conv2d_1a_7x7', 64, (7, 7), 2, 'SAME'), ('pool', 'MaxPool_2a_3x3', 3),
                     ('c', 'Conv2d_2b_1x1', 64, (1, 1), 1, 'SAME'), ('c', 'Conv2d_2c_3x3', 192, (3, 3), 1, 'SAME'),
                     ('pool', 'MaxPool_3a_3x3', 3)]
INCEPTION_V1_BLOCKS = [
    ('Mixed_3b', _v1(64, 96, 128, 16, 32, 32)), ('Mixed_3c', _v1(128, 128, 192, 32, 96, 64)),
    ('pool', 'MaxPool_4a_3x3', 3),
    ('Mixed_4b', _v1(192, 96, 208, 16, 48, 64)), ('Mixed_4c', _v1(160, 112, 224, 24, 64, 64)),
    ('Mixed_4d', _v1(128, 128, 256, 24, 64, 64)), ('Mixed_4e', _v1(112, 144, 288, 32, 64, 64)),
    ('Mixed_4f', _v1(256, 160, 320, 32, 128, 128)),
    ('pool', 'MaxPool_5a_2x2', 2),
    ('Mixed_5b', _v1(256, 160, 320, 32, 128, 128, quirk=True)),      # reference scope-name quirk (:240)
    ('Mixed_5c', _v1(384, 192, 384, 48, 128, 128)),
]


def stem_stream_supported(H0, W0):
    """Mirror of comic_stem_stream_supported (csrc/conv_stem.hip): two waves x four tiles x seven pooled columns."""
    return H0 >= 7 and W0 >= 7 and (W0 - 2 - 3) // 2 + 1 <= 56


def stem_stream_1a_supported(Hi, Wi):
    """Mirror of comic_stem_stream_1a_supported: image rows in 16-byte pieces, two new rows by 256 threads, 16 rows in LDS."""
    H0, W0 = (Hi - 3) // 2 + 1, (Wi - 3) // 2 + 1
    return (Hi >= 17 and Wi % 4 == 0 and 2 * (Wi * 3 // 4) <= 512 and stem_stream_supported(H0, W0)
            and 2 * (32 + 64) * 4 + 2 * 4 * 132 * 96 + 16 * Wi * 3 * 4 <= 160 * 1024)


def _out(size, k, s, pad):
    if pad == 'SAME':
        o = -(-size // s)
        tot = max((o - 1) * s + k - size, 0)
        return o, tot // 2
    return (size - k) // s + 1, 0


class CnnPlan:
    """Flat op list + buffer table for one backbone at one input size."""

    def __init__(self, name='inception_v3', image_size=(224, 224), final_endpoint='Mixed_7c', branch_streams=False,
                 group_branches=True, layers=None, pool_after_projection=False, ride_pools=False, side_pools=None,
                 fuse_pools=False, x3=False, fuse_stem_1a=True, fuse_chains=None):
        if name not in ('inception_v3', 'inception_v1', 'chain'):
            raise NotImplementedError('only inception_v3 / inception_v1 are on the MI355X hot path (got %r)' % name)
        self.name = name
        self.ops = []            # dicts
        self.buffers = []        # (H, W, C, f32)
        self.weights = []        # (var_prefix, kh, kw, cin, cout, stem)  -- the VARIABLE's (logical) shape
        self.wphys = []          # (cin, cout) as laid out in the buffers (channels padded to a multiple of 16)
        self.fm_src = None       # bf16 buffer the fp32 feature map is converted from (fm not the last block)
        self.end_points = {}     # name -> buffer id
        self.macs = 0
        self._lane = 0
        self._branch = None
        self.branch_streams = branch_streams
        # same-depth convs of the parallel branches share one launch (comic_cnn_forward_grouped)
        self.group_branches = group_branches and not branch_streams
        self._depth = 0
        self._next_group = 1
        # forward-only rewrite of the Inception pool branches (avg-pool 3x3 -> 1x1 conv -> BN -> ReLU): the
        # projection runs first, next to the other 1x1 convs of the block, into a small fp32 map, and the
        # pool (kind 7, with the BN + ReLU epilogue) then averages Cout instead of Cin channels.  Exact in
        # real arithmetic (the two linear maps act on different axes); frozen-CNN plans only (no backward).
        self.pool_after_projection = pool_after_projection
        # kind-7 ops as extra members of their depth's grouped conv launch instead of launches of their own.
        # Measured slower at batch 64 (the pool workgroups inherit the conv kernel's LDS / register footprint and
        # run at 2 per CU: 5x5|3x3|pool 29.0 us grouped vs 19.8 + 7.8 us), so off by default.
        self.ride_pools = ride_pools
        # kind-7 ops on a branch lane (fork after the block's 1x1 group, join at the block's end) so that the small
        # elementwise kernel runs beside the depth-1..4 convs.  Measured slower under hipGraph replay (forward alone
        # 1.17 -> 1.26 ms: a graph with side branches replays slower on this stack), so off by default.
        self.side_pools = bool(side_pools)
        # second forward-only rewrite (bf16 plans): a 3x3 / 2 max-pool whose only consumers are 1x1 convs
        # (MaxPool_3a -> Conv2d_3b_1x1, MaxPool_5a -> the four 1x1 convs at the head of Mixed_5b) is folded into
        # their loads (COMIC_OP_POOLED_SRC, csrc/conv_ws.hip): the pooled map is never written or re-read.
        self.fuse_pools = bool(fuse_pools)
        self.fuse_stem_1a = bool(fuse_stem_1a)   # with fuse_pools: Conv2d_1a inside the streaming stem op (kind 9)
        # third forward-only rewrite (bf16 plans; default: on with fuse_pools): the 1x7 / 7x1 convs of a Mixed_6b-e branch run as
        # ONE launch per block (tile CHAIN_TILE, csrc/conv_img.hip conv_img_chain_kernel: a workgroup per image and branch,
        # the 12x12 intermediate maps stay in the LDS).  The intermediate buffers of the plan are then never written.
        # Plans without the forward-only rewrites (cnn_finetune) fuse the chains too and KEEP the intermediate maps
        # (OP_CHAIN_KEEP: each linked conv also stores its output, which the backward reads): 26 launches become 5.
        # (ride_pools plans keep one launch per conv depth: their pool ops ride in the depth's conv launch)
        bf16_grouped = not x3 and self.group_branches and name == 'inception_v3' and not self.ride_pools
        self.fuse_chains = bf16_grouped if fuse_chains is None else bool(fuse_chains)
        self.keep_chain_maps = not self.pool_after_projection
        # fuse_chains=None (default): a SIBLING plan with one launch per conv depth serves small batches.  A fused chain is one
        # workgroup per (image, branch) whose convs run one after the other at the matrix rate of ONE CU (~12 us per conv
        # whatever the batch): 2 B workgroups leave most of the chip idle below ~100 images, where the depth-major launches
        # spread every conv over all CUs (forward of 32 images 0.69 ms against 0.78 fused, 64: equal, 200: 1.98 against 1.84,
        # 1280: 9.7-9.85 against 9.47-9.55 ms).  CnnEncoder takes the sibling when its batch is below CHAIN_MIN_BATCH.
        self.small_batch_plan = None
        if fuse_chains is None and self.fuse_chains:
            self.small_batch_plan = CnnPlan(name, image_size, final_endpoint, branch_streams, group_branches, layers,
                                            pool_after_projection, ride_pools, side_pools, fuse_pools, x3, fuse_stem_1a,
                                            fuse_chains=False)
        if self.fuse_chains and (x3 or not self.group_branches):
            raise ValueError('fuse_chains needs a grouped bf16 plan')
        if self.fuse_pools and not (pool_after_projection and name == 'inception_v3'):
            raise ValueError('fuse_pools needs an inception_v3 plan with pool_after_projection=True')
        self._pooled_src = None  # (buffer id, pooled H, pooled W) while a folded max-pool waits for its consumers
        self._logical = {}       # buffer id -> logical channel count where it differs from the padded one
        self.block_ranges = []   # [first op, end op) of the stem and of every block, in execution order (gradient buckets)
        if name == 'chain':
            self._build_chain(image_size, layers)
        elif name == 'inception_v1':
            self._build_v1(image_size, 'Mixed_4f' if final_endpoint == 'Mixed_7c' else final_endpoint)
        else:
            self._build_v3(image_size, final_endpoint)
        # "bf16x3": fp32-class accuracy on the bf16 kernels (COMIC_OP_X3, include/comic_hip.h).  Every bf16 activation
        # buffer holds its tensor as three channel regions [hi | lo | hi]; a conv reads all 3C channels against a filter
        # packed [W_hi | W_hi | W_lo] per tap (CnnEncoder), i.e. hi*W_hi + lo*W_hi + hi*W_lo with fp32 accumulation.
        self.x3 = bool(x3)
        if self.x3:
            if fuse_pools or self._logical or name == 'inception_v1':
                raise ValueError('x3 plans: InceptionV3 / chain layouts only (no folded max-pools, no padded channels)')
            for o in self.ops:
                if o['kind'] == 0:
                    o['Cin'] *= 3
                if o['kind'] in (0, 1, 2, 3, 7):       # (7: the pool-after-projection branches store the three regions too)
                    o['flags'] = o.get('flags', 0) | L.OP_X3
            self.buffers = [(H, W, Cc if f32 else 3 * Cc, f32) for (H, W, Cc, f32) in self.buffers]

    # -- builder helpers ---------------------------------------------------------------
    def _buf(self, H, W, Cc, f32=False):
        self.buffers.append((H, W, Cc, f32))
        return len(self.buffers) - 1

    def _conv(self, src, scope, spec, dst=None, dst_coff=0, out_f32=False, raw=False):
        _, name, cout, (kh, kw), stride, pad = spec
        H, W, Cin_p, _ = self.buffers[src]
        Cin = self._logical.get(src, Cin_p)          # channels of the producing variable (the rest is zero padding)
        pooled = self._pooled_src is not None and self._pooled_src[0] == src
        if pooled:
            assert (kh, kw, stride) == (1, 1, 1), 'a folded max-pool feeds 1x1 convs only'
            Ho, Wo, pt, pl = self._pooled_src[1], self._pooled_src[2], 0, 0
        else:
            Ho, pt = _out(H, kh, stride, pad)
            Wo, pl = _out(W, kw, stride, pad)
        cout_p = (cout + 15) // 16 * 16              # MFMA tile granularity; Inception-V1 has 24-channel reduces
        if dst is None:
            dst = self._buf(Ho, Wo, cout_p, out_f32)
            if cout_p != cout:
                self._logical[dst] = cout
        else:
            assert cout_p == cout, 'a conv that writes a concat slice needs Cout % 16 == 0'
        stem = Cin <= 4
        self.weights.append((scope + '/' + name, kh, kw, Cin, cout, stem))
        self.wphys.append((Cin_p, cout_p))
        self.ops.append(dict(kind=1 if stem else 0, src=src, dst=dst, src_coff=0, dst_coff=dst_coff, H=H, W=W,
                             Cin=Cin_p, Cout=cout_p, KH=kh, KW=kw, SH=stride, SW=stride, PT=pt, PL=pl, Ho=Ho, Wo=Wo,
                             weight=len(self.weights) - 1, relu=0 if raw else 1, out_f32=int(out_f32), lane=self._lane,
                             depth=self._depth, flags=(L.OP_RAW if raw else 0) | (L.OP_POOLED_SRC if pooled else 0),
                             branch=self._branch))
        self.macs += Ho * Wo * kh * kw * Cin * cout
        return dst, (Ho, Wo, cout)

    @staticmethod
    def _stem_dims(H, W):
        """(H0, W0) of Conv2d_1a_3x3's output (3x3 / 2 VALID) for an H x W image."""
        return _out(H, 3, 2, 'VALID')[0], _out(W, 3, 2, 'VALID')[0]

    def _stem_stream(self, src, scope, spec_a, spec_b, spec_1a=None):
        """kind 8: conv 3x3 VALID 32 -> 32, conv 3x3 SAME 32 -> 64 (+ BN + ReLU each), max-pool 3x3 / 2 VALID in one
        streaming pass; the weight records of the two convs are adjacent.  spec_1a: kind 9 -- `src` is the fp32 image and
        Conv2d_1a_3x3 (3x3 / 2 VALID, 3 -> 32) runs inside the same pass, its record in front of the other two."""
        Hs, Ws, Cin, _ = self.buffers[src]
        assert spec_a[2:] == (32, (3, 3), 1, 'VALID') and spec_b[2:] == (64, (3, 3), 1, 'SAME')
        if spec_1a is not None:
            assert Cin == 3 and spec_1a[2:] == (32, (3, 3), 2, 'VALID')
            H0, W0 = self._stem_dims(Hs, Ws)
            self.weights.append((scope + '/' + spec_1a[1], 3, 3, 3, 32, True))
            self.wphys.append((3, 32))
            self.macs += H0 * W0 * 27 * 32
        else:
            assert Cin == 32
            H0, W0 = Hs, Ws
        H1, W1 = H0 - 2, W0 - 2
        Hp, Wp = _out(H1, 3, 2, 'VALID')[0], _out(W1, 3, 2, 'VALID')[0]
        dst = self._buf(Hp, Wp, 64)
        self.weights.append((scope + '/' + spec_a[1], 3, 3, 32, 32, False))
        self.wphys.append((32, 32))
        self.weights.append((scope + '/' + spec_b[1], 3, 3, 32, 64, False))
        self.wphys.append((32, 64))
        n_w = 3 if spec_1a is not None else 2
        self.ops.append(dict(kind=9 if spec_1a is not None else 8, src=src, dst=dst, src_coff=0, dst_coff=0, H=Hs, W=Ws,
                             Cin=Cin, Cout=64, KH=3, KW=3, SH=1, SW=1, PT=0, PL=0, Ho=Hp, Wo=Wp,
                             weight=len(self.weights) - n_w, relu=1, out_f32=0, lane=0, depth=0))
        self.macs += H1 * W1 * 9 * 32 * 32 + H1 * W1 * 9 * 32 * 64
        return dst

    def _pool_bn_relu(self, src, weight, dst, dst_coff, out_f32):
        """kind 7: 3x3 s1 SAME average of the fp32 map `src` + BN (weight record `weight`) + ReLU -> dst slice."""
        H, W, Cc, f32 = self.buffers[src]
        assert f32
        self.ops.append(dict(kind=7, src=src, dst=dst, src_coff=0, dst_coff=dst_coff, H=H, W=W, Cin=Cc, Cout=Cc,
                             KH=3, KW=3, SH=1, SW=1, PT=1, PL=1, Ho=H, Wo=W, weight=weight, relu=1,
                             out_f32=int(out_f32), src_f32=1, lane=self._lane, depth=self._depth))

    def _pool(self, src, kind, k, stride, pad, dst=None, dst_coff=0):
        H, W, Cc, _ = self.buffers[src]
        Ho, pt = _out(H, k, stride, pad)
        Wo, pl = _out(W, k, stride, pad)
        if dst is None:
            dst = self._buf(Ho, Wo, Cc)
        self.ops.append(dict(kind=kind, src=src, dst=dst, src_coff=0, dst_coff=dst_coff, H=H, W=W, Cin=Cc, Cout=Cc,
                             KH=k, KW=k, SH=stride, SW=stride, PT=pt, PL=pl, Ho=Ho, Wo=Wo, weight=-1, relu=0,
                             out_f32=0, lane=self._lane, depth=self._depth, branch=self._branch))
        return dst, (Ho, Wo, Cc)

    def _schedule_by_depth(self, first_op):
        """Reorder one Inception block's ops depth-major (an op at depth d of a branch depends
        only on the depth d-1 op of the same branch), pools first, and give the >=2 convs of
        one depth a common group id, heaviest reduction first so the long workgroups start
        early."""
        block = self.ops[first_op:]
        del self.ops[first_op:]
        forked = False
        chains = self._find_chains(block) if self.fuse_chains else []
        chained = [o for ch in chains for o in ch]
        for ch in chains:        # what follows a chain in its branch moves up to the chain's depth (the chain launch precedes that level)
            rest = sorted((o for o in block if o.get('branch') == ch[0].get('branch') and o['depth'] > ch[-1]['depth']),
                          key=lambda o: o['depth'])
            for k, o in enumerate(rest):
                o['depth'] = ch[0]['depth'] + k
        block = [o for o in block if not any(o is c for c in chained)]
        for di, d in enumerate(sorted({o['depth'] for o in block})):
            if di == 1 and chains:
                self._emit_chains(chains)
                chains = []
            level = [o for o in block if o['depth'] == d]
            # pool + BN + ReLU ops (kind 7) ride in the conv launch of their depth as extra members (last, so the
            # group's tile id stays on its first conv): elementwise work under the MFMA tiles instead of a launch
            riders = [o for o in level if o['kind'] == 7] if self.ride_pools else []
            for o in level:
                if o['kind'] != 0 and o not in riders:
                    if o['kind'] == 7 and self.side_pools:
                        o['lane'] = 1
                        self.ops.append(self._sync_op(5))
                        forked = True
                    self.ops.append(o)
            convs = sorted((o for o in level if o['kind'] == 0), key=lambda o: -(o['KH'] * o['KW'] * o['Cin']))
            if not convs:
                self.ops += riders
                continue
            members = convs + riders
            if len(members) >= 2:
                for o in members:
                    o['group'] = self._next_group
                self._next_group += 1
            self.ops += members
        if chains:
            self._emit_chains(chains)
        if forked:
            self.ops.append(self._sync_op(6))

    CHAIN_CHANNELS = (128, 160, 192)          # comic_img_chain_supported (csrc/conv_img.hip)
    CHAIN_MIN_BATCH = 96                      # images per forward from which the fused chains pay (see small_batch_plan)

    def _find_chains(self, block):
        """Runs of >= 2 consecutive convs of one branch, from depth 1 on, that are stride-1 SAME 7-tap convs over 12x12 maps
        and keep the channel count until a last conv with 192 outputs (the branches of Mixed_6b-e and Branch_1 of Mixed_7a,
        inception_v3.py:262-366): at most two per block, all over one channel count.  -> [[op, ...], ...]"""
        def seven(o, cin):
            return (o['kind'] == 0 and o['H'] * o['W'] == 144 and (o['Ho'], o['Wo']) == (o['H'], o['W']) and
                    o['SH'] == o['SW'] == 1 and o['KH'] * o['KW'] == 7 and o['Cin'] == cin and not o.get('flags', 0))
        out = []
        for b in sorted({o.get('branch') for o in block if o.get('branch') is not None}):
            ops = sorted((o for o in block if o.get('branch') == b), key=lambda o: o['depth'])
            if len(ops) < 3 or ops[0]['kind'] != 0:
                continue
            cin = ops[1]['Cin']
            run = []
            for o in ops[1:]:
                if not (cin in self.CHAIN_CHANNELS and seven(o, cin) and len(run) < 4):
                    break
                if run and not (run[-1]['Cout'] == cin and not run[-1]['out_f32'] and run[-1]['dst'] == o['src']
                                and run[-1]['dst_coff'] == 0 and sum(1 for q in block if q['src'] == run[-1]['dst']) == 1):
                    break                               # (an intermediate map must have no other reader)
                run.append(o)
            while run and run[-1]['Cout'] != 192:
                run.pop()
            if len(run) >= 2:
                out.append(run)
        if len(out) > 2 or len({ch[0]['Cin'] for ch in out}) > 1:
            return []
        return out

    def _emit_chains(self, chains):
        for ch in chains:
            for o in ch[:-1]:
                o['flags'] = o.get('flags', 0) | L.OP_CHAIN_LINK | (L.OP_CHAIN_KEEP if self.keep_chain_maps else 0)
            for o in ch:
                o['group'] = self._next_group
                o['tile'] = L.CHAIN_TILE
                self.ops.append(o)
        self._next_group += 1

    @staticmethod
    def _sync_op(kind):
        z = dict.fromkeys(('src', 'dst', 'src_coff', 'dst_coff', 'H', 'W', 'Cin', 'Cout', 'KH', 'KW', 'SH', 'SW', 'PT',
                           'PL', 'Ho', 'Wo', 'relu', 'out_f32'), 0)
        z.update(kind=kind, weight=-1, lane=0)
        return z

    @staticmethod
    def _branch_out(branch, H, W, Cin):
        """(Ho, Wo, C) produced by a branch."""
        C_ = Cin
        for op in branch:
            if op[0] == 'c':
                _, _, cout, (kh, kw), s, pad = op
                H, W, C_ = _out(H, kh, s, pad)[0], _out(W, kw, s, pad)[0], cout
            elif op[0] == 'max':
                H, W = _out(H, 3, 2, 'VALID')[0], _out(W, 3, 2, 'VALID')[0]
            elif op[0] == 'fork':
                C_ = sum(o[2] for o in op[1])
        return H, W, C_

    def _build_chain(self, image_size, layers):
        """A plain sequence of the same op forms as the InceptionV3 table (('c', scope, cout, (kh, kw),
        stride, pad) | ('max',) | ('avg',)) under scope 'Chain', ending like the real network: the last
        conv writes the fp32 feature map, followed by the head's global average pool.  Used to
        exercise every forward / backward kernel on shallow stacks."""
        H, W = image_size
        cur = self._buf(H, W, 3, True)
        self.input = cur
        assert layers and layers[-1][0] == 'c'
        for li, op in enumerate(layers):
            last = li == len(layers) - 1
            if op[0] == 'c':
                cur, _ = self._conv(cur, 'Chain', op, out_f32=last)
                self.end_points[op[1]] = cur
            elif op[0] == 'max':
                cur, _ = self._pool(cur, 2, 3, 2, 'VALID')
            else:
                cur, _ = self._pool(cur, 3, 3, 1, 'SAME')
        self._finish_head(cur)

    def _finish_head(self, cur):
        # head (inception_v3.py:520-532): kernel = min(8, H_f), VALID, num_classes=None
        Hf, Wf, Cf, f32 = self.buffers[cur]
        self.fm = cur
        kh, kw = min(Hf, 8), min(Wf, 8)
        Hp, Wp = Hf - kh + 1, Wf - kw + 1
        pooled = self._buf(Hp, Wp, Cf, True)
        self.ops.append(dict(kind=4, src=cur, dst=pooled, src_coff=0, dst_coff=0, H=Hf, W=Wf, Cin=Cf, Cout=Cf, KH=kh,
                             KW=kw, SH=1, SW=1, PT=0, PL=0, Ho=Hp, Wo=Wp, weight=-1, relu=0, out_f32=1,
                             src_f32=int(f32), lane=0))
        self.pooled = pooled
        self.end_points['AvgPool_1a'] = pooled
        if not self.block_ranges:
            self.block_ranges.append([0, len(self.ops)])
        self.block_ranges[-1][1] = len(self.ops)       # the head pool belongs to the last range

    def _build_v1(self, image_size, fm_endpoint):
        """inception_v1_base (inception_v1.py:29-266) + head (:319-327): conv / max-pool defaults stride 1,
        SAME; feature map = `fm_endpoint` (reference default Mixed_4f, train.py:65), net = 7x7 VALID
        average of Mixed_5c."""
        H, W = image_size
        cur = self._buf(H, W, 3, True)
        self.input = cur
        root = 'InceptionV1'
        for op in INCEPTION_V1_STEM:
            if op[0] == 'c':
                cur, _ = self._conv(cur, root, op)
            else:
                cur, _ = self._pool(cur, 2, op[2], 2, 'SAME')
            self.end_points[op[1]] = cur
        self.block_ranges.append([0, len(self.ops)])
        for entry in INCEPTION_V1_BLOCKS:
            if entry[0] == 'pool':
                cur, _ = self._pool(cur, 2, entry[2], 2, 'SAME')
                self.end_points[entry[1]] = cur
                self.block_ranges[-1][1] = len(self.ops)
                continue
            bname, branches = entry
            Hi, Wi, Ci, _ = self.buffers[cur]
            Ctot = sum(b[-1][2] for b in branches)
            last = bname == 'Mixed_5c'
            blk = self._buf(Hi, Wi, Ctot, f32=last)          # pooled by the head in fp32 (as InceptionV3's last block)
            first_op = len(self.ops)
            coff = 0
            for bi, branch in enumerate(branches):
                scope = '%s/%s/Branch_%d' % (root, bname, bi)
                x = cur
                for oi, op in enumerate(branch):
                    self._depth = oi
                    if op[0] == 'maxs1':
                        x, _ = self._pool(x, 2, 3, 1, 'SAME')
                    elif oi == len(branch) - 1:
                        self._conv(x, scope, op, blk, coff, out_f32=last)
                    else:
                        x, _ = self._conv(x, scope, op)
                coff += branch[-1][2]
            self._depth = 0
            if self.group_branches:
                self._schedule_by_depth(first_op)
            self.block_ranges.append([first_op, len(self.ops)])
            cur = blk
            self.end_points[bname] = cur
        if fm_endpoint not in self.end_points:
            raise ValueError('unknown feature-map end point %r' % fm_endpoint)
        self._finish_head(cur)
        if fm_endpoint != 'Mixed_5c':
            # the attention feature map is an inner end point: kept in the plan dtype for the layers that
            # follow and converted to fp32 for the decoder by the encoder (CnnEncoder.forward)
            self.fm_src = self.end_points[fm_endpoint]
            self.fm = None

    def _build_v3(self, image_size, final_endpoint):
        H, W = image_size
        cur = self._buf(H, W, 3, True)          # fp32 images in [-1, 1]
        self.input = cur
        root = 'InceptionV3'
        stem = list(_STEM)
        if self.fuse_pools and stem_stream_supported(*self._stem_dims(H, W)):
            # Conv2d_2a -> Conv2d_2b -> MaxPool_3a as one streaming op (kind 8, csrc/conv_stem.hip); with Conv2d_1a in
            # front of them in the same pass (kind 9) when the image rows split into 16-byte pieces
            if self.fuse_stem_1a and stem_stream_1a_supported(H, W):
                cur = self._stem_stream(cur, root, stem[1], stem[2], spec_1a=stem[0])
            else:
                cur, _ = self._conv(cur, root, stem[0])
                self.end_points[stem[0][1]] = cur
                cur = self._stem_stream(cur, root, stem[1], stem[2])
            self.end_points[stem[3][1]] = cur
            stem = stem[4:]
        for op in stem:
            if op[0] == 'c':
                cur, _ = self._conv(cur, root, op)
                self._pooled_src = None
            elif self.fuse_pools:
                Hc, Wc = self.buffers[cur][:2]
                self._pooled_src = (cur, _out(Hc, 3, 2, 'VALID')[0], _out(Wc, 3, 2, 'VALID')[0])
                continue
            else:
                cur, _ = self._pool(cur, 2, 3, 2, 'VALID')
            self.end_points[op[1]] = cur
        self.block_ranges.append([0, len(self.ops)])
        for bname, branches in INCEPTION_V3_BLOCKS:
            Hi, Wi, Ci, _ = self.buffers[cur]
            block_first = len(self.ops)
            if self._pooled_src is not None:         # the block reads `cur` through the folded MaxPool_5a
                assert self._pooled_src[0] == cur and all(b[0][0] == 'c' and b[0][3] == (1, 1) or b[0][0] == 'avg'
                                                          for b in branches)
                Hi, Wi = self._pooled_src[1:]
            outs = [self._branch_out(b, Hi, Wi, Ci) for b in branches]
            Ho, Wo = outs[0][0], outs[0][1]
            Ctot = sum(o[2] for o in outs)
            last = bname == final_endpoint
            blk = self._buf(Ho, Wo, Ctot, f32=last)   # the attention feature map is handed over in fp32
            coff = 0
            if self.branch_streams:
                self.ops.append(self._sync_op(5))      # fork: the branches are independent
            first_op = len(self.ops)
            for bi, branch in enumerate(branches):
                scope = '%s/%s/Branch_%d' % (root, bname, bi)
                self._lane = bi if self.branch_streams else 0   # branch 0 stays on the caller's stream
                self._branch = bi
                x = cur
                if (self.pool_after_projection and len(branch) == 2 and branch[0][0] == 'avg' and branch[1][0] == 'c'
                        and branch[1][3] == (1, 1) and branch[1][2] % 16 == 0):
                    self._depth = 0
                    z, _ = self._conv(x, scope, branch[1], out_f32=True, raw=True)
                    self._depth = 1
                    self._pool_bn_relu(z, len(self.weights) - 1, blk, coff, out_f32=last)
                    coff += outs[bi][2]
                    continue
                for oi, op in enumerate(branch):
                    final = oi == len(branch) - 1
                    self._depth = oi
                    if op[0] == 'c':
                        if final:
                            self._conv(x, scope, op, blk, coff, out_f32=last)
                        else:
                            x, _ = self._conv(x, scope, op)
                    elif op[0] == 'avg':
                        x, _ = self._pool(x, 3, 3, 1, 'SAME')
                    elif op[0] == 'max':
                        assert final
                        if last:
                            raise NotImplementedError
                        self._pool(x, 2, 3, 2, 'VALID', blk, coff)
                    elif op[0] == 'fork':
                        assert final
                        o = coff
                        for sub in op[1]:
                            self._conv(x, scope, sub, blk, o, out_f32=last)
                            o += sub[2]
                coff += outs[bi][2]
            self._lane = 0
            self._depth = 0
            self._branch = None
            self._pooled_src = None
            for o in self.ops[first_op:]:
                o.setdefault('block_in', cur)        # (python-side: the block's shared input buffer, see backward_schedule)
            if self.group_branches:
                self._schedule_by_depth(first_op)
            if self.branch_streams:
                self.ops.append(self._sync_op(6))      # join
            self.block_ranges.append([block_first, len(self.ops)])
            cur = blk
            self.end_points[bname] = cur
            if last:
                break
        self._finish_head(cur)

    def fm_dims(self):
        """(H, W, C) of the attention feature map."""
        return self.buffers[self.fm if self.fm is not None else self.fm_src][:3]

    # -- parameters -----------------------------------------------------------------------
    def param_shapes(self):
        out = {}
        for prefix, kh, kw, cin, cout, _ in self.weights:
            out[prefix + '/weights'] = (kh, kw, cin, cout)
            for s in ('beta', 'moving_mean', 'moving_variance'):
                out[prefix + '/BatchNorm/' + s] = (cout,)
        return out

    def init_params(self, seed=0):
        """slim.variance_scaling_initializer() convs, BN beta 0 / mean 0 / var 1;
        normally overwritten by the slim checkpoint (model_base.py:468-482)."""
        rng = np.random.default_rng(seed)
        p = {}
        for prefix, kh, kw, cin, cout, _ in self.weights:
            std = math.sqrt(1.3 * 2.0 / (kh * kw * cin))
            w = np.clip(rng.standard_normal((kh, kw, cin, cout)), -2, 2).astype(np.float32) * np.float32(std)
            p[prefix + '/weights'] = w
            p[prefix + '/BatchNorm/beta'] = np.zeros(cout, np.float32)
            p[prefix + '/BatchNorm/moving_mean'] = np.zeros(cout, np.float32)
            p[prefix + '/BatchNorm/moving_variance'] = np.ones(cout, np.float32)
        return p


SCHED_RUN, SCHED_FORK, SCHED_JOIN_ADD = 0, 1, 2


def backward_schedule(plan):
    """cnn_finetune: the order and lanes of the backward pass for comic_cnn_backward_sched -- rows (action, index, lane, alt).
    The parallel branches of an Inception block are independent chains until the backward-data products of their head convs
    (and the pool branch's pool gradient) ACCUMULATE into the gradient of the block's shared input.  Inside a block the
    branches are dealt to two lanes (longest first, by conv count); lane 1 accumulates into an alternate copy of that shared
    gradient, and the join at the block's end adds it in.  Outside the blocks (stem, head) everything is lane 0."""
    rows = []
    alt_bufs = set()
    done = set()
    for (lo, hi) in reversed([tuple(r) for r in plan.block_ranges]):
        idx = [i for i in range(hi - 1, lo - 1, -1) if plan.ops[i]['kind'] not in (5, 6)]
        br = sorted({plan.ops[i].get('branch') for i in idx} - {None})
        lanes = {}
        if len(br) >= 2:
            load = [0, 0]
            # launches of a branch's chain: one per op (the activation gradients of its inner convs ride in the backward-data
            # epilogue of their reader) + the activation gradient of the conv that writes the block's output slice
            weight = {b: 1 + sum(1 for i in idx if plan.ops[i].get('branch') == b) for b in br}
            for b in sorted(br, key=lambda b: -weight[b]):
                ln = 0 if load[0] <= load[1] else 1
                lanes[b] = ln
                load[ln] += weight[b]
        # the fork goes in FRONT of the block's first op: lane 1's branches only need what precedes the block (the join of the
        # block behind it).  The ops of a block are walked deepest first, so a fork at the first lane-1 op came after the
        # deepest two or three convs of lane 0's branch -- lane 1 then sat out half of lane 0's chain and the lanes ran one
        # after the other (profiles/r05_finetune_lanes.txt: one kernel in flight for 3.6 of 5.9 ms)
        # ... of its BRANCHES: an op of the range that belongs to no branch (the head's global pool in the last range: it adds
        # into the gradient of the block's output, which lane 1's first ops read) runs on lane 0 outside the fork / join
        # region -- with the fork in front of it the two raced (run-to-run differences of 1e-4 in the weight gradients)
        two = any(ln == 1 for ln in lanes.values())
        forked, blk_in = False, None
        for i in idx:
            o = plan.ops[i]
            in_branch = o.get('branch') is not None
            if two and in_branch and not forked:
                rows.append((SCHED_FORK, 0, 0, 0))
                forked, blk_in = True, None
            elif forked and not in_branch:
                rows.append((SCHED_JOIN_ADD, blk_in if blk_in is not None else -1, 0, 0))
                forked = False
            ln = lanes.get(o.get('branch'), 0) if forked else 0
            alt = 0
            if ln == 1 and o.get('block_in') is not None and o['src'] == o['block_in'] and o['src'] != plan.input:
                alt, blk_in = 1, o['src']
                alt_bufs.add(o['src'])
            rows.append((SCHED_RUN, i, ln, alt))
            done.add(i)
        if forked:
            rows.append((SCHED_JOIN_ADD, blk_in if blk_in is not None else -1, 0, 0))
    assert done == {i for i, o in enumerate(plan.ops) if o['kind'] not in (5, 6)}
    return np.asarray(rows, np.int32).reshape(-1, 4), sorted(alt_bufs)


def flat_layout(plan):
    """Shapes of the flat master buffers: packed conv weights ([Cout][Kpad]; stem [K][Cout]) and per-channel vectors."""
    wshapes, bshapes = {}, {}
    for i, (prefix, kh, kw, cin, cout, stem) in enumerate(plan.weights):
        cin_p, cout_p = plan.wphys[i]            # physical (padded) channel counts
        K = kh * kw * cin_p
        wshapes['w%d' % i] = (K * cout_p,) if stem else (cout_p * ((K + 63) // 64 * 64),)
        bshapes['b%d' % i] = (cout_p,)
    return wshapes, bshapes


def plan_grad_buckets(plan, w_flat, b_flat, n=6):
    """See CnnEncoder.grad_buckets; w_flat / b_flat: the FlatParams of the weights / betas (offsets only)."""
    ranges = [tuple(r) for r in plan.block_ranges]
    assert ranges and ranges[0][0] == 0 and ranges[-1][1] == len(plan.ops)
    assert all(a[1] == b[0] for a, b in zip(ranges, ranges[1:])), 'block ranges must tile the op list'

    def widx(r):
        ws = [plan.ops[i]['weight'] for i in range(*r) if plan.ops[i].get('weight', -1) >= 0 and plan.ops[i]['kind'] in (0, 1, 8, 9)]
        return (min(ws), max(ws) + 1) if ws else None
    nW = len(plan.weights)
    woff = [w_flat.offsets['w%d' % i] for i in range(nW)] + [w_flat.numel]
    boff = [b_flat.offsets['b%d' % i] for i in range(nW)] + [b_flat.numel]
    total = woff[-1]
    out, hi_op, hi_w, acc_target = [], len(plan.ops), nW, total / float(max(1, n))
    lo_w = nW
    for k in range(len(ranges) - 1, -1, -1):
        wr = widx(ranges[k])
        if wr is not None:
            assert wr[1] <= lo_w or wr[1] == hi_w, 'weights of a block must form one range below the later blocks'
            lo_w = min(lo_w, wr[0])
        size = woff[hi_w] - woff[lo_w]
        if (size >= acc_target and len(out) < n - 1) or k == 0:
            out.append((ranges[k][0], hi_op, (woff[lo_w], woff[hi_w]), (boff[lo_w], boff[hi_w])))
            hi_op, hi_w = ranges[k][0], lo_w
    assert out[-1][0] == 0 and out[-1][2][0] == 0
    return out


class CnnEncoder:
    """Device-resident encoder: packed weights, folded BN, activation buffers, one native
    forward call.  `dtype` 'bf16' (throughput path) or 'f32' (exact-fp32 MFMA, parity path)."""

    def __init__(self, plan: CnnPlan, params: dict, batch: int, dtype='bf16', device='cuda:0', weights_from=None):
        import torch
        from .decoder import FlatParams
        self.torch = torch
        self.lib = L.load()
        if dtype == 'bf16x3':              # the bf16 kernels over an x3 plan (CnnPlan(x3=True)): fp32-class accuracy
            assert getattr(plan, 'x3', False), "dtype 'bf16x3' needs a plan built with x3=True"
            dtype = 'bf16'
        assert dtype in ('bf16', 'f32') and (dtype == 'bf16' or not getattr(plan, 'x3', False))
        small = getattr(plan, 'small_batch_plan', None)
        if small is not None and (batch < plan.CHAIN_MIN_BATCH or dtype != 'bf16'):
            plan = small          # same buffers, weights and end points; one launch per conv depth instead of fused chains
        self.plan, self.batch, self.dtype, self.device = plan, batch, dtype, device
        self.dcode = 1 if dtype == 'bf16' else 0
        if getattr(plan, 'fuse_pools', False) and dtype != 'bf16':
            raise ValueError('fuse_pools plans run on the bf16 kernels only')
        tdt = torch.bfloat16 if dtype == 'bf16' else torch.float32
        self._tdt = tdt
        st = L.stream_ptr()
        # fp32 masters in the packed kernel layout ([Cout][Kpad]; stem [K][Cout]) in ONE flat buffer, BN
        # vectors in a second one: the forward reads a plan-dtype copy of the first (the fp32 plan reads
        # the masters themselves), cnn_finetune trains both (comic_cnn_backward + AdamTF on the flats)
        if weights_from is not None:
            # a second encoder (other batch size) over the SAME variables, like the reference's
            # reuse=True graphs (train_fn.py:60-66): alias the flat buffers and the weight table
            o = weights_from
            assert o.plan.weights == plan.weights and o.dtype == dtype
            self.w_master, self.beta, self.mean, self.scale, self.shift = o.w_master, o.beta, o.mean, o.scale, o.shift
            self.w_plan, wt = o.w_plan, o._wt
            self.w_frag, self._frag_table = o.w_frag, o._frag_table
            self._x3_off = o._x3_off
            assert getattr(o.plan, 'x3', False) == getattr(plan, 'x3', False)
        else:
            wshapes, bshapes = flat_layout(plan)
            self.w_master = FlatParams(wshapes, device)
            self.beta = FlatParams(bshapes, device)
            self.mean, self.scale, self.shift = self.beta.like(), self.beta.like(), self.beta.like()
            self.scale.data.fill_(1.0)
            self.w_plan = self.w_master.data if self.dcode == 0 else torch.zeros(self.w_master.numel, dtype=tdt,
                                                                                  device=device)
            # x3 plans: the plan copy holds [W_hi | W_hi | W_lo] per filter tap, rows of roundup64(3K) elements
            self._x3_off = None
            if getattr(plan, 'x3', False):
                assert self.dcode == 1, 'x3 plans run on the bf16 kernels'
                offs, n = [], 0
                for i, (prefix, kh, kw, cin, cout, stem) in enumerate(plan.weights):
                    cin_p, cout_p = plan.wphys[i]
                    offs.append(n)
                    if not stem:
                        n += cout_p * ((3 * kh * kw * cin_p + 63) // 64 * 64)
                self._x3_off = offs
                self.w_plan = torch.zeros(max(n, 64), dtype=tdt, device=device)
            # bf16 plans: a second copy of the weights in MFMA-fragment order for the image-resident kernel
            # (csrc/conv_img.hip), refreshed with the plan copy; table = {element offset, Cout, Kpad} per weight
            self.w_frag, self._frag_table = None, None
            if self.dcode == 1 and self._x3_off is None:
                self.w_frag = torch.zeros(self.w_master.numel, dtype=tdt, device=device)
                tab = []
                for i, (prefix, kh, kw, cin, cout, stem) in enumerate(plan.weights):
                    cin_p, cout_p = plan.wphys[i]
                    kpad = (kh * kw * cin_p + 63) // 64 * 64
                    tab.append((self.w_master.offsets['w%d' % i], cout_p, 0 if stem else kpad))
                order = sorted(range(len(tab)), key=lambda i: tab[i][0])
                assert all(tab[i][0] % 8 == 0 for i in order), 'weight records must start on 16-byte boundaries'
                self._frag_table = torch.tensor([tab[i] for i in order], dtype=torch.int64, device=device)
            wt = (L.ConvWeight * len(plan.weights))()
            for i, (prefix, kh, kw, cin, cout, stem) in enumerate(plan.weights):
                bk = 'b%d' % i
                esz = 4 if (stem or self.dcode == 0) else 2
                wbase = self.w_master.data.data_ptr() if (stem or self.dcode == 0) else self.w_plan.data_ptr()
                wt[i].w = wbase + esz * (self._x3_off[i] if (self._x3_off is not None and not stem) else
                                         self.w_master.offsets['w%d' % i])
                wt[i].scale = self.scale.view(bk).data_ptr()
                wt[i].shift = self.shift.view(bk).data_ptr()
                if self.w_frag is not None and not stem:
                    wt[i].w_frag = self.w_frag.data_ptr() + 2 * self.w_master.offsets['w%d' % i]
            self.load_params(params)
        self._train = None
        torch.cuda.synchronize()
        self._wt = wt
        self.bufs = []
        for (H, W, Cc, f32) in plan.buffers:
            self.bufs.append(torch.empty((batch, H, W, Cc), dtype=torch.float32 if f32 else tdt, device=device))
        self._bufptr = (C.c_void_p * len(self.bufs))(*[b.data_ptr() for b in self.bufs])
        self._bufch = (C.c_int32 * len(self.bufs))(*[b[2] for b in plan.buffers])
        ops = (L.CnnOp * len(plan.ops))()
        for i, o in enumerate(plan.ops):
            for k, v in o.items():
                if k not in ('depth', 'branch', 'block_in'):
                    setattr(ops[i], k, v)
            if self.dcode != 1:
                ops[i].group = 0               # the fp32 parity path launches every conv on its own
                if o.get('tile') == L.CHAIN_TILE:
                    ops[i].tile = 0
                    ops[i].flags &= ~(L.OP_CHAIN_LINK | L.OP_CHAIN_KEEP)
        self._ops = ops
        self._group_args = None
        self._build_group_args()
        self._graph = None
        self._calls = 0
        self._alt = {}                 # adopted input tensors: address -> [pointer table, graph, calls, tensor]
        self._last_bufptr = None
        self.backward_lanes = True     # cnn_finetune: weight gradients on a lane of their own (enable_training)
        self.backward_branch_lanes = True   # ... and the branches of an Inception block on two chain lanes (backward_schedule)
        self._polite_lds_kb = 0
        self._fm_f32 = None

    @property
    def polite_lds_kb(self):
        """> 0: conv workgroups request at least this much LDS (1 per CU at 84) -- for forwards that run on a
        second stream under other kernels (CaptionTrainer's overlapped encoder).  Written into every op record
        (comic_cnn_op.min_lds); a captured graph is dropped because its launches hold the old value."""
        return self._polite_lds_kb

    @polite_lds_kb.setter
    def polite_lds_kb(self, kb):
        kb = int(kb)
        if kb == self._polite_lds_kb:
            return
        self._polite_lds_kb = kb
        for i in range(len(self.plan.ops)):
            self._ops[i].min_lds = kb * 1024
        self._drop_graphs()

    def load_params(self, params):
        """(Re)load every CNN variable from {slim name: array} into the flat masters IN PLACE (all
        encoders that share them, and any optimiser bound to them, see the new values)."""
        torch, st = self.torch, L.stream_ptr()
        for i, (prefix, kh, kw, cin, cout, stem) in enumerate(self.plan.weights):
            cin_p, cout_p = self.plan.wphys[i]
            w = torch.from_numpy(np.ascontiguousarray(params[prefix + '/weights'], np.float32)).to(self.device)
            assert tuple(w.shape) == (kh, kw, cin, cout), prefix
            beta, mean, var = (torch.from_numpy(np.ascontiguousarray(
                params[prefix + '/BatchNorm/' + s], np.float32)).to(self.device)
                for s in ('beta', 'moving_mean', 'moving_variance'))
            if (cin_p, cout_p) != (cin, cout):          # zero weights / beta 0, mean 0, var 1 on the padding channels
                wp = torch.zeros((kh, kw, cin_p, cout_p), dtype=torch.float32, device=self.device)
                wp[:, :, :cin, :cout] = w
                w = wp
                pad = lambda t, fill: torch.cat([t, torch.full((cout_p - cout,), fill, dtype=torch.float32,
                                                               device=self.device)])
                beta, mean, var = pad(beta, 0.0), pad(mean, 0.0), pad(var, 1.0)
            bk = 'b%d' % i
            self.beta.view(bk).copy_(beta)
            self.mean.view(bk).copy_(mean)
            L.check(self.lib.comic_fold_bn(beta.data_ptr(), mean.data_ptr(), var.data_ptr(), BN_EPS,
                                           self.scale.view(bk).data_ptr(), self.shift.view(bk).data_ptr(), cout_p, st),
                    'fold_bn')
            master = self.w_master.view('w%d' % i)
            if stem:
                master.copy_(w.reshape(-1))
            else:
                L.check(self.lib.comic_pack_conv_weights(w.data_ptr(), master.data_ptr(), kh, kw, cin_p, cout_p, 0, st),
                        'pack_conv_weights')
        self.refresh_weights()
        torch.cuda.synchronize()

    def refresh_weights(self):
        """Re-derive what the forward reads from the fp32 masters: the plan-dtype weight copy and
        shift = beta - mean*scale (after loading a checkpoint or an optimiser step)."""
        plan_copy = self.w_plan.data_ptr() if (self.dcode == 1 and self._x3_off is None) else None
        if self._x3_off is not None:
            self._pack_x3()
        L.check(self.lib.comic_cnn_refresh_weights(self.w_master.data.data_ptr(), plan_copy, self.w_master.numel,
                                                   self.beta.data.data_ptr(), self.mean.data.data_ptr(),
                                                   self.scale.data.data_ptr(), self.shift.data.data_ptr(),
                                                   self.beta.numel, L.stream_ptr()), 'cnn_refresh_weights')
        if self.w_frag is not None:
            L.check(self.lib.comic_cnn_pack_frag_weights(self.w_plan.data_ptr(), self.w_frag.data_ptr(),
                                                         self._frag_table.data_ptr(), self._frag_table.shape[0],
                                                         self.w_master.numel, L.stream_ptr()), 'cnn_pack_frag_weights')
        # masters changed: version stamp shared by every encoder aliasing them
        self.w_master.__dict__['_ver'] = self.w_master.__dict__.get('_ver', 0) + 1
        t = getattr(self, '_train', None)
        if t is not None:
            # the backward-data filters follow the masters: repack them on a second stream, off the
            # critical path (the next forward / decoder step run meanwhile); backward() waits for the event
            torch = self.torch
            t.aux.wait_stream(torch.cuda.current_stream())
            with torch.cuda.stream(t.aux):
                L.check(self.lib.comic_cnn_pack_bwd_filters(self._ops, len(self.plan.ops), t.grads, self.dcode,
                                                            L.stream_ptr()), 'cnn_pack_bwd_filters')
                t.filters_ev.record(t.aux)
            t.filters_ver = self.w_master.__dict__['_ver']

    def clear_grads_async(self):
        """After the optimiser has read the gradients of a step: clear the gradient buffers for the next backward on the aux
        stream (beside the next forward / decoder step) instead of at the head of that backward (85 us of fills between the
        decoder and the CNN backward).  The next backward() waits for the event."""
        t = getattr(self, '_train', None)
        if t is None:
            return
        torch = self.torch
        t.aux.wait_stream(torch.cuda.current_stream())
        with torch.cuda.stream(t.aux):
            t.gflat.zero_()
            t.dw.data.zero_()
            t.dbeta.data.zero_()
            ev = torch.cuda.Event()
            ev.record(t.aux)
        t.zero_ev = ev

    def _pack_x3(self):
        """The x3 plan copy of every conv filter from its fp32 master [Cout][Kpad]: per tap the channels
        [bf16(w) | bf16(w) | bf16(w - bf16(w))] against activations stored [hi | lo | hi] (COMIC_OP_X3); at load and after
        every optimiser step of cnn_finetune, one launch (comic_cnn_pack_x3_weights)."""
        tb = self.__dict__.get('_x3_table')
        if tb is None:          # pointer / shape tables of comic_cnn_pack_x3_weights (one launch for all convs)
            idx = [i for i, w in enumerate(self.plan.weights) if not w[5]]
            esz = self.w_plan.element_size()
            tb = self._x3_table = (
                (C.c_void_p * len(idx))(*[self.w_master.view('w%d' % i).data_ptr() for i in idx]),
                (C.c_void_p * len(idx))(*[self.w_plan.data_ptr() + esz * self._x3_off[i] for i in idx]),
                (C.c_int32 * len(idx))(*[self.plan.wphys[i][1] for i in idx]),
                (C.c_int32 * len(idx))(*[self.plan.weights[i][1] * self.plan.weights[i][2] for i in idx]),
                (C.c_int32 * len(idx))(*[self.plan.wphys[i][0] for i in idx]), len(idx))
        L.check(self.lib.comic_cnn_pack_x3_weights(tb[0], tb[1], tb[2], tb[3], tb[4], tb[5], L.stream_ptr()),
                'cnn_pack_x3_weights')

    def _unpack(self, i, flat_w, flat_b):
        """Packed master-layout buffers of weight i -> (HWIO array, per-channel vector) of the variable's shape."""
        prefix, kh, kw, cin, cout, stem = self.plan.weights[i]
        cin_p, cout_p = self.plan.wphys[i]
        K = kh * kw * cin_p
        m = flat_w.view('w%d' % i)
        w = m.view(K, cout_p) if stem else m.view(cout_p, -1)[:, :K].t()
        w = w.reshape(kh, kw, cin_p, cout_p)[:, :, :cin, :cout]
        return w.contiguous().cpu().numpy(), flat_b.view('b%d' % i)[:cout].cpu().numpy().copy()

    def _pack_into(self, i, flat_w, flat_b, w_hwio, vec):
        """Inverse of _unpack: a variable-shaped (HWIO array, per-channel vector) pair into the packed master-layout
        buffers of weight i (padding channels zero)."""
        torch = self.torch
        prefix, kh, kw, cin, cout, stem = self.plan.weights[i]
        cin_p, cout_p = self.plan.wphys[i]
        K = kh * kw * cin_p
        wp = torch.zeros((kh, kw, cin_p, cout_p), dtype=torch.float32, device=self.device)
        wp[:, :, :cin, :cout] = torch.from_numpy(np.ascontiguousarray(w_hwio, np.float32)).to(self.device)
        m = flat_w.view('w%d' % i)
        if stem:
            m.view(K, cout_p).copy_(wp.reshape(K, cout_p))
        else:
            mv = m.view(cout_p, -1)
            mv.zero_()
            mv[:, :K].copy_(wp.reshape(K, cout_p).t())
        b = flat_b.view('b%d' % i)
        b.zero_()
        b[:cout].copy_(torch.from_numpy(np.ascontiguousarray(vec, np.float32)).to(self.device))

    def export_slots(self, flat_w, flat_b, suffix):
        """Optimiser slot buffers with the masters' layout -> {`<variable name>/<suffix>`: array of the variable's shape}."""
        out = {}
        for i, (prefix, kh, kw, cin, cout, stem) in enumerate(self.plan.weights):
            w, b = self._unpack(i, flat_w, flat_b)
            out[prefix + '/weights/' + suffix], out[prefix + '/BatchNorm/beta/' + suffix] = w, b
        return out

    def import_slots(self, flat_w, flat_b, arrays, suffix, scope=''):
        """Inverse of export_slots; returns False (and loads nothing) unless every slot is present."""
        keys = [(scope + p[0] + '/weights/' + suffix, scope + p[0] + '/BatchNorm/beta/' + suffix) for p in self.plan.weights]
        if not all(a in arrays and b in arrays for a, b in keys):
            return False
        for i, (a, b) in enumerate(keys):
            self._pack_into(i, flat_w, flat_b, arrays[a], arrays[b])
        return True

    def export_params(self):
        """Trainable CNN variables back in the slim checkpoint layout: {name: HWIO weights / beta}."""
        out = {}
        for i, (prefix, kh, kw, cin, cout, stem) in enumerate(self.plan.weights):
            out[prefix + '/weights'], out[prefix + '/BatchNorm/beta'] = self._unpack(i, self.w_master, self.beta)
        return out

    def export_grads(self):
        """{variable name: gradient} of the last backward() in the slim layout."""
        t, out = self._train, {}
        for i, (prefix, kh, kw, cin, cout, stem) in enumerate(self.plan.weights):
            out[prefix + '/weights'], out[prefix + '/BatchNorm/beta'] = self._unpack(i, t.dw, t.dbeta)
        return out

    # -- cnn_finetune ------------------------------------------------------------------------------
    def enable_training(self):
        """Allocate what comic_cnn_backward needs: a gradient buffer per activation buffer (one
        flat allocation, zeroed per step), fp32 weight / beta gradients with the masters' layout,
        the backward-data filter scratch and the d-conv scratch."""
        if self._train is not None:
            return self._train
        torch, plan = self.torch, self.plan
        if plan.pool_after_projection:
            raise ValueError('cnn_finetune needs a plan built with pool_after_projection=False (a forward-only layout)')
        x3 = bool(getattr(plan, 'x3', False))
        t = type('CnnTrainState', (), {})()
        t.dw, t.dbeta = self.w_master.like(), self.beta.like()
        # gradient buffers: the plan dtype and geometry of the activation buffers; x3 plans ("bf16x3"): fp32 buffers of the
        # LOGICAL channels (an activation buffer holds three bf16 regions of them: csrc/conv.hip conv_backward_x3)
        geo, sizes, total = [], [], 0
        for bi, (H, W, Cc, f32) in enumerate(plan.buffers):
            g32 = f32 or self.dcode == 0 or x3
            Cg = Cc // 3 if (x3 and not f32) else Cc
            geo.append((Cg, g32))
            nbytes = 0 if bi == plan.input else self.batch * H * W * Cg * (4 if g32 else 2)
            sizes.append((total, nbytes))
            total += (nbytes + 255) // 256 * 256
        t.gflat = torch.zeros(total, dtype=torch.uint8, device=self.device)
        t.gbufs = []
        for bi, (H, W, Cc, f32) in enumerate(plan.buffers):
            off, nbytes = sizes[bi]
            if nbytes == 0:
                t.gbufs.append(None)
                continue
            Cg, g32 = geo[bi]
            dt = torch.float32 if g32 else self._tdt
            t.gbufs.append(t.gflat[off:off + nbytes].view(dt).view(self.batch, H, W, Cg))
        t.gptr = (C.c_void_p * len(t.gbufs))(*[g.data_ptr() if g is not None else None for g in t.gbufs])
        esz = 4 if self.dcode == 0 else 2
        wb_off, n = [], 0
        for (prefix, kh, kw, cin, cout, stem), (cin, cout) in zip(plan.weights, plan.wphys):
            wb_off.append(n)
            if not stem:      # backward-data filter [Cin][roundup64(taps * Cout)]; x3: three regions per tap
                n += (cin * ((kh * kw * cout * (3 if x3 else 1) + 63) // 64 * 64) * esz + 255) // 256 * 256
        t.w_bwd = torch.zeros(max(n, 256), dtype=torch.uint8, device=self.device)
        t.grads = (L.ConvGrad * len(plan.weights))()
        for i, (prefix, kh, kw, cin, cout, stem) in enumerate(plan.weights):
            t.grads[i].w_master = self.w_master.view('w%d' % i).data_ptr()
            t.grads[i].dw = t.dw.view('w%d' % i).data_ptr()
            t.grads[i].dbeta = t.dbeta.view('b%d' % i).data_ptr()
            t.grads[i].w_bwd = None if stem else t.w_bwd.data_ptr() + wb_off[i]
        # two lanes: every conv's d-conv tensor side by side, so the weight gradients (t.wlane) run beside the
        # backward-data chain (comic_hip.h, comic_cnn_backward)
        t.scratch_bytes = int(self.lib.comic_cnn_backward_scratch_bytes(self._ops, len(plan.ops), self.batch,
                                                                         self.dcode, 2 if self.backward_lanes else 1))
        t.scratch = torch.empty(t.scratch_bytes, dtype=torch.uint8, device=self.device)
        t.wlane = streams.lane(torch, self.device, 'wgrad') if self.backward_lanes else None
        # branch lanes of the backward (backward_schedule): a second chain stream and alternate gradient buffers of the
        # blocks' shared inputs (zero between steps: the join adds them in and clears them)
        t.sched, t.lane1 = None, None
        if self.backward_lanes and self.backward_branch_lanes and plan.name == 'inception_v3':
            sched, alt_bufs = backward_schedule(plan)
            if alt_bufs:
                t.sched = np.ascontiguousarray(sched)
                t.lane1 = streams.lane(torch, self.device, 'chain1')
                t.galt = {b: torch.zeros_like(t.gbufs[b]) for b in alt_bufs}
                t.gptr_alt = (C.c_void_p * len(t.gbufs))(*[t.galt[b].data_ptr() if b in t.galt else None
                                                           for b in range(len(t.gbufs))])
        t.aux = streams.lane(torch, self.device, 'aux')
        t.filters_ev = torch.cuda.Event()
        t.filters_ver = -1
        self._train = t
        return t

    def autotune_backward(self, reps=3, verbose=False, cache=None):
        """cnn_finetune: pick the kernel variant of every conv's BACKWARD-DATA convolution (the forward conv of the masked
        d-conv tensor with the flipped / transposed filter; comic_conv_grad.bwd_tile) by timing the candidates on the real
        buffers, as autotune() does for the forward -- the heuristic picks 32x64 / 64x64 tiles for most of them at batch
        32.  Every variant gives the same bits.  -> {weight index: (ms, tile)}."""
        if self.dcode != 1 or getattr(self.plan, 'x3', False):      # (x3: the backward-data convs run on the heuristic tiles)
            return {}
        t = self.enable_training()
        torch, plan = self.torch, self.plan
        st = L.stream_ptr()
        key = self._tune_key() + ':bwd'
        if cache and os.path.isfile(cache):
            tiles = json.load(open(cache)).get(key)
            if tiles is not None and len(tiles) == len(plan.weights):
                for i, tl in enumerate(tiles):
                    t.grads[i].bwd_tile = int(tl)
                return {i: (None, tl) for i, tl in enumerate(tiles)}
        L.check(self.lib.comic_cnn_pack_bwd_filters(self._ops, len(plan.ops), t.grads, self.dcode, st), 'cnn_pack_bwd_filters')
        ev0, ev1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        chosen = {}
        for i, o in enumerate(plan.ops):
            if o['kind'] != 0 or t.gbufs[o['src']] is None:
                continue
            wi = o['weight']
            S = o['SH']
            Hd, Wd = (o['Ho'] - 1) * S + 1, (o['Wo'] - 1) * S + 1
            gx = t.gbufs[o['src']]
            op = L.CnnOp(kind=0, src=0, dst=1, src_coff=0, dst_coff=o['src_coff'], H=Hd, W=Wd, Cin=o['Cout'], Cout=o['Cin'],
                         KH=o['KH'], KW=o['KW'], SH=1, SW=1, PT=o['KH'] - 1 - o['PT'], PL=o['KW'] - 1 - o['PL'], Ho=o['H'],
                         Wo=o['W'], weight=0, relu=0, out_f32=0, tile=0, flags=L.OP_RAW)
            wt = L.ConvWeight(t.grads[wi].w_bwd, None, None, None)
            x_ptr, y_ptr, yc = t.scratch.data_ptr(), gx.data_ptr(), gx.shape[3]

            def run():
                L.check(self.lib.comic_conv2d_bn_relu(C.byref(op), x_ptr, o['Cout'], y_ptr, yc, C.byref(wt), self.batch,
                                                      self.dcode, st), 'backward-data conv (autotune)')

            def timed(n_rep, blocks=2):
                best = None
                for _ in range(blocks):
                    ev0.record()
                    for _ in range(n_rep):
                        run()
                    ev1.record()
                    ev1.synchronize()
                    tb = ev0.elapsed_time(ev1) / n_rep
                    best = tb if best is None else min(best, tb)
                return best
            cands = []
            for tile in range(0, L.IMG_TILE):              # (no fragment-order copy of the backward filters: id 55 is out)
                if tile == L.WS_TILE:
                    continue
                op.tile = tile
                try:
                    run(); run()
                except L.ComicHipError:
                    if L.is_im2col_tile(tile):
                        raise
                    continue
                cands.append((timed(reps), tile))
            cands.sort()
            finals = []
            for _, tile in cands[:3]:
                op.tile = tile
                run()
                finals.append((timed(4 * reps, 3), tile))
            best = min(finals)
            t.grads[wi].bwd_tile = best[1]
            chosen[wi] = best
            if verbose:
                print('autotune bwd-data op %3d %3dx%-3d %4d->%4d %dx%d -> tile %2d  %.1f us (heuristic %.1f us)' % (
                    i, o['H'], o['W'], o['Cout'], o['Cin'], o['KH'], o['KW'], best[1], best[0] * 1e3,
                    dict((tl, ms) for ms, tl in cands)[0] * 1e3))
        t.gflat.zero_()
        if cache:
            db = json.load(open(cache)) if os.path.isfile(cache) else {}
            db[key] = [int(t.grads[i].bwd_tile) for i in range(len(plan.weights))]
            json.dump(db, open(cache, 'w'))
        return chosen

    def grad_buckets(self, n=6):
        """Partition of the backward pass for the data-parallel gradient exchange: runs of whole
        blocks, LAST block first (the order the backward produces gradients), of roughly equal weight bytes.
        -> [(op_lo, op_hi, (w_lo, w_hi), (b_lo, b_hi))]: op range and the element ranges of the flat dw / dbeta buffers
        that are complete once the backward of those ops has been issued."""
        return plan_grad_buckets(self.plan, self.w_master, self.beta, n)


    def backward(self, d_fm, d_im_embed, buckets=None, on_bucket=None, act_fusion=True):
        """d_fm [B, M, C] / d_im_embed [B, C_g] fp32 (the gradients of `forward`'s two outputs; either
        may be None) -> fills the weight / beta gradients of `enable_training()`'s state.
        buckets (from grad_buckets) + on_bucket(train_state, bucket): the backward is issued bucket by bucket and the
        callback runs after each one -- the data-parallel step starts that bucket's all-reduce there, under the
        backward of the earlier blocks.  act_fusion=False: THIS call of the scheduled backward runs the unfused chain
        (COMIC_CNN_BWD_NO_ACT_FUSION; A/B timing, parity tests)."""
        t = self.enable_training()
        if getattr(t, 'zero_ev', None) is not None:          # cleared on the aux stream behind the last optimiser step
            self.torch.cuda.current_stream().wait_event(t.zero_ev)
            t.zero_ev = None
        else:
            t.gflat.zero_()
            t.dw.data.zero_()
            t.dbeta.data.zero_()
        if d_fm is not None:
            g = t.gbufs[self.plan.fm if self.plan.fm is not None else self.plan.fm_src]
            g.view(-1).copy_(d_fm.reshape(-1))              # converts to the buffer's dtype
        if d_im_embed is not None:
            t.gbufs[self.plan.pooled].view(-1).copy_(d_im_embed.reshape(-1))
        ready = t.filters_ver == self.w_master.__dict__.get('_ver', 0)   # else: packed inline by the executor
        if ready:
            self.torch.cuda.current_stream().wait_event(t.filters_ev)
        if buckets is None and on_bucket is None and t.sched is not None:
            L.check(self.lib.comic_cnn_backward_sched(self._ops, len(self.plan.ops), t.sched.ctypes.data, len(t.sched),
                                                      self._last_bufptr or self._bufptr, t.gptr, t.gptr_alt, self._bufch,
                                                      self._wt, t.grads, self.batch, self.dcode,
                                                      int(ready) | (0 if act_fusion else L.CNN_BWD_NO_ACT_FUSION),
                                                      t.scratch.data_ptr(), t.scratch_bytes, L.stream_ptr(),
                                                      t.lane1.cuda_stream, t.wlane.cuda_stream), 'cnn_backward_sched')
            return t
        for bk in (buckets or [(0, len(self.plan.ops), None, None)]):
            lo, hi = bk[0], bk[1]
            first = C.byref(self._ops, lo * C.sizeof(L.CnnOp))
            L.check(self.lib.comic_cnn_backward(first, hi - lo, self._last_bufptr or self._bufptr, t.gptr, self._bufch, self._wt,
                                                t.grads, self.batch, self.dcode, int(ready), t.scratch.data_ptr(),
                                                t.scratch_bytes, L.stream_ptr(),
                                                t.wlane.cuda_stream if t.wlane is not None else None), 'cnn_backward')
            if on_bucket is not None:
                on_bucket(t, bk)
        return t

    def _build_group_args(self):
        """(Re)build the device-resident argument records of the grouped launches; they embed
        buffer addresses and the tile-dependent workgroup ranges, so this follows any change
        of `tile` ids."""
        torch = self.torch
        n = self.lib.comic_cnn_group_args_bytes(self._ops, len(self.plan.ops))
        if n == 0:
            self._group_args = None
            return
        host = np.zeros(n, np.uint8)
        L.check(self.lib.comic_cnn_build_group_args(self._ops, len(self.plan.ops), self._bufptr, self._bufch,
                                                    self._wt, self.batch, host.ctypes.data), 'cnn_build_group_args')
        if self._group_args is None or self._group_args.numel() != n:
            self._group_args = torch.empty(n, dtype=torch.uint8, device=self.device)
        self._group_args.copy_(torch.from_numpy(host))
        torch.cuda.synchronize()

    def _drop_graphs(self):
        """Captured graphs hold launch parameters: drop them (the next calls capture again)."""
        self._graph, self._calls = None, 0
        for a in self._alt.values():
            a[1], a[2] = None, 0

    def _run(self, bufptr=None):
        bufptr = bufptr or self._bufptr
        if self._group_args is not None:
            L.check(self.lib.comic_cnn_forward_grouped(self._ops, len(self.plan.ops), bufptr, self._bufch,
                                                       self._wt, self.batch, self.dcode,
                                                       self._group_args.data_ptr(), L.stream_ptr()),
                    'cnn_forward_grouped')
            return
        L.check(self.lib.comic_cnn_forward(self._ops, len(self.plan.ops), bufptr, self._bufch, self._wt,
                                           self.batch, self.dcode, L.stream_ptr()), 'cnn_forward')

    def _adopt(self, images):
        """Pointer table (and graph slot) for a forward that reads `images` IN PLACE instead of from a copy in the plan's
        input buffer (385 MB of device-to-device traffic per forward of 640 images).  Possible when only ungrouped ops read
        the input (grouped launches keep their buffer addresses in device records); at most four distinct tensors are
        adopted -- a caller that hands over a fresh tensor every time gets the copy."""
        key = images.data_ptr()
        a = self._alt.get(key)
        if a is not None:
            return a
        if len(self._alt) >= 4 or not images.is_contiguous() or self._train is not None:
            return None
        if any(o['src'] == self.plan.input and o.get('group', 0) for o in self.plan.ops):
            return None
        bp = (C.c_void_p * len(self.bufs))(*[b.data_ptr() for b in self.bufs])
        bp[self.plan.input] = key
        a = self._alt[key] = [bp, None, 0, images]
        return a

    def forward(self, images, use_graph=False):
        """images fp32 NHWC [B,H,W,3] in [-1,1] (device) -> (im_embed [B,C_g], fmaps [B,M,C]) fp32.
        ModelBase._encoder, non-legacy (model_base.py:93-104).  use_graph: the ~130 launches of
        the plan (incl. the fork/join of the branch streams) are replayed from a hipGraph that
        is captured on the second call."""
        inp = self.bufs[self.plan.input]
        assert images.shape == inp.shape and images.dtype == inp.dtype, (images.shape, inp.shape)
        alt = None
        if images.data_ptr() != inp.data_ptr():
            alt = self._adopt(images)
            if alt is None:
                inp.copy_(images)
        bufptr, graph, calls = (alt[0], alt[1], alt[2]) if alt is not None else (None, self._graph, self._calls)
        self._last_bufptr = bufptr      # backward() reads the activations (the input among them) where this forward did
        if use_graph and graph is None and calls >= 1:
            graph = self.torch.cuda.CUDAGraph()
            # thread_local: the input pipeline's prefetch thread allocates, copies and launches on this device
            # meanwhile; only the capturing thread is held to the capture rules
            with self.torch.cuda.graph(graph, capture_error_mode='thread_local'):
                self._run(bufptr)
            if alt is not None:
                alt[1] = graph
            else:
                self._graph = graph
        if use_graph and graph is not None:
            graph.replay()
        else:
            self._run(bufptr)
        if alt is not None:
            alt[2] += 1
        else:
            self._calls += 1
        pooled = self.bufs[self.plan.pooled]
        B = self.batch
        if self.plan.fm is None:           # inner end point (Inception-V1 Mixed_4f): fp32 copy for the decoder
            src = self.bufs[self.plan.fm_src]
            if self._fm_f32 is None:
                self._fm_f32 = self.torch.empty(src.shape, dtype=self.torch.float32, device=self.device)
            self._fm_f32.copy_(src)
            fm = self._fm_f32
        else:
            fm = self.bufs[self.plan.fm]
        return pooled.reshape(B, -1), fm.reshape(B, fm.shape[1] * fm.shape[2], fm.shape[3])

    def _tune_key(self):
        p = self.plan
        H, W = p.buffers[p.input][:2]
        return '%s:%dx%d:B%d:%s%s:%dops:polite%d' % (p.name, H, W, self.batch, 'par' if p.pool_after_projection else 'plain',
                                                    ('+fp' if getattr(p, 'fuse_pools', False) else '') +
                                                    ('+ch' if getattr(p, 'fuse_chains', False) else '') +
                                                    ('+x3' if getattr(p, 'x3', False) else ''),
                                                  len(p.ops), self.polite_lds_kb)

    def autotune(self, reps=5, verbose=False, cache=None):
        """Pick the fastest tile / pipeline-depth variant of the conv kernels for every conv of the plan
        at this batch size (times each variant with HIP events on the real buffers; ~0.5 s).  Results
        are bit-identical across variants: the k order per accumulator does not depend on the tile.
        cache: path of a JSON file {plan key: [tile id per op]}: loaded when it holds this plan (no timing
        runs -- e.g. under a profiler), written after a tuning run."""
        if self.dcode != 1:
            return {}
        torch = self.torch
        st = L.stream_ptr()
        if cache and os.path.isfile(cache):
            tiles = json.load(open(cache)).get(self._tune_key())
            if tiles is not None and len(tiles) == len(self.plan.ops):
                for i, t in enumerate(tiles):
                    self._ops[i].tile = int(t)
                if self._group_args is not None:
                    self._build_group_args()
                self._drop_graphs()
                return {i: (None, t) for i, t in enumerate(tiles)}
        chosen = self._autotune(reps, verbose, torch, st)      # op.min_lds: tuned under the occupancy the forward runs with
        if cache:
            db = json.load(open(cache)) if os.path.isfile(cache) else {}
            db[self._tune_key()] = [int(self._ops[i].tile) for i in range(len(self.plan.ops))]
            json.dump(db, open(cache, 'w'))
        return chosen

    def _autotune(self, reps, verbose, torch, st):
        chosen = {}
        ev0, ev1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        n_ops = len(self.plan.ops)
        grouped = self._group_args is not None
        rec_bytes = 0
        if grouped:
            n_rec = sum(1 for j in range(n_ops) if self._ops[j].group > 0)
            rec_bytes = self.lib.comic_cnn_group_args_bytes(self._ops, n_ops) // n_rec
        rec = 0
        i = 0
        while i < n_ops:
            o = self.plan.ops[i]
            if o['kind'] != 0:
                i += 1
                continue
            op = self._ops[i]
            if o.get('flags', 0) & L.OP_POOLED_SRC or o.get('tile') == L.CHAIN_TILE:   # one kernel serves these (conv_ws.hip / the chain kernel): nothing to choose
                n = 1
                while i + n < n_ops and op.group > 0 and self._ops[i + n].group == op.group:
                    n += 1
                rec += n if (grouped and op.group > 0) else 0
                i += n
                continue
            n = 1
            if grouped and op.group > 0:
                while i + n < n_ops and self._ops[i + n].group == op.group:
                    n += 1
                first = C.byref(self._ops, i * C.sizeof(L.CnnOp))
                rec_off = rec * rec_bytes

                def run():
                    L.check(self.lib.comic_cnn_forward_grouped(first, n, self._bufptr, self._bufch, self._wt,
                                                               self.batch, self.dcode,
                                                               self._group_args.data_ptr() + rec_off, st),
                            'grouped conv (autotune)')
                rec += n
            else:
                src, dst = self.bufs[o['src']], self.bufs[o['dst']]
                wt = self._wt[o['weight']]

                def run():
                    L.check(self.lib.comic_conv2d_bn_relu(C.byref(op), src.data_ptr(), src.shape[3], dst.data_ptr(),
                                                          dst.shape[3], C.byref(wt), self.batch, self.dcode, st),
                            'conv (autotune)')
            def timed(n_rep, blocks=2):
                t = None
                for _ in range(blocks):   # best of several timed blocks: variants are often within the run-to-run jitter
                    ev0.record()
                    for _ in range(n_rep):
                        run()
                    ev1.record()
                    ev1.synchronize()
                    tb = ev0.elapsed_time(ev1) / n_rep
                    t = tb if t is None else min(t, tb)
                return t
            cands = []
            for tile in range(0, L.CONV_TILES + 1):
                op.tile = tile
                try:
                    if n > 1:
                        self._build_group_args()
                    run(); run()
                except L.ComicHipError:
                    if L.is_im2col_tile(tile):
                        raise
                    continue              # a patch-resident variant this layer (or a group member) is not eligible for
                cands.append((timed(reps), tile))
            # the minimum over ~50 noisy measurements favours lucky ones: the three fastest are timed again, longer
            cands.sort()
            if verbose and os.environ.get('COMIC_TUNE_DUMP') == '1':      # every candidate of the launch, fastest first
                print('   candidates op %3d: %s' % (i, ' '.join('%d:%.0f' % (t, us * 1e3) for us, t in cands[:14])))
            finals = []
            for _, tile in cands[:3]:
                op.tile = tile
                if n > 1:
                    self._build_group_args()
                run()
                finals.append((timed(4 * reps, 3), tile))
            best = min(finals)
            op.tile = best[1]
            chosen[i] = best
            if verbose:
                print('autotune op %3d x%d %3dx%-3d Cin%4d Cout%4d %dx%d -> tile %2d  %.1f us' % (
                    i, n, o['Ho'], o['Wo'], o['Cin'], o['Cout'], o['KH'], o['KW'], best[1], best[0] * 1e3))
            i += n
        if grouped:
            self._build_group_args()
        self._drop_graphs()       # a captured graph holds the old variants
        self._calls = 0
        return chosen

    def end_point(self, name):
        b = self.bufs[self.plan.end_points[name]]
        if getattr(self.plan, 'x3', False) and b.dtype != self.torch.float32:
            C3 = b.shape[-1] // 3                   # [hi | lo | hi] regions (COMIC_OP_X3): the value is hi + lo
            return b[..., :C3].float() + b[..., C3:2 * C3].float()
        return b

    @property
    def flops_per_image(self):
        return 2 * self.plan.macs


def get_network_fn(name, num_classes=None, weight_decay=0.0, is_training=False):
    """Signature-compatible with nets_factory.get_network_fn for the supported backbone;
    returns a builder of `CnnPlan` (the graph object of this framework)."""
    if num_classes:
        raise NotImplementedError('classification heads are outside the captioning hot path')
    if is_training:
        raise NotImplementedError('the reference always builds the CNN with is_training=False (model_base.py:76)')

    def network_fn(image_size=(224, 224), final_endpoint=None, pool_after_projection=False, fuse_pools=False, x3=False):
        par = pool_after_projection and name == 'inception_v3'
        return CnnPlan(name, image_size, final_endpoint or ('Mixed_4f' if name == 'inception_v1' else 'Mixed_7c'),
                       pool_after_projection=par, fuse_pools=fuse_pools and par and not x3, x3=x3)
    network_fn.default_image_size = 224 if name == 'inception_v1' else 299
    return network_fn
