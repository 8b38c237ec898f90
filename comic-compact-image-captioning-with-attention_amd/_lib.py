"""ctypes binding of include/comic_hip.h (the drop-in boundary).

Fails loudly when the HIP library is missing or a symbol is absent: the product has no
CPU fallback."""
import ctypes as C
import os

_HERE = os.path.dirname(os.path.abspath(__file__))
# COMIC_HIP_LIB: another build of the same library (A/B timing of two kernel versions on one box, tools/ only)
LIB_PATH = os.environ.get('COMIC_HIP_LIB') or os.path.join(_HERE, 'lib', 'libcomic_hip.so')

c_void_p, c_int, c_int32, c_int64, c_float, c_double, c_char_p, c_uint64 = (
    C.c_void_p, C.c_int, C.c_int32, C.c_int64, C.c_float, C.c_double, C.c_char_p, C.c_uint64)


class CnnOp(C.Structure):
    _fields_ = [(n, c_int32) for n in (
        'kind', 'src', 'dst', 'src_coff', 'dst_coff', 'H', 'W', 'Cin', 'Cout', 'KH', 'KW', 'SH', 'SW',
        'PT', 'PL', 'Ho', 'Wo', 'weight', 'relu', 'out_f32', 'src_f32', 'lane', 'tile', 'group', 'flags', 'min_lds')]


class ImageDesc(C.Structure):           # struct comic_image_desc
    _fields_ = [('offset', C.c_int64), ('in_h', c_int32), ('in_w', c_int32), ('flip', c_int32), ('oy', c_int32),
                ('ox', c_int32), ('sy', C.c_float), ('sx', C.c_float)]


class ConvWeight(C.Structure):
    _fields_ = [('w', c_void_p), ('scale', c_void_p), ('shift', c_void_p), ('w_frag', c_void_p)]


class AttnDesc(C.Structure):
    _fields_ = [(n, c_int32) for n in ('B', 'M', 'D', 'H', 'Cv', 'method', 'prob', 'tied')]


class DecoderDesc(C.Structure):
    _fields_ = [(n, c_int32) for n in (
        'D', 'E', 'A', 'V', 'C', 'Cg', 'H', 'M', 'Cv', 'fm_projection', 'method', 'prob', 'context_layer',
        'init_method', 'start_id', 'end_id')] + [(n, c_float) for n in (
            'keep_in', 'keep_out', 'keep_alpha', 'map_loss_scale')] + [('flags', C.c_uint32),
                                                                       ('length_penalty_weight', c_float),
                                                                       ('cell', c_int32)]


CELLS = {'LSTM': 0, 'LN_LSTM': 1, 'GRU': 2}          # include/comic_hip.h COMIC_CELL_*


# comic_decoder_desc.flags (include/comic_hip.h COMIC_DEC_*).  The library reads no environment: the A/B switches of
# the decoder executors are environment variables of the PYTHON side, read at every call (tests flip them inside one
# process) by decoder_flags_from_env() and handed over in the descriptor.
(DEC_NO_PERSIST, DEC_NO_PERSIST_BWD, DEC_NO_FUSED_STEP, DEC_NO_SPLIT_ATTN_BWD, DEC_ONE_LANE, DEC_EXACT_GEMM, DEC_STAMPS,
 DEC_NO_BEAM_LOGITS, DEC_NO_LSTM_STREAM) = (1, 2, 4, 8, 16, 32, 64, 128, 256)
DEC_PHASE_FWD, DEC_PHASE_BWD = 512, 1024       # comic_decoder_train_step in two calls (Decoder.train_step(phase=...))
DEC_NO_GROUP_GEMM = 2048
DEC_BWD_OWN_ROWS = 4096
DEC_INJECT_TIMEOUT = 8192       # per-call fault injection (tests)
CNN_BWD_NO_ACT_FUSION = 2       # comic_cnn_backward_sched: bit 1 of filters_ready
_DEC_ENV = (('COMIC_PERSIST', '0', DEC_NO_PERSIST), ('COMIC_PERSIST_BWD', '0', DEC_NO_PERSIST_BWD),
            ('COMIC_FUSED_STEP', '0', DEC_NO_FUSED_STEP), ('COMIC_SPLIT_ATTN_BWD', '0', DEC_NO_SPLIT_ATTN_BWD),
            ('COMIC_GRAD_LANES', '0', DEC_ONE_LANE), ('COMIC_SPLIT3', '0', DEC_EXACT_GEMM),
            ('COMIC_PERSIST_STAMPS', '1', DEC_STAMPS), ('COMIC_BEAM_LOGITS', '0', DEC_NO_BEAM_LOGITS),
            ('COMIC_LSTM_STREAM', '0', DEC_NO_LSTM_STREAM), ('COMIC_GROUP_GEMM', '0', DEC_NO_GROUP_GEMM),
            ('COMIC_BWD_OWN_ROWS', '1', DEC_BWD_OWN_ROWS))


def decoder_flags_from_env():
    f = 0
    for name, val, bit in _DEC_ENV:
        if os.environ.get(name, '')[:1] == val:
            f |= bit
    return f


CONV_TILES = 61          # 1..12 im2col LDS-DMA variants, 13..25 patch-resident variants, 26..47 wide two-stage im2col variants, 56..61 walk forms
WS_TILE = 54             # weight-stationary 1x1 groups (csrc/conv_ws.hip)
IMG_TILE = 55            # image-resident stride-1 convs on 25x25 / 12x12 / 5x5 maps (csrc/conv_img.hip)
CHAIN_TILE = 62          # a group of one or two CHAINS of image-resident convs (OP_CHAIN_LINK), one launch (conv_img_chain_kernel)
OP_RAW, OP_POOLED_SRC, OP_X3 = 1, 2, 4      # COMIC_OP_X3: [hi | lo | hi] channel regions of a bf16x3 plan
OP_CHAIN_LINK = 8        # the conv's output goes to the next op of the table through the LDS (its dst buffer is not written)
OP_CHAIN_KEEP = 16       # ... and to its dst buffer as well (trainable plans: the backward reads it)
IM2COL_CONV_TILES = 12   # 13..25 are the patch-resident variants (stride-1 layers whose input window fits the LDS)


def is_im2col_tile(tile):
    """Every layer is eligible for these ids (a failure is an error); the patch-resident ids may refuse a layer."""
    return tile <= IM2COL_CONV_TILES or 26 <= tile <= 47        # 48..53: patch-resident with loader waves
PARAM_NAMES = ('W_init', 'K', 'b', 'W_m', 'W_v', 'W_q', 'v', 'ln_g', 'ln_b', 'tau', 'W_a', 'W_o', 'b_o', 'emb', 'cell_ln', 'K_c', 'b_c', 'status')


class DecoderParams(C.Structure):
    _fields_ = [(n, c_void_p) for n in PARAM_NAMES]


class GemmProb(C.Structure):         # struct comic_gemm_prob
    _fields_ = [(n, c_void_p) for n in ('A', 'B', 'C', 'bias', 'mask')] + [(n, c_int32) for n in (
        'M', 'N', 'K', 'lda', 'ldb', 'ldc', 'ld_mask')] + [(n, c_float) for n in ('alpha', 'beta', 'keep')] + [
        ('type', c_int32), ('ones_a', c_int32)]


class ConvGrad(C.Structure):         # struct comic_conv_grad
    _fields_ = [('w_master', c_void_p), ('dw', c_void_p), ('dbeta', c_void_p), ('w_bwd', c_void_p),
                ('bwd_tile', c_int32), ('reserved', c_int32)]


P = c_void_p
_SIGS = {
    'comic_last_error': (c_char_p, []),
    'comic_abi_version': (c_int, []),
    'comic_device_count': (c_int, []),
    'comic_pack_conv_weights': (c_int, [P, P, c_int, c_int, c_int, c_int, c_int, P]),
    'comic_fold_bn': (c_int, [P, P, P, c_float, P, P, c_int, P]),
    'comic_cnn_forward': (c_int, [P, c_int, P, P, P, c_int, c_int, P]),
    'comic_cnn_group_args_bytes': (C.c_long, [P, c_int]),
    'comic_cnn_build_group_args': (c_int, [P, c_int, P, P, P, c_int, P]),
    'comic_cnn_forward_grouped': (c_int, [P, c_int, P, P, P, c_int, c_int, P, P]),
    'comic_clip_by_norm': (c_int, [P, P, P, c_int, c_float, c_float, c_float, P, P, P]),
    'comic_cnn_backward_scratch_bytes': (c_int64, [P, c_int, c_int, c_int, c_int]),
    'comic_cnn_backward': (c_int, [P, c_int, P, P, P, P, P, c_int, c_int, c_int, P, c_int64, P, P]),
    'comic_cnn_backward_sched': (c_int, [P, c_int, P, c_int, P, P, P, P, P, P, c_int, c_int, c_int, P, c_int64, P, P, P]),
    'comic_cnn_pack_bwd_filters': (c_int, [P, c_int, P, c_int, P]),
    'comic_cnn_pack_x3_weights': (c_int, [P, P, P, P, P, c_int, P]),
    'comic_cnn_refresh_weights': (c_int, [P, P, c_int64, P, P, P, P, c_int64, P]),
    'comic_cnn_pack_frag_weights': (c_int, [P, P, P, c_int, c_int64, P]),
    'comic_crc32c': (C.c_uint32, [P, C.c_size_t, C.c_uint32]),
    'comic_conv2d_bn_relu': (c_int, [P, P, c_int, P, c_int, P, c_int, c_int, P]),
    'comic_gemm_f32': (c_int, [P, P, P, P, c_int, c_int, c_int, c_int, c_int, c_int, c_int, c_int, c_float,
                               c_float, P]),
    'comic_gemm_f32_split3': (c_int, [P, P, P, P, c_int, c_int, c_int, c_int, c_int, c_int, c_int, c_int, c_float,
                                      c_float, P, c_int64, P]),
    'comic_gemm_f32_splitk': (c_int, [P, P, P, P, c_int, c_int, c_int, c_int, c_int, c_int, c_int, c_int, c_float,
                                      c_float, P, c_int64, P]),
    'comic_gemm_group_workspace': (c_int64, [P, c_int]),
    'comic_gemm_group': (c_int, [P, c_int, P, c_int64, P]),
    'comic_embed_fwd': (c_int, [P, P, P, c_int, c_int, c_int, P]),
    'comic_embed_bwd': (c_int, [P, P, P, c_int, c_int, c_int, P]),
    'comic_dropout_apply': (c_int, [P, P, c_float, P, c_int64, P]),
    'comic_dropout_mask': (c_int, [P, c_int64, c_float, c_uint64, c_uint64, P]),
    'comic_dropout_mask_dev': (c_int, [P, c_int64, c_float, P, c_uint64, P]),
    'comic_dropout_masks4_dev': (c_int, [P, P, P, P, P]),
    'comic_image_preprocess': (c_int, [P, P, c_int, P, c_int, c_int, c_int, P]),
    'comic_jpeg_pixels': (c_int, [P, P, c_int, c_int, c_int, c_int, P, P, P]),
    'comic_jpeg_preprocess': (c_int, [P, P, c_int, c_int, P, P, P, P, c_int, c_int, c_int, P]),
    'comic_jpeg_preprocess_packed': (c_int, [P, P, c_int, c_int, P, P, P, P, c_int, c_int, c_int, P]),
    'comic_weighted_sum_tb': (c_int, [P, P, c_int, c_int, P, P]),
    'comic_lstm_gates_fwd': (c_int, [P, P, P, P, P, P, P, P, c_float, P, c_int, P, P, c_int, c_int, P]),
    'comic_lstm_gates_bwd': (c_int, [P, P, P, P, P, c_float, P, c_int, P, P, P, c_int, c_int, P]),
    'comic_attn_step_fwd': (c_int, [P, P, P, P, P, P, P, P, P, c_float, P, P, P, P]),
    'comic_attn_step_bwd': (c_int, [P, P, P, P, P, P, P, P, P, P, c_float, P, P, P, P, P, P, P]),
    'comic_xent_fwd_bwd': (c_int, [P, P, P, P, P, P, P, P, c_int, c_int, c_int, P]),
    'comic_argmax_rows': (c_int, [P, P, c_int, c_int, P]),
    'comic_beam_step': (c_int, [P, P, P, P, P, P, P, c_int, c_int, c_int, c_int, P]),
    'comic_gather_rows': (c_int, [P, P, P, c_int, c_int, c_int, P]),
    'comic_gather_tree': (c_int, [P, P, P, P, c_int, c_int, c_int, c_int, P]),
    'comic_adam_tf': (c_int, [P, P, P, P, c_int64, c_float, c_float, c_float, c_float, c_float, c_float, P]),
    'comic_adam_tf_gated': (c_int, [P, P, P, P, c_int64, c_float, c_float, c_float, c_float, c_float, c_float, P, P]),
    'comic_momentum_tf_gated': (c_int, [P, P, P, c_int64, c_float, c_float, c_float, c_float, P, P]),
    'comic_debug_occupy_cus': (c_int, [c_int, c_int, P]),
    'comic_colsum': (c_int, [P, P, c_int, c_int, c_float, P]),
    'comic_ln_tanh_fwd': (c_int, [P, P, P, P, P, c_int, c_int, c_float, P]),
    'comic_ln_tanh_bwd_rows': (c_int, [P, P, P, P, P, c_int, c_int, P]),
    'comic_momentum_tf': (c_int, [P, P, P, c_int64, c_float, c_float, c_float, c_float, P]),
    'comic_axpy': (c_int, [P, P, c_float, c_int64, P]),
    'comic_decoder_train_workspace': (c_int64, [P, c_int, c_int]),
    'comic_decoder_train_path': (c_int, []),
    'comic_decoder_greedy_path': (c_int, []),
    'comic_decoder_beam_path': (c_int, []),
    'comic_beam_step_dense_workspace': (c_int64, [c_int, c_int, c_int, c_int]),
    'comic_beam_step_dense': (c_int, [c_void_p] * 9 + [c_int] * 5 + [c_void_p, c_int64, c_void_p]),
    'comic_gemm_f32_stream_workspace': (c_int64, [c_int, c_int, c_int]),
    'comic_gemm_f32_stream': (c_int, [c_void_p, c_void_p, c_void_p, c_void_p, c_int, c_int, c_int, c_void_p, c_int64, c_void_p]),
    'comic_decoder_infer_workspace': (c_int64, [P, c_int, c_int]),
    'comic_decoder_train_step': (c_int, [P, P, P, P, P, P, P, P, P, P, c_int, c_int, c_int, P, P, P, P, P, P, P,
                                         P, P, P, P, P, c_int64, P]),
    'comic_decoder_greedy': (c_int, [P, P, P, P, c_int, c_int, P, P, P, P, P, c_int64, P]),
    'comic_decoder_sample': (c_int, [P, P, P, P, c_int, c_int, P, P, P, P, P, P, c_int64, P]),
    'comic_decoder_beam': (c_int, [P, P, P, P, c_int, c_int, c_int, P, P, P, P, P, P, P, P, c_int64, P]),
    'comic_scorer_create': (c_void_p, [c_char_p, P, c_int64, c_double]),
    'comic_scorer_destroy': (None, [c_void_p]),
    'comic_scorer_score': (c_int, [c_void_p, P, c_int, P, P, P, P, c_int]),
}
EXPORTED_SYMBOLS = tuple(_SIGS)

_lib = None


class ComicHipError(RuntimeError):
    pass


def load():
    """Load libcomic_hip.so (built by __graft_entry__.build() / `make -C csrc`)."""
    global _lib
    if _lib is not None:
        return _lib
    if not os.path.exists(LIB_PATH):
        raise ComicHipError(
            'HIP library not built: %s is missing (run `python -c "import __graft_entry__ as g; g.build()"`). '
            'There is no CPU fallback.' % LIB_PATH)
    # torch first: its wheel carries its own libamdhip64.so, and the process must end up with ONE HIP runtime.  Loaded
    # after torch, libcomic_hip.so binds to the runtime torch brought in; loaded before it, /opt/rocm's copy comes in,
    # torch adds its own and this library's kernels are registered with a runtime that never sees the device
    # ("no ROCm-capable device is detected" at the first launch).
    try:
        import torch  # noqa: F401
    except ImportError:
        pass
    lib = C.CDLL(LIB_PATH)
    for name, (res, args) in _SIGS.items():
        try:
            fn = getattr(lib, name)
        except AttributeError:
            raise ComicHipError('libcomic_hip.so does not export %s' % name)
        fn.restype = res
        fn.argtypes = args
    if lib.comic_abi_version() != 1:
        raise ComicHipError('ABI version mismatch')
    _lib = lib
    return lib


# ---- libcomic_jpeg.so: host half of the split JPEG decoder (include/comic_jpeg.h; plain C, no GPU runtime) ----------------
JPEG_LIB_PATH = os.path.join(os.path.dirname(LIB_PATH), 'libcomic_jpeg.so')
JPEG_OK, JPEG_UNSUPPORTED, JPEG_CORRUPT, JPEG_TOO_SMALL, JPEG_IO = 0, 1, -1, -2, -3


class JpegInfo(C.Structure):
    """comic_jpeg_info (include/comic_jpeg.h), 512 bytes; the device reads the same records."""
    _fields_ = [('width', C.c_int32), ('height', C.c_int32), ('ncomp', C.c_int32), ('hmax', C.c_int32), ('vmax', C.c_int32),
                ('mcus_x', C.c_int32), ('mcus_y', C.c_int32), ('restart_interval', C.c_int32),
                ('blocks_w', C.c_int32 * 3), ('blocks_h', C.c_int32 * 3), ('comp_w', C.c_int32 * 3), ('comp_h', C.c_int32 * 3),
                ('coef_off', C.c_int64 * 3), ('coef_count', C.c_int64), ('coef_base', C.c_int64), ('pixel_off', C.c_int64),
                ('quant', (C.c_uint16 * 64) * 3)]


# the same record as a numpy structured dtype (vectorised access to a batch's records)
JPEG_INFO_DTYPE = [('width', '<i4'), ('height', '<i4'), ('ncomp', '<i4'), ('hmax', '<i4'), ('vmax', '<i4'), ('mcus_x', '<i4'),
                   ('mcus_y', '<i4'), ('restart_interval', '<i4'), ('blocks_w', '<i4', 3), ('blocks_h', '<i4', 3),
                   ('comp_w', '<i4', 3), ('comp_h', '<i4', 3), ('coef_off', '<i8', 3), ('coef_count', '<i8'),
                   ('coef_base', '<i8'), ('pixel_off', '<i8'), ('quant', '<u2', (3, 64))]

_JPEG_SIGS = {
    'comic_jpeg_read_header': (c_int, [P, C.c_int64, P]),
    'comic_jpeg_decode_coefficients': (c_int, [P, C.c_int64, P, P]),
    'comic_jpeg_decode_file': (c_int, [C.c_char_p, P, P, C.c_int64]),
    'comic_jpeg_pool_create': (P, [c_int]),
    'comic_jpeg_pool_destroy': (None, [P]),
    'comic_jpeg_pool_submit': (P, [P, P, c_int, P, P, P, C.c_int64]),
    'comic_jpeg_pool_submit_packed': (P, [P, P, c_int, P, P, P, C.c_int64]),
    'comic_jpeg_pool_wait': (c_int, [P, P, C.c_double, P, P]),
    'comic_jpeg_pool_enable_cache': (c_int, [P, C.c_int64]),
    'comic_jpeg_pool_cache_stats': (c_int, [P, P, P, P]),
}
JPEG_EXPORTED_SYMBOLS = tuple(_JPEG_SIGS)
_jpeg_lib = None


def load_jpeg():
    """Load libcomic_jpeg.so (built by `make -C csrc`).  Never touches the GPU runtime."""
    global _jpeg_lib
    if _jpeg_lib is not None:
        return _jpeg_lib
    if not os.path.exists(JPEG_LIB_PATH):
        raise ComicHipError('JPEG entropy library not built: %s is missing (run `make -C csrc`)' % JPEG_LIB_PATH)
    assert C.sizeof(JpegInfo) == 512
    lib = C.CDLL(JPEG_LIB_PATH)
    for name, (res, args) in _JPEG_SIGS.items():
        try:
            fn = getattr(lib, name)
        except AttributeError:
            raise ComicHipError('libcomic_jpeg.so does not export %s' % name)
        fn.restype = res
        fn.argtypes = args
    _jpeg_lib = lib
    return lib


def check(rc, what=''):
    if rc != 0:
        msg = load().comic_last_error()
        raise ComicHipError('%s failed (rc=%d): %s' % (what or 'comic_hip call', rc, msg.decode() if msg else ''))


def ptr(t):
    """Device/host pointer of a torch tensor (None -> NULL)."""
    if t is None:
        return None
    assert t.is_contiguous(), 'tensor must be contiguous'
    return t.data_ptr()


def stream_ptr():
    import torch
    return torch.cuda.current_stream().cuda_stream
