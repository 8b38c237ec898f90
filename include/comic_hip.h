/*
 * comic_hip.h -- C-ABI of the MI355X (gfx950) hot path of COMIC image captioning.
 *
 * The reference (jiahuei/COMIC-Compact-Image-Captioning-with-Attention) has NO FFI:
 * its hot path is a Python operator API over the TensorFlow-1.9 runtime.  Each entry
 * point below therefore replaces the TF op call-sites of one reference function and
 * cites it (file:line relative to the reference root).  INTEGRATION.md shows the
 * ctypes binding a maintainer of the reference would add.
 *
 * Conventions
 *   - every function returns 0 on success, non-zero on error; comic_last_error()
 *     returns a thread-local description.  No exceptions cross the boundary.
 *   - all pointers are DEVICE pointers unless the name ends in _host; the caller
 *     owns every buffer (the library never allocates, frees or retains pointers).
 *   - `stream` is a hipStream_t passed as void*; calls are asynchronous on it and
 *     safe to capture in a hipGraph (no allocation / synchronisation inside).
 *   - activations NHWC; conv weights packed [Cout][Kpad] with K = (kh, kw, cin)
 *     contiguous (see comic_pack_conv_weights); dense weights row-major [in][out]
 *     (TensorFlow layout, so checkpoints map 1:1).
 *   - dtype codes: COMIC_F32 = 0, COMIC_BF16 = 1.
 */
#ifndef COMIC_HIP_H_
#define COMIC_HIP_H_

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define COMIC_F32 0
#define COMIC_BF16 1
#define COMIC_ABI_VERSION 1
#define COMIC_CONV_TILES 61
#define COMIC_WS_TILE 54     /* weight-stationary 1x1 group kernel (csrc/conv_ws.hip) */
#define COMIC_IMG_TILE 55    /* image-resident kernel for stride-1 SAME convs on small maps (csrc/conv_img.hip) */
#define COMIC_CHAIN_TILE 62  /* the grouped ops are one or two CHAINS of image-resident convs (COMIC_OP_CHAIN_LINK): one launch */

const char* comic_last_error(void);
int comic_abi_version(void);
/* number of HIP devices visible; does not create a context */
int comic_device_count(void);

/* ------------------------------------------------------------------------- */
/* CNN encoder  (common/nets/inception_v3.py:100-415, inception_utils.py:32-82) */
/* ------------------------------------------------------------------------- */

/* Repack TF HWIO conv weights [kh][kw][cin][cout] (fp32) to [cout][Kpad] in `dtype`,
 * K = kh*kw*cin, Kpad = K rounded up to 64 elements, zero padded.
 * Replaces nothing in the reference (layout prep done once at checkpoint load). */
int comic_pack_conv_weights(const float* w_hwio, void* w_packed, int kh, int kw, int cin, int cout,
                            int dtype, void* stream);

/* Fold BatchNorm(inference, no gamma, eps) into per-channel scale/shift:
 * scale = rsqrt(var+eps), shift = beta - mean*scale   (inception_utils.py:56-66). */
int comic_fold_bn(const float* beta, const float* mean, const float* var, float eps, float* scale,
                  float* shift, int c, void* stream);

/* One op of a CNN forward plan. */
typedef struct comic_cnn_op {
  int32_t kind;      /* 0 conv(implicit GEMM, MFMA)  1 stem conv (cin<=4, fp32 input)
                        2 max-pool  3 avg-pool 3x3 s1 SAME (count excludes padding)
                        4 global avg-pool KHxKW VALID -> fp32
                        5 fork: the branch lanes 1..3 start after everything issued so far
                        6 join: the main lane waits for every branch lane
                        7 avg-pool 3x3 s1 SAME of an fp32 source (count excludes padding), then the
                          folded BatchNorm of weight record `weight` and ReLU: the second half of an
                          Inception pool branch whose 1x1 projection was applied BEFORE the pool
                          (both are linear and act on different axes, so they commute; the pool then
                          runs on Cout instead of Cin channels)
                        8 (bf16 plans) streaming stem: conv 3x3 VALID 32 -> 32 (weight record `weight`), conv 3x3
                          SAME 32 -> 64 (record `weight` + 1), each + BatchNorm + ReLU, then max-pool 3x3 / 2 VALID,
                          as ONE line-buffered pass (csrc/conv_stem.hip; inception_v3.py:104-111): H, W = the source
                          map, Cin 32, Cout 64, KH = KW = 3, Ho, Wo = the pooled grid; the two intermediate maps
                          are never materialised
                        9 (bf16 plans) kind 8 with Conv2d_1a_3x3 (3x3 / 2 VALID, 3 -> 32, inception_v3.py:100-104) in
                          front, in the same pass: src = the fp32 image (H, W = the image, W % 4 == 0, Cin 3), weight
                          records `weight` (Conv2d_1a, stem layout), + 1, + 2; the 32-channel map between the stem
                          conv and Conv2d_2a is never materialised either (bit-identical to kinds 1 + 8) */
  int32_t src, dst;  /* indices into the buffer table */
  int32_t src_coff, dst_coff; /* channel offsets inside src/dst (concat without a copy) */
  int32_t H, W, Cin, Cout, KH, KW, SH, SW, PT, PL, Ho, Wo;
  int32_t weight;    /* index into the weight table (conv ops) */
  int32_t relu;
  int32_t out_f32;   /* store fp32 even when the plan dtype is bf16 */
  int32_t src_f32;   /* the source buffer holds fp32 although the plan dtype is bf16
                        (global avg-pool over the fp32 attention feature map) */
  int32_t lane;      /* 0 = caller's stream; 1..3 = internal branch streams (independent
                        Inception branches run concurrently between a fork and a join) */
  int32_t tile;      /* conv: 0 = built-in heuristic, 1..COMIC_CONV_TILES = explicit kernel variant (chosen by
                        the host-side autotuner): 1..12 im2col LDS-DMA tiles / pipeline depths; 13..25
                        patch-resident variants (stride-1 layers whose input window fits the LDS: the window
                        is loaded once per tile, only the weight k-tiles stream; 4, 8 or 12 waves per
                        workgroup); 26..47 wide two-stage im2col tiles (128x128 .. 256x256, 4 or 8 waves: less LDS
                        fill per MFMA).  Ids 48..53: patch-resident variants with
                        loader waves.  Id 54 (COMIC_WS_TILE): weight-stationary kernel for 1x1 convs / groups of 1x1 convs
                        over one source with Cin <= 288 and <= 256 output channels in total (all weights in registers,
                        persistent workgroups, activation tiles streamed once).  An ineligible layer returns an error
                        for ids 13..25 and 48..61.  Ids 56..58 (59..61: the same with the walk of a pixel tile shared by two
                        workgroups that start together on one XCD): "walk" forms of 44 / 38 / 35 for grouped launches whose
                        members share their im2col matrix (the 1x1 convs at the head of an Inception block): one workgroup per
                        pixel tile walks over the out-channel tiles of all members -- the loader waves' k-tile stream runs on
                        across the tiles (no pipeline fill after the first) and the pixel rows are re-read from the L2 the
                        workgroup has just filled.  Id 55 (COMIC_IMG_TILE): image-resident kernel for the stride-1 SAME
                        convs of Mixed_5 / Mixed_6 / Mixed_7 (whole 25x25 / 12x12 / 5x5 images in the LDS without halo,
                        weights streamed global -> VGPR in fragment order: needs comic_conv_weight::w_frag).  In a group
                        the id of the first member applies to all members.  Every variant gives identical bits. */
  int32_t group;     /* conv, bf16 plans: 0 = own launch; ops that are ADJACENT in the table and
                        share a non-zero id are mutually independent (the same-depth convs of
                        the parallel Inception branches) and comic_cnn_forward_grouped runs
                        them as one launch */
  int32_t flags;     /* bit 0 (COMIC_OP_RAW), conv: store the raw product (no BatchNorm, no ReLU);
                        the epilogue is applied by a later kind-7 op */
  int32_t min_lds;   /* conv, bf16 plans: lower bound (bytes, 0 = off) on the dynamic LDS the op's workgroups request --
                        an occupancy knob for forwards that share the GPU with other kernels (84 KiB = one conv
                        workgroup per CU).  A grouped launch takes the value of its first member. */
} comic_cnn_op;
#define COMIC_OP_RAW 1
#define COMIC_OP_POOLED_SRC 2   /* bit 1, 1x1 conv of a bf16 plan: the conv reads its source through a 3x3 / stride-2 VALID
                                   max-pool (H, W = the un-pooled source, Ho, Wo = the pooled grid); the pooled map is never
                                   materialised (slim.max_pool2d + slim.conv2d 1x1, inception_v3.py:111-114,124-199) */

#define COMIC_OP_X3 4           /* bit 2, bf16 plans at fp32-class accuracy ("bf16x3" plans): every bf16 activation buffer
                                   holds a tensor of C channels as THREE channel regions [hi | lo | hi] of its 3C physical
                                   channels -- hi = bf16(v), lo = bf16(v - hi) -- and a conv's packed filter holds
                                   [W_hi | W_hi | W_lo] per tap, so that the ordinary bf16 product over 3C input channels is
                                   hi*W_hi + lo*W_hi + hi*W_lo: the split-bf16 product of the decoder's GEMMs (~2^-16 per
                                   product, fp32 accumulation) on the unchanged conv kernels.  With the bit set a conv /
                                   stem conv stores its bf16 output as the three regions (region stride = a third of the
                                   destination buffer's channels; Cout, dst_coff count channels of ONE region) and a pool
                                   reads hi + lo and writes the three regions (Cin = channels of one region); kind 7 (3x3
                                   average + BN + ReLU of an fp32 map) stores the three regions likewise.  fp32
                                   outputs (out_f32) are stored once, as always. */
#define COMIC_OP_CHAIN_LINK 8    /* bit 3, convs of a COMIC_CHAIN_TILE group (forward-only bf16 plans): this conv's output is the
                                   input of the NEXT op of the table and travels through the LDS of the workgroup that computes
                                   both (csrc/conv_img.hip, conv_img_chain_kernel: the 1x7 / 7x1 convs of a Mixed_6b-e branch,
                                   inception_v3.py:262-345) -- its dst buffer is NOT written.  The group holds one or two chains,
                                   chain after chain; an op without the bit ends its chain and stores to its dst slice.
                                   Every conv: stride 1, SAME, 12x12 maps, Cin in {128, 160, 192}; linked convs keep the
                                   channel count, a chain's last conv has 192 output channels.  Bit-identical to the same ops
                                   run one launch each. */
#define COMIC_OP_CHAIN_KEEP 16   /* bit 4, with COMIC_OP_CHAIN_LINK (trainable plans, whose backward reads every conv's output):
                                   the conv's output is ALSO stored to its dst slice, the same bits the unfused op writes. */

typedef struct comic_conv_weight {
  const void* w;       /* packed [Cout][Kpad], plan dtype (stem conv: fp32 [K][Cout]) */
  const float* scale;  /* [Cout] */
  const float* shift;  /* [Cout] */
  const void* w_frag;  /* bf16 plans, optional (NULL: tile id 55 is not eligible): the same weights in MFMA-fragment
                          order [Cout / 16][Kpad / 32][64 lanes][8] (comic_cnn_pack_frag_weights) */
} comic_conv_weight;

/* Run a whole forward plan on `stream`.  buffers[i] has buf_channels[i] channels per
 * pixel.  Replaces nets_factory.get_network_fn(...)(images) (nets/nets_factory.py:116-159;
 * src/model_base.py:72-77) for the ops listed above. */
int comic_cnn_forward(const comic_cnn_op* ops, int n_ops, void* const* buffers,
                      const int32_t* buf_channels, const comic_conv_weight* weights, int batch,
                      int dtype, void* stream);

/* w_plan: the flat bf16 weight buffer of a plan ([Cout][Kpad] records); w_frag: same size, receives every record in
 * MFMA-fragment order (lane l of k32-step s of 16-channel tile t holds W[16 t + (l & 15)][32 s + 8 (l >> 4) .. + 8]).
 * table_dev: n_weights x {element offset, Cout, Kpad} as int64 in device memory, ascending offsets; Kpad 0 = copy the
 * record unchanged up to the next offset (the fp32 stem filter). */
int comic_cnn_pack_frag_weights(const void* w_plan, void* w_frag, const int64_t* table_dev, int n_weights,
                                int64_t total_elems, void* stream);

/* Grouped execution of the same plan (bf16 plans): every run of ops with the same non-zero
 * `group` becomes ONE launch whose workgroups are spread over all member convolutions (a
 * 12x12 or 5x5 Inception stage has too few tiles per conv to fill 256 CUs at batch 64).
 * Members are convs (kind 0) and, optionally, kind-7 pool+BN+ReLU ops (elementwise work items).
 * The per-conv argument records are built once on the host and kept in device memory:
 *   n = comic_cnn_group_args_bytes(ops, n_ops)            bytes of records the plan needs
 *   comic_cnn_build_group_args(..., host_out)             fills `n` bytes (validates the ops)
 *   <caller copies host_out to device memory: group_args_dev>
 *   comic_cnn_forward_grouped(..., group_args_dev, stream)
 * Results are bit-identical to comic_cnn_forward (same kernels body, same k order). */
long comic_cnn_group_args_bytes(const comic_cnn_op* ops, int n_ops);
int comic_cnn_build_group_args(const comic_cnn_op* ops, int n_ops, void* const* buffers,
                               const int32_t* buf_channels, const comic_conv_weight* weights,
                               int batch, void* host_out);
int comic_cnn_forward_grouped(const comic_cnn_op* ops, int n_ops, void* const* buffers,
                              const int32_t* buf_channels, const comic_conv_weight* weights,
                              int batch, int dtype, const void* group_args_dev, void* stream);

/* ---- input pipeline on the device ------------------------------------------
 * The decoded uint8 RGB images of a batch -> the network input: float [0,1], TF-1 bilinear resize to
 * `resize` x `resize` (256; tf.image.resize_images with align_corners=False), optional horizontal flip, crop of
 * out_h x out_w at (oy, ox), (x - 0.5) * 2.  Replaces inception_preprocessing_radix.preprocess_image as called by
 * manager_image_caption.py:111-228; bit-identical to the numpy restatement comic_amd.inputs.preprocess_image.
 * `blob`: the images back to back (device memory); `desc`: n records (device memory). */
typedef struct comic_image_desc {
  int64_t offset;        /* byte offset of image i in `blob` (in_h x in_w x 3 uint8, row-major) */
  int32_t in_h, in_w;
  int32_t flip, oy, ox;  /* augmentation: flip of the resized image, crop origin in the resized image */
  float sy, sx;          /* float32(in_h / resize), float32(in_w / resize) */
} comic_image_desc;
int comic_image_preprocess(const uint8_t* blob, const void* desc, int n, float* dst /* [n,out_h,out_w,3] */,
                           int out_h, int out_w, int resize, void* stream);

/* Device half of the split JPEG decoder (host half: include/comic_jpeg.h, libcomic_jpeg.so).  Replaces the pixel stage of
 * tf.image.decode_jpeg / libjpeg as the reference's tf.data map runs it per image on host cores
 * (manager_image_caption.py:163-175): dequantisation + jidctint.c "ISLOW" inverse DCT, jdsample.c fancy upsampling,
 * jdcolor.c YCbCr -> RGB, in libjpeg's integer arithmetic -- the RGB bytes are the ones PIL gives for the same file
 * (oracle/jpeg_ref.py, tests/test_jpeg_split.py).
 * `coef`: the batch's quantised coefficients (device, int16, 16-byte aligned); `infos`: n comic_jpeg_info records (device)
 * with coef_base (multiple of 8 elements; comic_jpeg_pool lays the images out back to back) and pixel_off filled, ncomp == 0 for an image to skip; `max_blocks`, `max_w`,
 * `max_h`: the largest coef_count / 64, width and height among them; `planes`: scratch of as many BYTES as `coef` has
 * elements (the component planes); `pixels`: image i as height x width x 3 uint8 at pixel_off -- the blob
 * comic_image_preprocess reads. */
int comic_jpeg_pixels(const int16_t* coef, const void* infos, int n, int max_blocks, int max_w, int max_h, uint8_t* planes,
                      uint8_t* pixels, void* stream);
/* The loader's form: inverse DCT into the component planes, then comic_image_preprocess with its four taps per output
 * pixel converted from the planes on the fly (same integer upsampling / colour arithmetic, same float32 roundings: the
 * values of comic_jpeg_pixels + comic_image_preprocess, bit for bit) -- the RGB image is never written.  `desc`: n
 * comic_image_desc records (in_h / in_w / flip / crop / scales; `offset` is used for images with ncomp == 0 only, which are
 * read as RGB bytes from `blob`: the files the loader's PIL path decoded; `blob` may be NULL when there are none);
 * max_blocks 0: no image needs the inverse DCT. */
int comic_jpeg_preprocess(const int16_t* coef, const void* infos, int n, int max_blocks, uint8_t* planes, const uint8_t* blob,
                          const void* desc, float* dst /* [n,out_h,out_w,3] */, int out_h, int out_w, int resize, void* stream);
/* The same from the PACKED form of a batch (comic_jpeg.h, comic_jpeg_pool_submit_packed: per block a descriptor and the DC value,
 * per non-zero AC coefficient a 16-bit entry -- 3-4x fewer bytes across the bus than the dense blocks): a thread expands its block
 * in LDS in front of the inverse DCT.  `packed`: the batch's blob (device, 4-byte aligned, 16-bit units); image i at
 * infos[i].pixel_off, its component planes at infos[i].coef_base of `planes` (as many bytes as the images' coef_count sum).
 * Same results as comic_jpeg_preprocess on the dense coefficients, bit for bit. */
int comic_jpeg_preprocess_packed(const uint16_t* packed, const void* infos, int n, int max_blocks, uint8_t* planes,
                                 const uint8_t* blob, const void* desc, float* dst /* [n,out_h,out_w,3] */, int out_h, int out_w,
                                 int resize, void* stream);

/* ---- cnn_finetune: backward of the plan (train.py:241-249; model_base.py:76,834-849) ------
 * The CNN variables (conv weights, BN beta) become trainable; BN stays in inference mode, so a
 * conv contributes d beta = sum(1[y>0] dy), d w (backward-weight) and d x (backward-data).
 * grad_buffers[i] mirrors buffers[i] (same shape and element type; NULL where no gradient is
 * wanted, e.g. the image).  They are ACCUMULATED into: zero them, seed the gradients of the
 * plan outputs (feature map, pooled vector), then call.  Per conv weight record `grads[w]`:
 *   w_master  fp32 packed [Cout][Kpad] (stem: [K][Cout]) -- the trainable copy
 *   dw        fp32, same layout, accumulated (atomics: summation order over pixel slices is
 *             not fixed; zero it per step)
 *   dbeta     fp32 [Cout], accumulated (atomics)
 *   w_bwd     plan-dtype scratch of Cin * roundup64(KH*KW*Cout) elements for the flipped /
 *             transposed filter of the backward-data pass (NULL for the stem conv)
 * comic_cnn_refresh_weights re-derives the plan-dtype weight copy (flat bf16 conversion of all
 * masters) and shift = beta - mean*scale after an optimiser step.
 * COMIC_OP_X3 plans ("bf16x3": the same training step at fp32-class accuracy on the bf16 matrix cores): activation buffers
 * hold the three bf16 regions, grad_buffers[i] is an fp32 buffer of the LOGICAL channels (a third of buffers[i]'s; fp32
 * activation buffers as they are), masters / dw / dbeta keep the logical layout of the bf16 plan, and w_bwd holds
 * (Cin / 3) * roundup64(KH*KW*3*Cout) bf16 -- [Wt_hi | Wt_hi | Wt_lo] per tap.  Per conv: d conv as [hi | lo | hi] regions,
 * dw += x_hi dz_hi + x_lo dz_hi + x_hi dz_lo (one launch of the bf16 backward-weight kernel, three work items per tile), backward-data as the bf16
 * conv of d conv over 3 Cout channels accumulated in fp32; pools compare / sum hi + lo.  Both passes below take them (the
 * scheduled one without the fused activation gradients); the forward's [W_hi | W_hi | W_lo] copy follows the masters with
 * comic_cnn_pack_x3_weights. */
typedef struct comic_conv_grad {
  const float* w_master;
  float* dw;
  float* dbeta;
  void* w_bwd;
  int32_t bwd_tile;   /* kernel variant of the backward-data convolution (the ids of comic_cnn_op::tile; 0 = heuristic) */
  int32_t reserved;
} comic_conv_grad;
/* Two lanes: with wgrad_stream != NULL (and != stream) and a scratch of comic_cnn_backward_scratch_bytes(..., lanes = 2)
 * bytes -- every conv's d-conv tensor side by side instead of the largest one -- the weight-gradient launch of a conv goes
 * to wgrad_stream behind its act_grad launch and runs beside the backward-data chain of the earlier layers on `stream`
 * (at batch 32 neither chain fills the chip: 7.2 -> 5.x ms per cnn_finetune step).  The call joins the lanes before it
 * returns: work issued on `stream` afterwards (all-reduce, optimiser) sees every gradient.  lanes = 1 / wgrad_stream = NULL:
 * one chain on `stream`. */
int64_t comic_cnn_backward_scratch_bytes(const comic_cnn_op* ops, int n_ops, int batch, int dtype, int lanes);
int comic_cnn_backward(const comic_cnn_op* ops, int n_ops, void* const* buffers,
                       void* const* grad_buffers, const int32_t* buf_channels,
                       const comic_conv_weight* weights, const comic_conv_grad* grads, int batch,
                       int dtype, int filters_ready /* 1: w_bwd already packed, see below */,
                       void* scratch, int64_t scratch_bytes, void* stream, void* wgrad_stream);
/* The same pass with the parallel branches of the Inception blocks on TWO chain lanes (cnn_finetune at batch 32 is a chain of
 * ~300 dependent small launches).  sched: n_sched rows of four int32 (action, index, lane, alt) in issue order --
 *   0 RUN       backward of ops[index] on chain lane `lane` (0: stream0, 1: stream1); alt = 1: its input gradient
 *               accumulates into grad_buffers_alt[ops[index].src] instead of grad_buffers[...] (lane 1's contributions to a
 *               block's shared input)
 *   1 FORK      stream1 waits for everything issued on stream0 so far
 *   2 JOIN_ADD  stream0 waits for stream1; index >= 0: grad_buffers[index] += grad_buffers_alt[index], the alternate buffer is
 *               cleared (alternate buffers are zero between calls)
 * Every op (kinds 5 / 6 excepted) appears exactly once, consumers before producers inside a lane; the scratch is the
 * lanes = 2 size; weight gradients go to wgrad_stream as in comic_cnn_backward.  All lanes are joined into stream0.
 * The pass fuses the activation gradient of a conv whose output has exactly one reader (a conv on the same lane: the inner
 * convs of the Inception branches) into the epilogue of that reader's backward-data launch -- one launch less per conv on the
 * serial chain of a block's longest branch.  filters_ready carries two bits here: bit 0 as in comic_cnn_backward, bit 1
 * (COMIC_CNN_BWD_NO_ACT_FUSION) runs THIS call with the unfused chain (A/B timing, parity tests). */
#define COMIC_CNN_BWD_NO_ACT_FUSION 2
int comic_cnn_backward_sched(const comic_cnn_op* ops, int n_ops, const int32_t* sched, int n_sched, void* const* buffers,
                             void* const* grad_buffers, void* const* grad_buffers_alt, const int32_t* buf_channels,
                             const comic_conv_weight* weights, const comic_conv_grad* grads, int batch, int dtype,
                             int filters_ready, void* scratch, int64_t scratch_bytes, void* stream0, void* stream1,
                             void* wgrad_stream);
/* Packs the backward-data filters of every conv from the masters (what comic_cnn_backward does per
 * conv when filters_ready == 0).  They only change with the optimiser step, so the caller can do
 * this once per step off the critical path (e.g. on a second stream during the next forward). */
/* COMIC_OP_X3 plans: the forward filter copy of n convs from their fp32 masters in one launch -- masters[i] [cout[i]][roundup64(taps[i]
 * cin[i])] (k = tap cin + ci) -> outs[i] bf16 [cout[i]][roundup64(3 taps[i] cin[i])], per tap [bf16(w) | bf16(w) | bf16(w - bf16(w))]
 * (cin = channels of ONE activation region; padding columns zero).  After an optimiser step of cnn_finetune on a bf16x3 plan. */
int comic_cnn_pack_x3_weights(const float* const* masters, void* const* outs, const int32_t* cout, const int32_t* taps,
                              const int32_t* cin, int n, void* stream);
int comic_cnn_pack_bwd_filters(const comic_cnn_op* ops, int n_ops, const comic_conv_grad* grads,
                               int dtype, void* stream);
int comic_cnn_refresh_weights(const float* master, void* plan_copy, int64_t n, const float* beta,
                              const float* mean, const float* scale, float* shift,
                              int64_t channels, void* stream);

/* Single conv + folded BN + ReLU (slim.conv2d under inception_arg_scope). */
int comic_conv2d_bn_relu(const comic_cnn_op* op, const void* x, int x_channels, void* y,
                         int y_channels, const comic_conv_weight* wt, int batch, int dtype,
                         void* stream);

/* ------------------------------------------------------------------------- */
/* Dense algebra for the decoder (tf MatMul call-sites: common/ops.py:200-238,   */
/* ops_rnn.py:440-447,545; model_base.py:541-543,618-621)                       */
/* ------------------------------------------------------------------------- */
/* C[M,N] = alpha * op(A) * op(B) + beta * C + bias[N]      (fp32, exact-f32 MFMA)
 *   trans_a = 0: A is [M,K] (lda);  1: A is [K,M] (lda)
 *   trans_b = 0: B is [K,N] (ldb);  1: B is [N,K] (ldb)
 * bias may be NULL. */
int comic_gemm_f32(const float* A, const float* B, float* C, const float* bias, int M, int N, int K,
                   int lda, int ldb, int ldc, int trans_a, int trans_b, float alpha, float beta,
                   void* stream);

/* Same contract on the bf16 matrix cores: each fp32 operand element is split into hi + lo bf16 and a
 * product is accumulated as hi*hi + hi*lo + lo*hi in fp32 (relative error of a product <= ~2^-15,
 * ~5x the MFMA rate of the exact path).  With a workspace, shapes with few output tiles split K over
 * workgroups (partial slabs, fixed-order combine).  The decoder executors use it for the time-batched
 * products (keys, logits, weight gradients). */
int comic_gemm_f32_split3(const float* A, const float* B, float* C, const float* bias, int M, int N,
                          int K, int lda, int ldb, int ldc, int trans_a, int trans_b, float alpha,
                          float beta, void* workspace /* may be NULL: no split-K */,
                          int64_t workspace_bytes, void* stream);

/* Several independent products of the comic_gemm_f32_split3 kind in ONE launch (csrc/gemm_group.hip): the training
 * step's products outside the time loops -- the memory / rnn-init projections (common/ops_rnn.py:440-447,
 * src/model_base.py:651-689) and, after the backward loop, every weight gradient, bias sum and attention-parameter sum of
 * tf.gradients over the dense layers (src/model_base.py:325-405; common/ops.py:200-238).  A problem is
 *   C[M][N] = (alpha * op(A) op(B) + bias) [/ keep * mask] + beta * C
 * with type 0: A [K][M], B [K][N] (a weight gradient X^T dY);  1: A [M][K], B [K][N];  2: A [M][K], B [N][K];
 * ones_a: A is all ones and M = 1 (column sums of B: a bias gradient as a product; type 0).  Split-K partials are
 * combined inside the launch in slice order (bit-reproducible).  At most 20 problems; outputs must not alias inputs of
 * other problems of the call. */
typedef struct comic_gemm_prob {
  const float* A;
  const float* B;
  float* C;
  const float* bias;   /* [N] or NULL */
  const float* mask;   /* [M][ld_mask] dropout keep mask or NULL */
  int32_t M, N, K, lda, ldb, ldc, ld_mask;
  float alpha, beta, keep;
  int32_t type, ones_a;
} comic_gemm_prob;
int64_t comic_gemm_group_workspace(const comic_gemm_prob* probs, int n);
int comic_gemm_group(const comic_gemm_prob* probs, int n, void* workspace, int64_t workspace_bytes, void* stream);

/* Skinny product for the decode steps: out[R][N] = x[R][Kin] W[Kin][N] + bias for 33 ... 256 rows (batch x beam), Kin a
 * multiple of 8, any N -- the [TF-1.9] dense layers inside rnn_decoder_beam_search's step (BasicLSTMCell's gate product
 * model_base.py:618-621, the query layer ops_rnn.py:440-447, the output projection model_base.py:531-594) when the
 * weight matrix dwarfs the rows: W is packed to bf16 hi / lo halves in MFMA-fragment order and read from HBM exactly
 * once for all rows through an LDS-DMA stream (csrc/lstm_stream.hip), products hi*hi + hi*lo + lo*hi as
 * comic_gemm_f32_split3.  This entry packs, streams and sums in one call (the decode executors keep the packed weights
 * across the steps of a decode).  bias may be NULL. */
int64_t comic_gemm_f32_stream_workspace(int R, int Kin, int N);
int comic_gemm_f32_stream(const float* x, const float* W, const float* bias, float* out, int R, int Kin, int N,
                          void* workspace, int64_t workspace_bytes, void* stream);

/* Same product; a caller-provided workspace lets skinny problems (M <= 2048, no trans_a)
 * split K over extra workgroups (deterministic slab reduction). */
int comic_gemm_f32_splitk(const float* A, const float* B, float* C, const float* bias, int M, int N,
                          int K, int lda, int ldb, int ldc, int trans_a, int trans_b, float alpha,
                          float beta, void* workspace, int64_t workspace_bytes, void* stream);

/* ------------------------------------------------------------------------- */
/* Decoder step kernels                                                       */
/* ------------------------------------------------------------------------- */
/* out[r,:] = ids[r] >= 0 ? table[ids[r],:] : 0        (model_base.py:523-526,587-593) */
int comic_embed_fwd(const float* table, const int32_t* ids, float* out, int rows, int E, int V,
                    void* stream);
/* dtable[v,:] += sum_{r: ids[r]==v} dout[r,:]   (deterministic, no atomics) */
int comic_embed_bwd(const int32_t* ids, const float* dout, float* dtable, int rows, int E, int V,
                    void* stream);

/* y = (x / keep) * mask  (tf.nn.dropout, TF-1.9 form); mask NULL -> copy */
int comic_dropout_apply(const float* x, const float* mask, float keep, float* y, int64_t n,
                        void* stream);
/* Bernoulli(keep) 0/1 masks from a counter-based generator (stateless, seed+offset). */
int comic_dropout_mask(float* mask, int64_t n, float keep, uint64_t seed, uint64_t offset,
                       void* stream);
/* same generator, seed read from device memory at run time (hipGraph replays draw new masks) */
int comic_dropout_mask_dev(float* mask, int64_t n, float keep, const uint64_t* seed_dev,
                           uint64_t offset, void* stream);
/* The four consecutive masks of a training step (n4[i] elements with keep probability keep4[i], one buffer) in one
 * launch: the same bits as four comic_dropout_mask_dev calls with cumulative offsets. */
int comic_dropout_masks4_dev(float* mask, const int64_t* n4, const float* keep4, const uint64_t* seed_dev,
                             void* stream);
/* out[0] = sum_{t,b} rows_tb[t*B + b] * w_bt[b*T + t]: the tf.contrib.seq2seq.sequence_loss reduction
 * (model_base.py:337-347) over the per-row cross-entropies of comic_decoder_train_step. */
int comic_weighted_sum_tb(const float* rows_tb, const float* w_bt, int T, int B, float* out, void* stream);

/* BasicLSTMCell gate math (model_base.py:618-621; gate order i,j,f,o; forget_bias 1).
 *  g [B,4D] pre-activations (bias included).  Writes activated gates [B,4D] (for bwd),
 *  c_new/h_new [B,D], y = dropout(h_new) [B,D].  A row is FINISHED when lens != NULL and
 *  t >= lens[b] (TrainingHelper rule); finished rows keep their state (impute_finished,
 *  ops_rnn.py:222-228): c_state = fin ? c_prev : c_new, h_state likewise.  c_prev/h_prev
 *  NULL mean zero state.  Any output pointer may be NULL. */
int comic_lstm_gates_fwd(const float* g, const float* c_prev, const float* h_prev, float* gates_act,
                         float* c_new, float* h_new, float* y, const float* mask_out, float keep_out,
                         const int32_t* lens, int t, float* c_state, float* h_state, int B, int D,
                         void* stream);
/* Backward of the above.  dc_state/dh_state are the gradients w.r.t. the carried state
 * (in/out: on return they hold the pass-through part for finished rows plus dc_prev; the
 * caller adds dh_prev from dg * K^T); dy is the gradient w.r.t. y (may be NULL).
 * dg [B,4D] is the gradient w.r.t. the pre-activations. */
int comic_lstm_gates_bwd(const float* gates_act, const float* c_prev, const float* c_new,
                         const float* dy, const float* mask_out, float keep_out, const int32_t* lens,
                         int t, float* dc_state, float* dh_state, float* dg, int B, int D, void* stream);

/* Attention descriptor shared by fwd/bwd. */
typedef struct comic_attn_desc {
  int32_t B, M, D, H, Cv;  /* keys [B,M,D]; values [B,M,Cv]; H heads */
  int32_t method;          /* 0 add_LN (ops_rnn.py:531-565)   1 dot (ops_rnn.py:611-632) */
  int32_t prob;            /* 0 softmax   1 sigmoid-normalised (model_base.py:599-603) */
  int32_t tied;            /* values alias keys (cnn_fm_projection == 'tied') */
} comic_attn_desc;

/* Fused per-step attention: score (LN-tanh-v | dot) -> per-head softmax over M ->
 * dropout(alpha) -> context.  MultiHeadAddLN.__call__ + MultiHeadAttentionWrapperV3.call
 * (ops_rnn.py:543-563, :692-716).  alpha [B,H,M] (pre-dropout, kept for backward),
 * alpha_d [B,H,M] (post-dropout = alignment history entry), ctx [B,Cv]. */
int comic_attn_step_fwd(const comic_attn_desc* d, const float* keys, const float* values,
                        const float* q, const float* ln_g, const float* ln_b, const float* v,
                        const float* tau, const float* mask_alpha, float keep_alpha, float* alpha,
                        float* alpha_d, float* ctx, void* stream);
/* Backward: given dctx [B,Cv] and the map-loss term dmap [B,M] (added to every head's
 * d alpha_d; may be NULL) produces dq [B,D]; ACCUMULATES into dkeys [B,M,D] and dvalues
 * [B,M,Cv] (same buffer when tied) and into the per-row parameter partials
 * pgrad [B, 3*D+1] = (d v | d ln_g | d ln_b | d tau). */
int comic_attn_step_bwd(const comic_attn_desc* d, const float* keys, const float* values,
                        const float* q, const float* ln_g, const float* ln_b, const float* v,
                        const float* tau, const float* alpha, const float* mask_alpha,
                        float keep_alpha, const float* dctx, const float* dmap, float* dq,
                        float* dkeys, float* dvalues, float* pgrad, void* stream);

/* sequence_loss forward+backward over time-major logits [T,B,V] (model_base.py:337-347):
 * rows with t >= lens[b] are zeroed (impute_finished), loss_rows[t*B+b] = xent*w,
 * dlogits = (softmax - onehot) * coef[b*T+t]; ids = argmax (lowest index wins). */
int comic_xent_fwd_bwd(float* logits, const int32_t* targets_bt, const float* coef_bt,
                       const float* wmask_bt, const int32_t* lens, float* loss_rows, float* dlogits,
                       int32_t* ids_tb, int T, int B, int V, void* stream);

/* ------------------------------------------------------------------------- */
/* Decoding (ops_rnn.py:49-180; tf.contrib.seq2seq BeamSearchDecoder, gather_tree) */
/* ------------------------------------------------------------------------- */
int comic_argmax_rows(const float* x, int32_t* idx, int rows, int V, void* stream);
/* One _beam_search_step: logits [B,W,V]; in/out log_probs [B,W], finished [B,W] (int32),
 * lengths [B,W] (int64); out word/parent/scores [B,W].  Ties: lower flat index first. */
int comic_beam_step(const float* logits, float* log_probs, int32_t* finished, int64_t* lengths,
                    int32_t* word_ids, int32_t* parent_ids, float* scores, int B, int W, int V,
                    int end_id, void* stream);

/* The same step from the decoder outputs y [B*W][D]: logits = y W_o + b_o (W_o [D][V], TensorFlow layout), then as
 * comic_beam_step -- with the kernels comic_decoder_beam picks for the shape: from V = 4096 (D % 128 == 0, <= 256 rows,
 * beam <= 8) the projection, log-softmax and top-k are one streaming launch over packed hi/lo W_o fragments plus a merge and
 * the logits are never written (csrc/beam_logits.hip); up to V = 1024 a beam's logits stay in a wave's registers;
 * otherwise GEMM + comic_beam_step.  Total order (score descending, flat index ascending), _mask_probs, lengths and
 * finished flags as there. */
int64_t comic_beam_step_dense_workspace(int B, int W, int D, int V);
int comic_beam_step_dense(const float* y, const float* W_o, const float* b_o, float* log_probs, int32_t* finished,
                          int64_t* lengths, int32_t* word_ids, int32_t* parent_ids, float* scores, int B, int W, int D,
                          int V, int end_id, void* workspace, int64_t workspace_bytes, void* stream);
/* out[r,:] = in[(r/W)*W + parent[r], :]   (state re-ordering by parent beam) */
int comic_gather_rows(const float* in, const int32_t* parent, float* out, int rows, int W, int cols,
                      void* stream);
/* beam_search_ops.gather_tree: step_ids/parent_ids/out [T,B,W]; max_len [B]. */
int comic_gather_tree(const int32_t* step_ids, const int32_t* parent_ids, const int32_t* max_len,
                      int32_t* out, int T, int B, int W, int end_id, void* stream);

/* ------------------------------------------------------------------------- */
/* Optimiser (model_base.py:852-883; tf.train.AdamOptimizer ApplyAdam, TF-1.9)  */
/* ------------------------------------------------------------------------- */
/* g_eff = g*gscale + l2*w; m += (g_eff-m)(1-b1); v += (g_eff^2-v)(1-b2);
 * w -= lr_t*m/(sqrt(v)+eps), lr_t = lr*sqrt(1-b2^t)/(1-b1^t) computed by the caller. */
int comic_adam_tf(float* w, const float* g, float* m, float* v, int64_t n, float lr_t, float beta1,
                  float beta2, float eps, float l2, float gscale, void* stream);
/* Legacy encoder head (`--legacy`, model_base.py:80-91): ops.layer_norm_activate('LN_tanh', squeeze(net), tanh)
 * (common/ops.py:241-275: tf.contrib.layers.layer_norm over the last axis, eps 1e-12, center + scale) followed by
 * ops.linear('im_embed', 1024, no bias) = comic_gemm_f32.  y = tanh(xhat*gamma + beta), xhat = (x-mean)*rsqrt(var+eps)
 * (kept for the backward).  comic_ln_tanh_bwd_rows writes the per-row terms of d gamma / d beta
 * (dy*(1-y^2)*xhat and dy*(1-y^2)); their column sums (comic_colsum) are the parameter gradients.  The head is never
 * back-propagated into the CNN: train.py:241-249 refuses cnn_finetune / scst with --legacy. */
int comic_ln_tanh_fwd(const float* x, const float* gamma, const float* beta, float* y, float* xhat, int B, int C,
                      float eps, void* stream);
int comic_ln_tanh_bwd_rows(const float* dy, const float* y, const float* xhat, float* pgamma, float* pbeta, int B,
                           int C, void* stream);
/* tf.train.MomentumOptimizer(momentum 0.9, use_nesterov=False) (model_base.py:867-880; ApplyMomentum, TF-1.9):
 * g_eff = g*gscale + l2*w; accum = momentum*accum + g_eff; w -= lr*accum. */
int comic_momentum_tf(float* w, const float* g, float* accum, int64_t n, float lr, float momentum, float l2,
                      float gscale, void* stream);
/* The same updates, skipped ON THE DEVICE (w, m, v / accum untouched) when skip_flag[0] != 0: the optimiser step behind a
 * training step that comic_decoder_train_step voided (comic_decoder_params::status of the gradient table).  NULL = never
 * skip. */
int comic_adam_tf_gated(float* w, const float* g, float* m, float* v, int64_t n, float lr_t, float beta1, float beta2,
                        float epsilon, float l2, float gscale, const float* skip_flag, void* stream);
int comic_momentum_tf_gated(float* w, const float* g, float* accum, int64_t n, float lr, float momentum, float l2,
                            float gscale, const float* skip_flag, void* stream);
/* Gradient clipping of `--clip_gradient_norm` (train.py:137 -> model_base.py:394-401: slim.learning.create_train_op(
 * clip_gradient_norm=c) = clip_gradient_norms [TF-1.9 slim]: tf.clip_by_norm on EVERY variable's gradient by that tensor's
 * own L2 norm, after the gradient multipliers).  The clipped quantity is g_eff = g*gscale + l2*w (the L2 term is part of
 * the reference's loss; gscale carries the data-parallel 1/W and the CNN multiplier): g_eff * clip / max(||g_eff||, clip).
 * In place on g, such that the optimiser's own g*gscale + l2*w yields the clipped g_eff.  chunks: n_chunks records of five
 * int64 on the device -- (variable, first element, elements, first chunk of the variable, chunks of the variable), a
 * variable cut into pieces of <= 8192 elements; partial: n_chunks floats of scratch.  Sums run in a fixed order.  skip_flag
 * as in comic_adam_tf_gated. */
int comic_clip_by_norm(float* g, const float* w, const int64_t* chunks, int n_chunks, float l2, float gscale,
                       float clip_norm, float* partial, const float* skip_flag, void* stream);
/* Test aid: n_workgroups workgroups (256 threads) that stay resident for about `microseconds` each (bounded spin on the
 * 100 MHz clock) -- what a collective's kernels do to some CUs while a training step runs beside them. */
int comic_debug_occupy_cus(int n_workgroups, int microseconds, void* stream);
/* out[j] = sum_i in[i*cols + j]  (deterministic column sums; parameter-partial reduce) */
int comic_colsum(const float* in, float* out, int rows, int cols, float beta, void* stream);
int comic_axpy(float* y, const float* x, float a, int64_t n, void* stream);

/* ------------------------------------------------------------------------- */
/* Native decoder executors                                                   */
/* ------------------------------------------------------------------------- */
typedef struct comic_decoder_desc {
  int32_t D, E, A, V, C, Cg, H, M;     /* rnn, word, attention, softmax, fm ch, im_embed, heads, map */
  int32_t Cv;                          /* value channels (D, or C when projection is none) */
  int32_t fm_projection;               /* 0 none, 1 independent, 2 tied */
  int32_t method, prob;                /* as comic_attn_desc */
  int32_t context_layer;               /* attn_context_layer */
  int32_t init_method;                 /* 0 first_input, 1 project_hidden */
  int32_t start_id, end_id;
  float keep_in, keep_out, keep_alpha; /* 1.0 disables the dropout */
  float map_loss_scale;
  uint32_t flags;                      /* COMIC_DEC_* executor switches, 0 = every fast path on (A/B measurements and tests;
                                          results agree to fp32 summation order).  The library reads no environment. */
  float length_penalty_weight;         /* comic_decoder_beam only: BeamSearchDecoder(length_penalty_weight) -- candidates
                                          are ranked by total / ((5 + length) / 6)^w (ops_rnn.py:96, infer.py:65); 0 = none */
  int32_t cell;                        /* COMIC_CELL_*: --rnn_name (src/model_base.py:606-632) */
} comic_decoder_desc;
#define COMIC_CELL_LSTM 0               /* tf.contrib.rnn.BasicLSTMCell: K [E+A+D][4D], b [4D] */
#define COMIC_CELL_LN_LSTM 1            /* tf.contrib.rnn.LayerNormBasicLSTMCell: K [E+A+D][4D], no bias, cell_ln */
#define COMIC_CELL_GRU 2                /* tf.contrib.rnn.GRUCell: K [E+A+D][2D] + b [2D] (gates), K_c [E+A+D][D] + b_c [D] (candidate) */
#define COMIC_DEC_NO_PERSIST 1u         /* time loops as per-step launches (forward and backward; greedy too) */
#define COMIC_DEC_NO_PERSIST_BWD 2u     /* backward time loop as per-step launches */
#define COMIC_DEC_NO_FUSED_STEP 4u      /* split-K GEMM + element-wise kernel chain instead of the fused step kernels */
#define COMIC_DEC_NO_SPLIT_ATTN_BWD 8u  /* one attention-backward workgroup per batch row */
#define COMIC_DEC_ONE_LANE 16u          /* no second stream inside the training executor */
#define COMIC_DEC_EXACT_GEMM 32u        /* exact-fp32 MFMA for the time-batched products (no hi/lo-split bf16) */
#define COMIC_DEC_STAMPS 64u            /* diagnostic phase clocks of the persistent loops (host sync per launch) */
#define COMIC_DEC_PHASE_FWD 512u        /* comic_decoder_train_step: only the part that needs no loss coefficient (forward to the logits) */
#define COMIC_DEC_PHASE_BWD 1024u       /* ... only the rest (loss, backward), over the SAME workspace and arguments as the forward call */
#define COMIC_DEC_NO_GROUP_GEMM 2048u    /* comic_decoder_train_step: the products outside the time loops as separate launches on two lanes
                                           instead of grouped launches (comic_gemm_group) */
#define COMIC_DEC_INJECT_TIMEOUT 8192u  /* fault injection (tests): THIS comic_decoder_train_step call, if it runs a persistent loop, reports a loop
                                           timeout (NaN loss_rows[0] / map_loss, zero gradients, status words raised) although its kernels completed */
#define COMIC_DEC_BWD_OWN_ROWS 4096u    /* persistent backward loop in its own-rows form (the form of memories of more than 64 rows) whatever M is */
#define COMIC_DEC_NO_LSTM_STREAM 256u   /* decode steps at > 32 rows with the per-row-tile fused LSTM kernel instead of the streaming one */
#define COMIC_DEC_NO_BEAM_LOGITS 128u   /* beam step as GEMM + statistics + chunk top-k + merge (large V) / comic_beam_step's kernel (small V)
                                           instead of the streaming logits + top-k launch / the register-resident small step */

/* Parameter (or gradient) table; every pointer is a view into one flat fp32 buffer. */
typedef struct comic_decoder_params {
  float *W_init, *K, *b, *W_m, *W_v, *W_q, *v, *ln_g, *ln_b, *tau, *W_a, *W_o, *b_o, *emb;
  float* cell_ln;   /* LN_LSTM: ten [D] vectors gamma, beta of the scopes input, transform, forget, output, state, in that
                       order, (D + 63) / 64 * 64 floats apart; NULL otherwise */
  float *K_c, *b_c; /* GRU: candidate kernel and bias; NULL otherwise */
  float* status;    /* optional (may be NULL), one float behind the flat buffer.  In the GRADIENT table: comic_decoder_train_step
                       writes 1 when it voided the step (a bounded wait of a persistent loop expired), else 0 -- the word the
                       gated optimiser entries read; it lies inside the all-reduced buffer, so under data parallelism every
                       rank sees a non-zero sum and skips the same update.  In the PARAMETER table: a sticky count of voided
                       steps (incremented, never cleared by the library) for the host to read at its log points. */
} comic_decoder_params;

/* Workspace size in bytes for a training step at (B, T, M) / a decode at rows=B*W. */
int64_t comic_decoder_train_workspace(const comic_decoder_desc* d, int B, int T);
int64_t comic_decoder_infer_workspace(const comic_decoder_desc* d, int rows, int max_steps);

/* One teacher-forced forward + backward (CaptionModel train graph: src/model.py:44-59;
 * rnn_decoder_training ops_rnn.py:183-243; losses model_base.py:325-405).
 *   fm [B,M,C], im_embed [B,Cg] fp32; inputs_bt/targets_bt [B,T] int32; wmask/coef [B,T];
 *   lens [B] int32 (device) and Tp = max(lens) (host);
 *   masks (may be NULL when the keeps are 1): init_in [B,E+A], in [Tp,B,E+A],
 *   out [Tp,B,D], alpha [Tp,B,H,M].
 * Outputs: logits [T,B,V] (time-major), ids [T,B], attn history [Tp,B,H,M], loss_rows [T*B],
 *   map_loss (1 float), grads (no L2), dfm [B,M,C] / dim_embed [B,Cg] (may be NULL).
 * The forward and the backward time loop each run as ONE persistent launch per 64 batch rows when the shape allows
 * (D = 512, B <= 256, no context layer, M <= 64 or tied keys/values and M <= 256; backward: tied keys/values, softmax,
 * M <= 256; a device with >= 256 CUs), as
 * per-step launches otherwise: same results to fp32 summation order (comic_decoder_train_path tells which).  If a
 * bounded wait inside a persistent loop ever expires, the step's results are void and the call says so in its
 * outputs: loss_rows[0] and map_loss[0] are NaN (the sequence loss reduced from loss_rows is then NaN too) and every
 * gradient (grads, dfm, dim_embed) is zero, so an optimiser step issued without a host check applies no gradient. */
int comic_decoder_train_step(const comic_decoder_desc* d, const comic_decoder_params* p,
                             const comic_decoder_params* grads, const float* fm,
                             const float* im_embed, const int32_t* inputs_bt,
                             const int32_t* targets_bt, const float* wmask_bt, const float* coef_bt,
                             const int32_t* lens, int B, int T, int Tp, const float* mask_init_in,
                             const float* mask_in, const float* mask_out, const float* mask_alpha,
                             float* logits_tb, int32_t* ids_tb, float* attn_hist, float* loss_rows,
                             float* map_loss, float* dfm, float* dim_embed, void* workspace,
                             int64_t workspace_bytes, void* stream);

/* Which form of the time loops the LAST comic_decoder_train_step of this thread used: bit 0 = forward loop as one
 * persistent launch (csrc/decoder_persist.hip), bit 1 = backward loop as one persistent launch
 * (csrc/decoder_persist_bwd.hip); 0 = per-step launches.  The choice depends on the shape (D = 512, B <= 64, ...), the
 * device (one workgroup per CU must be resident) and the COMIC_PERSIST / COMIC_PERSIST_BWD switches. */
int comic_decoder_train_path(void);
/* 1 when the LAST comic_decoder_greedy of this thread ran its loop as one persistent launch (D = 512, B <= 64,
 * V <= 512, no context layer, a device with enough CUs, COMIC_PERSIST != 0), 0 for per-step launches. */
int comic_decoder_greedy_path(void);
/* Paths of the LAST comic_decoder_beam of this thread, a bit mask: 1 = projection + top-k as the streaming launch over
 * a packed W_o (csrc/beam_logits.hip: D % 128 == 0, V >= 4096, batch * beam <= 256, beam <= 8, fused step available)
 * instead of the GEMM + statistics + top-k launches; 2 = the LSTM step as one streaming pass over the packed kernel
 * (csrc/lstm_stream.hip: 33 ... 256 rows) instead of the per-row-tile fused kernel; 4 = small vocabulary (V <= 1024, beam
 * <= 8): the entry's beam step with a beam's logits held in a wave's registers (beam_step_small_kernel) instead of
 * comic_beam_step's kernel + the all-finished launch. */
int comic_decoder_beam_path(void);

/* Greedy decode (rnn_decoder_search, ops_rnn.py:115-180): runs `max_steps` steps on the
 * device without host sync; ids [max_steps,B], attn [max_steps,B,H,M]; finished-at step per
 * row in first_eos [B] (max_steps when no EOS).  The host trims to the executed length. */
int comic_decoder_greedy(const comic_decoder_desc* d, const comic_decoder_params* p, const float* fm,
                         const float* im_embed, int B, int max_steps, int32_t* ids_tb,
                         float* logits_tb, float* attn_hist, int32_t* first_eos, void* workspace,
                         int64_t workspace_bytes, void* stream);

/* Sampled decode (rnn_decoder_search(greedy_search=False), ops_rnn.py:158-166: SampleEmbeddingHelper [TF-1.9]: the next
 * token is a draw of Categorical(logits), fed back through the embedding; finished when it is EOS).  The draw is
 * argmax(logits + g) with Gumbel noise g = -log(-log u) supplied by the caller (gumbel_tb [max_steps,B,V]), i.e. the
 * random stream is the caller's; logits_tb stays the undisturbed projection (BasicDecoder's rnn_output).  Per-step
 * launches; outputs as comic_decoder_greedy. */
int comic_decoder_sample(const comic_decoder_desc* d, const comic_decoder_params* p, const float* fm,
                         const float* im_embed, int B, int max_steps, const float* gumbel_tb, int32_t* ids_tb,
                         float* logits_tb, float* attn_hist, int32_t* first_eos, void* workspace,
                         int64_t workspace_bytes, void* stream);

/* Beam search (rnn_decoder_beam_search, ops_rnn.py:49-112).  fm/im_embed are UN-tiled
 * [B,...]; tiling by W happens inside (model_base.py:127-131).  Outputs, all [max_steps,B,W]:
 * step_ids, parent_ids, scores; final lengths [B,W] (int64), finished [B,W];
 * attn_hist [max_steps, B*W, H*M] (un-sorted; sort on host with gather_tree_from_array).
 * The loop always issues max_steps steps (no host sync); steps_executed (device int32)
 * receives the step count after which every beam was finished, i.e. the length the
 * reference's dynamic_decode would have produced; later steps are identity and ignored. */
int comic_decoder_beam(const comic_decoder_desc* d, const comic_decoder_params* p, const float* fm,
                       const float* im_embed, int B, int W, int max_steps, int32_t* step_ids,
                       int32_t* parent_ids, float* scores, int64_t* lengths, int32_t* finished,
                       float* attn_hist, int32_t* steps_executed, void* workspace,
                       int64_t workspace_bytes, void* stream);

/* ------------------------------------------------------------------------- */
/* SCST reward scorer (host, multi-threaded)  common/scst/scorers.py:43-171      */
/* ------------------------------------------------------------------------- */
typedef struct comic_scorer comic_scorer;
/* df entries: `ngrams_host` = n_entries NUL-terminated strings (words joined by ' '),
 * concatenated; counts_host[n_entries]; ref_len = number of images in the df corpus. */
comic_scorer* comic_scorer_create(const char* ngrams_host, const double* counts_host,
                                  int64_t n_entries, double ref_len);
void comic_scorer_destroy(comic_scorer* s);
/* hypos: n strings; refs: for hypothesis i, refs_per[i] strings, concatenated in order.
 * out_cider[n]; out_bleu[n*4] (BLEU-1..4 per sentence, 'closest' reflen). */
int comic_scorer_score(const comic_scorer* s, const char* const* hypos_host, int n,
                       const char* const* refs_host, const int32_t* refs_per_host,
                       double* out_cider_host, double* out_bleu_host, int n_threads);

/* CRC-32C of a host buffer, continuing from `crc` (0 to start): checksum of the TF checkpoint-V2
 * container the reference's tf.train.Saver reads and writes (train_fn.py:67-70,131-132). */
uint32_t comic_crc32c(const void* data_host, size_t n, uint32_t crc);

#ifdef __cplusplus
}
#endif
#endif /* COMIC_HIP_H_ */
