// Internal interface between conv.hip (plan executor) and conv_img.hip (image-resident convolutions).
#pragma once
#include "conv_common.h"

constexpr int kImgMaxMembers = 4;

struct ComicImgMember {
  const bf16_t* x;       // source activation buffer (NHWC bf16)
  const bf16_t* wf;      // weights in MFMA-fragment order (comic_cnn_pack_frag_weights)
  const float* scale;    // null: raw product (no BatchNorm)
  const float* shift;
  void* y;
  int x_cs, x_co, y_cs, y_co;
  int KH, KW, PT, PL, relu, out_f32;
};
struct ComicImgArgs {
  ComicImgMember m[kImgMaxMembers];
  int n_members;
  int B, H, W, Cin, Cout;   // common to all members: stride-1 SAME convolutions over H x W maps
  int G, groups;            // images per workgroup, workgroups per member = ceil(B / G)
  int PXBp;                 // bytes per resident pixel (Cin * 2, padded to an odd multiple of 32)
  int KS32;                 // 32-deep k-steps per 16-channel tile in the packed weights (= Kpad / 32)
};

// config id (>= 0) of the kernel instantiation that serves this shape, -1: not eligible
int comic_img_config(int H, int W, int Cin, int Cout, int KH, int KW, int SH, int SW, int Ho, int Wo);
int comic_img_images_per_group(int cfg);
int comic_img_launch(int cfg, const ComicImgArgs& a, hipStream_t st);

// ---- chains of image-resident convs (conv_img_chain_kernel): the 1x7 / 7x1 convs of a Mixed_6 branch as one launch --------
constexpr int kChainMaxConvs = 4, kChainMaxMembers = 2;
struct ComicChainConv {
  const bf16_t* wf;      // fragment-order weights
  const float* scale;
  const float* shift;
  int KH, KW, PT, PL, KS32, Cout, relu, keep_cs;
  void* keep;            // linked convs of a trainable plan (COMIC_OP_CHAIN_KEEP): the conv's destination buffer, written as well
  int keep_co, pad_;
};
struct ComicChainMember {
  const bf16_t* x;       // input of the first conv
  void* y;               // output of the last conv
  int x_cs, x_co, y_cs, y_co, out_f32, n_convs;
  ComicChainConv c[kChainMaxConvs];
};
struct ComicChainArgs {
  ComicChainMember m[kChainMaxMembers];     // member 0: the longer chain
  int n_members;
  int B, H, W, Cin, PXBp;   // Cin = input channels of every conv = output channels of every conv but a chain's last (192)
};
int comic_img_chain_supported(int H, int W, int Cin);
int comic_img_chain_launch(const ComicChainArgs& a, hipStream_t st);
