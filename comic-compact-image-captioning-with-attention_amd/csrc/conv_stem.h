// Internal interface between conv.hip (plan executor) and conv_stem.hip (streaming Conv2d_2a -> Conv2d_2b -> MaxPool_3a).
#pragma once
#include "conv_common.h"

struct ComicStemArgs {
  const bf16_t* x;        // Conv2d_1a output [B][H0][W0][x_cs] bf16, channels [x_co, x_co + 32)
  int B, H0, W0, x_cs, x_co;
  const bf16_t* w1;       // [32][Kpad]  k = (kh*3 + kw)*32 + c
  const bf16_t* w2;       // [64][Kpad]
  int Kpad;
  const float *sc1, *sh1, *sc2, *sh2;
  bf16_t* y;              // pooled output [B][Hp][Wp][y_cs], channels [y_co, y_co + 64)
  int y_cs, y_co, Hp, Wp;
  int n_tasks, parts;     // set by comic_stem_stream_launch: tasks = (image, band of the pooled rows), bands per image
  // op kind 9 (Conv2d_1a_3x3 inside the pass): the fp32 image instead of x; null = kind 8
  const float* img;       // [B][Hi][Wi][3]
  int Hi, Wi;
  const float* w0;        // Conv2d_1a filter, stem layout fp32 [27][32]
  const float *sc0, *sh0;
};

bool comic_stem_stream_supported(int H0, int W0);
bool comic_stem_stream_1a_supported(int Hi, int Wi);
int comic_stem_stream_launch(const ComicStemArgs& a, hipStream_t st);
