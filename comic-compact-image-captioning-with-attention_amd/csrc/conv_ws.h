// Internal interface between conv.hip (plan executor) and conv_ws.hip (weight-stationary 1x1 convolution groups).
#pragma once
#include "conv_common.h"

struct ComicWsMember {
  const bf16_t* w;       // [cout][Kpad] bf16
  const float* scale;    // null: raw product (no BatchNorm)
  const float* shift;
  void* y;
  int y_cs, y_co, cout, relu, out_f32;
  int tile0;             // first 16-channel tile of this member in the concatenated N
};
struct ComicWsArgs {
  const bf16_t* x;
  int B, H, W, x_cs, x_co, Cin, Kpad;
  int Ho, Wo, M;         // output pixel grid (= H x W, or the pooled grid)
  int pooled;            // 1: an activation row is the 3x3 / stride-2 VALID max-pool window of x
  int n_members, n_tiles, tiles_m;
  ComicWsMember m[4];
};

bool comic_ws_supported(int Cin, int n_tiles);
int comic_ws_launch(const ComicWsArgs& a, hipStream_t st);
