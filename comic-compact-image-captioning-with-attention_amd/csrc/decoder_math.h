// Scalar math shared by the decoder kernels (decoder.hip, decoder_persist.hip): forward and backward must evaluate
// the same tanh / sigmoid forms.
#pragma once
#include "common.h"

namespace {

constexpr float kLnEps = 1e-12f;

__device__ __forceinline__ float sigmoidf_(float x) { return 1.0f / (1.0f + expf(-x)); }

// tanh for the attention score (called B*M*D times per step): odd polynomial below 0.25
// (next term < 2e-9), 1 - 2/(1+e^{2|x|}) on v_exp_f32 above; |error| < 3e-7 absolute.
__device__ __forceinline__ float fast_tanh(float x) {
  const float ax = fabsf(x);
  const float x2 = x * x;
  const float p = x * (1.0f + x2 * (-0.33333334f + x2 * (0.13333334f + x2 * (-0.053968254f + x2 * 0.021869488f))));
  const float e = __expf(2.0f * ax);
  const float t = 1.0f - __fdividef(2.0f, e + 1.0f);
  return ax < 0.25f ? p : copysignf(t, x);
}

}  // namespace
