// Image preprocessing on the device: the decoded uint8 RGB images of a batch -> the network input.
//
// Replaces inception_preprocessing_radix.preprocess_image for the captioning pipeline (reference
// common/inputs/preprocessing/inception_preprocessing_radix.py:158-278 as used by
// manager_image_caption.py:111-228): uint8 -> float [0,1] -> TF-1 bilinear resize to 256x256
// (tf.image.resize_images, align_corners=False: src = dst * in/out, no half-pixel offset) -> optional horizontal
// flip -> 224x224 (or any h x w) crop at (oy, ox) -> (x - 0.5) * 2.  The host keeps JPEG decoding only.  float32
// arithmetic at TF-1.9's rounding points (convert_image_dtype: cast * (1/255); resize_bilinear_op.cc compute_lerp:
// top = tl + (tr - tl) * xl, bottom likewise, out = top + (bottom - top) * yl); contraction into FMAs is
// switched off (#pragma clang fp contract(off)) so every operation rounds on its own.  Parity reference: oracle/preprocess_ref.py (bit-identical).
#include "common.h"

namespace {

struct ImgDesc {
  int64_t offset;     // byte offset of the image in the uint8 blob (H x W x 3, row-major)
  int32_t in_h, in_w;
  int32_t flip, oy, ox;
  float sy, sx;       // float32(in_h / 256), float32(in_w / 256)
};

// a + (b - a) * w, three roundings (resize_bilinear_op.cc compute_lerp [TF-1.9]).  HIP's __f*_rn intrinsics are plain
// operators on AMD and hipcc contracts a + b * c into an FMA by default (-ffp-contract=fast): contraction is switched
// off for these functions, the TF kernel (built without FMA) rounds after the product.
__device__ __forceinline__ float lerp_rn(float a, float b, float w) {
#pragma clang fp contract(off)
  const float d = b - a;
  const float m = d * w;
  return a + m;
}

__global__ __launch_bounds__(256) void image_preprocess_kernel(const uint8_t* __restrict__ blob,
                                                               const ImgDesc* __restrict__ desc, float* __restrict__ dst,
                                                               int out_h, int out_w, int resize) {
#pragma clang fp contract(off)
  const int i = blockIdx.y;
  const int p = blockIdx.x * blockDim.x + threadIdx.x;
  if (p >= out_h * out_w) return;
  const ImgDesc d = desc[i];
  const int y = p / out_w, x = p % out_w;
  const int Y = d.oy + y;
  const int X = d.flip ? resize - 1 - (d.ox + x) : d.ox + x;
  const float ys = (float)Y * d.sy, xs = (float)X * d.sx;
  const int y0 = (int)floorf(ys), x0 = (int)floorf(xs);
  const int y1 = min((int)ceilf(ys), d.in_h - 1), x1 = min((int)ceilf(xs), d.in_w - 1);
  const float wy = ys - (float)y0, wx = xs - (float)x0;
  const float inv255 = (float)(1.0 / 255);
  const uint8_t* src = blob + d.offset;
  const uint8_t* r0 = src + (size_t)y0 * d.in_w * 3;
  const uint8_t* r1 = src + (size_t)y1 * d.in_w * 3;
  float* o = dst + ((size_t)i * out_h * out_w + p) * 3;
#pragma unroll
  for (int c = 0; c < 3; ++c) {
    const float p00 = (float)r0[x0 * 3 + c] * inv255, p01 = (float)r0[x1 * 3 + c] * inv255;
    const float p10 = (float)r1[x0 * 3 + c] * inv255, p11 = (float)r1[x1 * 3 + c] * inv255;
    const float top = lerp_rn(p00, p01, wx), bot = lerp_rn(p10, p11, wx);
    const float v = lerp_rn(top, bot, wy);
    const float centred = v - 0.5f;
    o[c] = centred * 2.0f;
  }
}

}  // namespace

extern "C" int comic_image_preprocess(const uint8_t* blob, const void* desc, int n, float* dst, int out_h, int out_w,
                                      int resize, void* stream) {
  COMIC_REQUIRE(blob && desc && dst, "image_preprocess: null pointer");
  COMIC_REQUIRE(n > 0 && out_h > 0 && out_w > 0 && resize >= out_h && resize >= out_w, "image_preprocess: bad sizes");
  static_assert(sizeof(ImgDesc) == 40, "comic_image_desc layout");
  hipLaunchKernelGGL(image_preprocess_kernel, dim3(cdiv(out_h * out_w, 256), n), dim3(256), 0, (hipStream_t)stream, blob,
                     (const ImgDesc*)desc, dst, out_h, out_w, resize);
  COMIC_LAUNCH_CHECK("image_preprocess");
  return 0;
}
