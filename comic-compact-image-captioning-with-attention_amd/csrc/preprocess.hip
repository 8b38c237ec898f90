// Image preprocessing on the device: the decoded uint8 RGB images of a batch -> the network input.
//
// Replaces inception_preprocessing_radix.preprocess_image for the captioning pipeline (reference
// common/inputs/preprocessing/inception_preprocessing_radix.py:158-278 as used by
// manager_image_caption.py:111-228): uint8 -> float [0,1] -> TF-1 bilinear resize to 256x256
// (tf.image.resize_images, align_corners=False: src = dst * in/out, no half-pixel offset) -> optional horizontal
// flip -> 224x224 (or any h x w) crop at (oy, ox) -> (x - 0.5) * 2.  The host keeps JPEG decoding only.  float32
// arithmetic at TF-1.9's rounding points (convert_image_dtype: cast * (1/255); resize_bilinear_op.cc compute_lerp:
// top = tl + (tr - tl) * xl, bottom likewise, out = top + (bottom - top) * yl); every operation is an explicit _rn
// intrinsic so nothing is contracted into an FMA.  Parity reference: oracle/preprocess_ref.py (bit-identical).
#include "common.h"

namespace {

struct ImgDesc {
  int64_t offset;     // byte offset of the image in the uint8 blob (H x W x 3, row-major)
  int32_t in_h, in_w;
  int32_t flip, oy, ox;
  float sy, sx;       // float32(in_h / 256), float32(in_w / 256)
};

// a + (b - a) * w, three roundings (resize_bilinear_op.cc compute_lerp [TF-1.9])
__device__ __forceinline__ float lerp_rn(float a, float b, float w) {
  return __fadd_rn(a, __fmul_rn(__fsub_rn(b, a), w));
}

__global__ __launch_bounds__(256) void image_preprocess_kernel(const uint8_t* __restrict__ blob,
                                                               const ImgDesc* __restrict__ desc, float* __restrict__ dst,
                                                               int out_h, int out_w, int resize) {
  const int i = blockIdx.y;
  const int p = blockIdx.x * blockDim.x + threadIdx.x;
  if (p >= out_h * out_w) return;
  const ImgDesc d = desc[i];
  const int y = p / out_w, x = p % out_w;
  const int Y = d.oy + y;
  const int X = d.flip ? resize - 1 - (d.ox + x) : d.ox + x;
  const float ys = __fmul_rn((float)Y, d.sy), xs = __fmul_rn((float)X, d.sx);
  const int y0 = (int)floorf(ys), x0 = (int)floorf(xs);
  const int y1 = min((int)ceilf(ys), d.in_h - 1), x1 = min((int)ceilf(xs), d.in_w - 1);
  const float wy = __fsub_rn(ys, (float)y0), wx = __fsub_rn(xs, (float)x0);
  const float inv255 = (float)(1.0 / 255);
  const uint8_t* src = blob + d.offset;
  const uint8_t* r0 = src + (size_t)y0 * d.in_w * 3;
  const uint8_t* r1 = src + (size_t)y1 * d.in_w * 3;
  float* o = dst + ((size_t)i * out_h * out_w + p) * 3;
#pragma unroll
  for (int c = 0; c < 3; ++c) {
    const float p00 = __fmul_rn((float)r0[x0 * 3 + c], inv255), p01 = __fmul_rn((float)r0[x1 * 3 + c], inv255);
    const float p10 = __fmul_rn((float)r1[x0 * 3 + c], inv255), p11 = __fmul_rn((float)r1[x1 * 3 + c], inv255);
    const float top = lerp_rn(p00, p01, wx), bot = lerp_rn(p10, p11, wx);
    const float v = lerp_rn(top, bot, wy);
    o[c] = __fmul_rn(__fsub_rn(v, 0.5f), 2.0f);
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
