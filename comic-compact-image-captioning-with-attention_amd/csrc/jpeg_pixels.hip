// Device half of the split JPEG decoder: quantised DCT coefficients (libcomic_jpeg.so, csrc/jpeg_entropy.c) -> uint8 RGB.
//
// Stands where the reference's tf.data map decodes every image with libjpeg on host cores (tf.image.decode_jpeg in
// common/inputs/manager_image_caption.py:163-175 -> preprocessing/inception_preprocessing_radix.py).  The arithmetic is
// libjpeg's, integer for integer, so the pixels are the bits PIL / libjpeg-turbo give for the same file:
//   jidctint.c jpeg_idct_islow  (CONST_BITS 13, PASS1_BITS 2; dequantisation inside),
//   jdsample.c h2v1 / h2v2 fancy upsampling (triangle filter), jdcolor.c ycc_rgb_convert (16-bit fixed point).
// Parity reference: oracle/jpeg_ref.py (pinned against PIL's decode), tests/test_jpeg_split.py.
//
// Two launches per batch, HBM-bound byte work (3 B of coefficients in, 1.5-3 B of planes out and in, 3 B of RGB out per pixel):
//   jpeg_idct_kernel   one thread per 8x8 block: 128 B of coefficients in registers, both passes, 8 rows of 8 bytes out
//                      into the component's plane (neighbouring threads write neighbouring 8-byte pieces of a row)
//   jpeg_colour_kernel one thread per four pixels of a row: Y + the chroma neighbourhood -> 12 bytes of RGB
#include "common.h"
#include "../../include/comic_jpeg.h"

namespace {

constexpr int kConstBits = 13, kPass1Bits = 2;
constexpr int F_0_298631336 = 2446, F_0_390180644 = 3196, F_0_541196100 = 4433, F_0_765366865 = 6270;
constexpr int F_0_899976223 = 7373, F_1_175875602 = 9633, F_1_501321110 = 12299, F_1_847759065 = 15137;
constexpr int F_1_961570560 = 16069, F_2_053119869 = 16819, F_2_562915447 = 20995, F_3_072711026 = 25172;

template <int SHIFT>
__device__ __forceinline__ void idct8(const int (&v)[8], int (&o)[8]) {
  int z2 = v[2], z3 = v[6];
  int z1 = (z2 + z3) * F_0_541196100;
  const int tmp2 = z1 - z3 * F_1_847759065;
  const int tmp3 = z1 + z2 * F_0_765366865;
  const int tmp0 = (v[0] + v[4]) * (1 << kConstBits);
  const int tmp1 = (v[0] - v[4]) * (1 << kConstBits);
  const int tmp10 = tmp0 + tmp3, tmp13 = tmp0 - tmp3, tmp11 = tmp1 + tmp2, tmp12 = tmp1 - tmp2;
  int t0 = v[7], t1 = v[5], t2 = v[3], t3 = v[1];
  z1 = t0 + t3;
  z2 = t1 + t2;
  z3 = t0 + t2;
  int z4 = t1 + t3;
  const int z5 = (z3 + z4) * F_1_175875602;
  t0 *= F_0_298631336;
  t1 *= F_2_053119869;
  t2 *= F_3_072711026;
  t3 *= F_1_501321110;
  z1 *= -F_0_899976223;
  z2 *= -F_2_562915447;
  z3 = z3 * -F_1_961570560 + z5;
  z4 = z4 * -F_0_390180644 + z5;
  t0 += z1 + z3;
  t1 += z2 + z4;
  t2 += z2 + z3;
  t3 += z1 + z4;
  constexpr int R = 1 << (SHIFT - 1);
  o[0] = (tmp10 + t3 + R) >> SHIFT;
  o[7] = (tmp10 - t3 + R) >> SHIFT;
  o[1] = (tmp11 + t2 + R) >> SHIFT;
  o[6] = (tmp11 - t2 + R) >> SHIFT;
  o[2] = (tmp12 + t1 + R) >> SHIFT;
  o[5] = (tmp12 - t1 + R) >> SHIFT;
  o[3] = (tmp13 + t0 + R) >> SHIFT;
  o[4] = (tmp13 - t0 + R) >> SHIFT;
}

__device__ __forceinline__ unsigned clamp_u8(int v) { return (unsigned)min(max(v, 0), 255); }

// dequantisation + both passes + range limit of ONE block (64 int16 in raw, row-major) -> 8 rows of 8 bytes at dst
__device__ __forceinline__ void idct_block(const uint4 (&raw)[8], const int* __restrict__ q, uint8_t* __restrict__ dst, const int stride) {
  int ws[8][8];                                          // [row][column] after pass 1
  // pass 1: columns of the dequantised block
#pragma unroll
  for (int col = 0; col < 8; ++col) {
    int v[8], o[8];
#pragma unroll
    for (int r = 0; r < 8; ++r) {
      const unsigned w = (&raw[r].x)[col >> 1];
      const int cf = (int)(int16_t)(col & 1 ? (w >> 16) : (w & 0xffffu));
      v[r] = cf * q[r * 8 + col];
    }
    idct8<kConstBits - kPass1Bits>(v, o);
#pragma unroll
    for (int r = 0; r < 8; ++r) ws[r][col] = o[r];
  }
  // pass 2: rows; + 128 and the range limit (0..255; the vector code of libjpeg-turbo saturates the same way)
#pragma unroll
  for (int r = 0; r < 8; ++r) {
    int o[8];
    idct8<kConstBits + kPass1Bits + 3>(ws[r], o);
    uint2 pk;
    pk.x = clamp_u8(o[0] + 128) | (clamp_u8(o[1] + 128) << 8) | (clamp_u8(o[2] + 128) << 16) | (clamp_u8(o[3] + 128) << 24);
    pk.y = clamp_u8(o[4] + 128) | (clamp_u8(o[5] + 128) << 8) | (clamp_u8(o[6] + 128) << 16) | (clamp_u8(o[7] + 128) << 24);
    *(uint2*)(dst + (long)r * stride) = pk;
  }
}

// grid (ceil(max blocks of an image / 256), n images)
__global__ __launch_bounds__(256) void jpeg_idct_kernel(const int16_t* __restrict__ coef, const comic_jpeg_info* __restrict__ infos,
                                                        uint8_t* __restrict__ planes) {
  const comic_jpeg_info* in = infos + blockIdx.y;
  __shared__ int quant[3][64];
  if (in->ncomp == 0) return;                          // an image of the loader's PIL path: nothing to do
  const int nc = in->ncomp;
  if (threadIdx.x < 64 * nc) quant[threadIdx.x >> 6][threadIdx.x & 63] = in->quant[threadIdx.x >> 6][threadIdx.x & 63];
  __syncthreads();
  const long g = (long)blockIdx.x * blockDim.x + threadIdx.x;      // block of the image, components back to back
  const long total = in->coef_count >> 6;
  if (g >= total) return;
  int c = 0;
  if (nc == 3) c = g >= (in->coef_off[2] >> 6) ? 2 : (g >= (in->coef_off[1] >> 6) ? 1 : 0);
  const long gb = g - (in->coef_off[c] >> 6);
  const int bw = in->blocks_w[c];
  const int by = (int)(gb / bw), bx = (int)(gb - (long)by * bw);
  const int16_t* src = coef + in->coef_base + (g << 6);
  uint4 raw[8];
#pragma unroll
  for (int r = 0; r < 8; ++r) raw[r] = *(const uint4*)(src + r * 8);
  idct_block(raw, quant[c], planes + in->coef_base + in->coef_off[c] + ((long)by * 8) * (bw * 8) + bx * 8, bw * 8);
}

// The same from the PACKED form of the loader's batches (comic_jpeg.h, comic_jpeg_pool_submit_packed): a thread expands its
// block in LDS (32 words + 1 of padding per thread: the stride keeps the lanes of a wave on different banks), then runs the
// transform above.  packed: 16-bit units; image i at infos[i].pixel_off, its planes at infos[i].coef_base.
__global__ __launch_bounds__(256) void jpeg_idct_packed_kernel(const uint16_t* __restrict__ packed, const comic_jpeg_info* __restrict__ infos,
                                                               uint8_t* __restrict__ planes) {
  const comic_jpeg_info* in = infos + blockIdx.y;
  __shared__ int quant[3][64];
  __shared__ uint32_t blk[256][33];
  if (in->ncomp == 0) return;
  const int nc = in->ncomp;
  if (threadIdx.x < 64 * nc) quant[threadIdx.x >> 6][threadIdx.x & 63] = in->quant[threadIdx.x >> 6][threadIdx.x & 63];
  __syncthreads();
  const long g = (long)blockIdx.x * blockDim.x + threadIdx.x;
  const long total = in->coef_count >> 6;
  if (g >= total) return;
  int c = 0;
  if (nc == 3) c = g >= (in->coef_off[2] >> 6) ? 2 : (g >= (in->coef_off[1] >> 6) ? 1 : 0);
  const long gb = g - (in->coef_off[c] >> 6);
  const int bw = in->blocks_w[c];
  const int by = (int)(gb / bw), bx = (int)(gb - (long)by * bw);
  const uint16_t* img = packed + in->pixel_off;
  const uint32_t d = ((const uint32_t*)img)[g];
  const uint16_t* ent = img + 3 * total + (d >> 7);
  const int n = (int)(d & 127);
  uint32_t* mine = blk[threadIdx.x];
  // (one access type for the LDS block: 16-bit stores through a second pointer type let the compiler move the 32-bit reads
  // below across them)
  auto put = [&](unsigned pos, unsigned v16) {
    const unsigned sh = (pos & 1u) * 16u;
    mine[pos >> 1] = (mine[pos >> 1] & ~(0xffffu << sh)) | ((v16 & 0xffffu) << sh);
  };
#pragma unroll
  for (int k = 0; k < 32; ++k) mine[k] = 0;
  put(0, img[2 * total + g]);                            // DC
  for (int j = 0; j < n; ++j) {
    const unsigned e = ent[j];
    const unsigned pos = e >> 10;
    if (pos) put(pos, (unsigned)((int)(int16_t)(e << 6) >> 6));       // 10-bit value, sign-extended
    else {
      put(e & 63u, ent[j + 1]);
      ++j;
    }
  }
  uint4 raw[8];
#pragma unroll
  for (int r = 0; r < 8; ++r) raw[r] = make_uint4(mine[4 * r], mine[4 * r + 1], mine[4 * r + 2], mine[4 * r + 3]);
  idct_block(raw, quant[c], planes + in->coef_base + in->coef_off[c] + ((long)by * 8) * (bw * 8) + bx * 8, bw * 8);
}

// jdcolor.c build_ycc_rgb_table as arithmetic: FIX(x) = (int)(x * 65536 + 0.5)
constexpr int kFix140200 = 91881, kFix177200 = 116130, kFix071414 = 46802, kFix034414 = 22554, kHalf = 1 << 15;

__device__ __forceinline__ void ycc_store(int y, int cb, int cr, uint8_t* o) {
  const int xb = cb - 128, xr = cr - 128;
  const int r = y + ((kFix140200 * xr + kHalf) >> 16);
  const int g = y + ((-kFix034414 * xb + kHalf - kFix071414 * xr) >> 16);
  const int b = y + ((kFix177200 * xb + kHalf) >> 16);
  o[0] = (uint8_t)clamp_u8(r);
  o[1] = (uint8_t)clamp_u8(g);
  o[2] = (uint8_t)clamp_u8(b);
}

// one chroma sample at full resolution (jdsample.c, fancy upsampling)
template <int HS, int VS>
__device__ __forceinline__ int chroma_at(const uint8_t* __restrict__ pl, int stride, int cw, int ch, int y, int x) {
  if (HS == 1) return pl[(long)y * stride + x];
  const int cx = x >> 1;
  if (VS == 1) {                                         // h2v1: (3 near + far + {1, 2}) >> 2, the ends copied
    const uint8_t* row = pl + (long)y * stride;
    const int t = row[cx];
    if (x & 1) return cx == cw - 1 ? t : (3 * t + row[cx + 1] + 2) >> 2;
    return cx == 0 ? t : (3 * t + row[cx - 1] + 1) >> 2;
  }
  // h2v2: column sums 3 near + far (the row above for even output rows, below for odd ones, the image's first / last
  // real row for a missing one), then (3 this + neighbour + {8, 7}) >> 4; first / last column (4 this + {8, 7}) >> 4
  const int cy = y >> 1;
  const int ny = (y & 1) ? min(cy + 1, ch - 1) : max(cy - 1, 0);
  const uint8_t* r0 = pl + (long)cy * stride;
  const uint8_t* r1 = pl + (long)ny * stride;
  const int t = 3 * r0[cx] + r1[cx];
  if (x & 1) return cx == cw - 1 ? (4 * t + 7) >> 4 : (3 * t + 3 * r0[cx + 1] + r1[cx + 1] + 7) >> 4;
  return cx == 0 ? (4 * t + 8) >> 4 : (3 * t + 3 * r0[cx - 1] + r1[cx - 1] + 8) >> 4;
}

// grid (ceil(max_w / 4 / 64), max_h, n images), block 64: a thread owns four pixels of a row
template <int HS, int VS>
__device__ __forceinline__ void colour_row(const comic_jpeg_info* in, const uint8_t* __restrict__ planes,
                                           uint8_t* __restrict__ pixels, int y, int x0) {
  const int W = in->width;
  const uint8_t* yp = planes + in->coef_base + in->coef_off[0];
  const uint8_t* bp = planes + in->coef_base + in->coef_off[1];
  const uint8_t* rp = planes + in->coef_base + in->coef_off[2];
  const int ys = in->blocks_w[0] * 8, cs = in->blocks_w[1] * 8;
  const int cw = in->comp_w[1], ch = in->comp_h[1];
  uint8_t* out = pixels + in->pixel_off + ((long)y * W + x0) * 3;
#pragma unroll
  for (int j = 0; j < 4; ++j) {
    const int x = x0 + j;
    if (x >= W) break;
    const int yy = yp[(long)y * ys + x];
    const int cb = chroma_at<HS, VS>(bp, cs, cw, ch, y, x), cr = chroma_at<HS, VS>(rp, cs, cw, ch, y, x);
    ycc_store(yy, cb, cr, out + 3 * j);
  }
}

__global__ __launch_bounds__(64) void jpeg_colour_kernel(const comic_jpeg_info* __restrict__ infos,
                                                         const uint8_t* __restrict__ planes, uint8_t* __restrict__ pixels) {
  const comic_jpeg_info* in = infos + blockIdx.z;
  if (in->ncomp == 0) return;
  const int y = blockIdx.y, x0 = (blockIdx.x * 64 + threadIdx.x) * 4;
  if (y >= in->height || x0 >= in->width) return;
  if (in->ncomp == 1) {
    const uint8_t* yp = planes + in->coef_base + (long)y * in->blocks_w[0] * 8;
    uint8_t* out = pixels + in->pixel_off + ((long)y * in->width + x0) * 3;
    for (int j = 0; j < 4 && x0 + j < in->width; ++j) out[3 * j] = out[3 * j + 1] = out[3 * j + 2] = yp[x0 + j];
    return;
  }
  if (in->hmax == 1) colour_row<1, 1>(in, planes, pixels, y, x0);
  else if (in->vmax == 1) colour_row<2, 1>(in, planes, pixels, y, x0);
  else colour_row<2, 2>(in, planes, pixels, y, x0);
}

// ---- component planes -> network input in one launch (the loader's path) ---------------------------------------------------
// comic_image_preprocess (csrc/preprocess.hip: uint8 -> [0,1] -> TF-1 bilinear resize -> flip -> crop -> [-1,1], same float32
// roundings, contraction off) whose four taps per output pixel are converted from the planes on the fly: the RGB image is
// never written (59 MB per 64 images of 640 x 480, and the 88 us of jpeg_colour_kernel).  Images of the PIL path (ncomp == 0)
// are read from the RGB blob as before.
struct ImgDesc {                  // comic_image_desc
  int64_t offset;
  int32_t in_h, in_w;
  int32_t flip, oy, ox;
  float sy, sx;
};

__device__ __forceinline__ float lerp_rn(float a, float b, float w) {
#pragma clang fp contract(off)
  const float d = b - a;
  const float m = d * w;
  return a + m;
}

template <int HS, int VS>
__device__ __forceinline__ void tap_rgb(const comic_jpeg_info* in, const uint8_t* __restrict__ planes, int y, int x, int (&rgb)[3]) {
  const uint8_t* base = planes + in->coef_base;
  const int yy = base[in->coef_off[0] + (long)y * (in->blocks_w[0] * 8) + x];
  const int cs = in->blocks_w[1] * 8, cw = in->comp_w[1], ch = in->comp_h[1];
  const int cb = chroma_at<HS, VS>(base + in->coef_off[1], cs, cw, ch, y, x);
  const int cr = chroma_at<HS, VS>(base + in->coef_off[2], cs, cw, ch, y, x);
  const int xb = cb - 128, xr = cr - 128;
  rgb[0] = (int)clamp_u8(yy + ((kFix140200 * xr + kHalf) >> 16));
  rgb[1] = (int)clamp_u8(yy + ((-kFix034414 * xb + kHalf - kFix071414 * xr) >> 16));
  rgb[2] = (int)clamp_u8(yy + ((kFix177200 * xb + kHalf) >> 16));
}

__device__ __forceinline__ void tap(const comic_jpeg_info* in, const uint8_t* __restrict__ planes, const uint8_t* __restrict__ rgb_src,
                                    int in_w, int y, int x, int (&rgb)[3]) {
  if (in->ncomp == 0) {                                  // decoded by PIL: RGB bytes in the blob
    const uint8_t* p = rgb_src + ((size_t)y * in_w + x) * 3;
    rgb[0] = p[0]; rgb[1] = p[1]; rgb[2] = p[2];
  } else if (in->ncomp == 1) {
    rgb[0] = rgb[1] = rgb[2] = planes[in->coef_base + (long)y * (in->blocks_w[0] * 8) + x];
  } else if (in->hmax == 1) {
    tap_rgb<1, 1>(in, planes, y, x, rgb);
  } else if (in->vmax == 1) {
    tap_rgb<2, 1>(in, planes, y, x, rgb);
  } else {
    tap_rgb<2, 2>(in, planes, y, x, rgb);
  }
}

__global__ __launch_bounds__(256) void jpeg_preprocess_kernel(const comic_jpeg_info* __restrict__ infos,
                                                              const uint8_t* __restrict__ planes, const uint8_t* __restrict__ blob,
                                                              const ImgDesc* __restrict__ desc, float* __restrict__ dst, int out_h,
                                                              int out_w, int resize) {
#pragma clang fp contract(off)
  const int i = blockIdx.y;
  const int p = blockIdx.x * blockDim.x + threadIdx.x;
  if (p >= out_h * out_w) return;
  const comic_jpeg_info* in = infos + i;
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
  int t00[3], t01[3], t10[3], t11[3];
  tap(in, planes, src, d.in_w, y0, x0, t00);
  tap(in, planes, src, d.in_w, y0, x1, t01);
  tap(in, planes, src, d.in_w, y1, x0, t10);
  tap(in, planes, src, d.in_w, y1, x1, t11);
  float* o = dst + ((size_t)i * out_h * out_w + p) * 3;
#pragma unroll
  for (int c = 0; c < 3; ++c) {
    const float p00 = (float)t00[c] * inv255, p01 = (float)t01[c] * inv255;
    const float p10 = (float)t10[c] * inv255, p11 = (float)t11[c] * inv255;
    const float top = lerp_rn(p00, p01, wx), bot = lerp_rn(p10, p11, wx);
    const float v = lerp_rn(top, bot, wy);
    const float centred = v - 0.5f;
    o[c] = centred * 2.0f;
  }
}

}  // namespace

extern "C" int comic_jpeg_preprocess(const int16_t* coef, const void* infos, int n, int max_blocks, uint8_t* planes,
                                     const uint8_t* blob, const void* desc, float* dst, int out_h, int out_w, int resize,
                                     void* stream) {
  COMIC_REQUIRE(coef && infos && planes && desc && dst, "jpeg_preprocess: null pointer");
  COMIC_REQUIRE(n > 0 && n <= 65535 && max_blocks >= 0 && out_h > 0 && out_w > 0 && resize >= out_h && resize >= out_w,
                "jpeg_preprocess: bad sizes");
  COMIC_REQUIRE(((uintptr_t)coef & 15) == 0 && ((uintptr_t)planes & 7) == 0, "jpeg_preprocess: coefficient / plane blob alignment");
  static_assert(sizeof(ImgDesc) == 40, "comic_image_desc layout");
  hipStream_t st = (hipStream_t)stream;
  if (max_blocks > 0) {
    hipLaunchKernelGGL(jpeg_idct_kernel, dim3(cdiv(max_blocks, 256), n), dim3(256), 0, st, coef, (const comic_jpeg_info*)infos,
                       planes);
    COMIC_LAUNCH_CHECK("jpeg_idct");
  }
  hipLaunchKernelGGL(jpeg_preprocess_kernel, dim3(cdiv(out_h * out_w, 256), n), dim3(256), 0, st, (const comic_jpeg_info*)infos,
                     planes, blob, (const ImgDesc*)desc, dst, out_h, out_w, resize);
  COMIC_LAUNCH_CHECK("jpeg_preprocess");
  return 0;
}

extern "C" int comic_jpeg_preprocess_packed(const uint16_t* packed, const void* infos, int n, int max_blocks, uint8_t* planes,
                                            const uint8_t* blob, const void* desc, float* dst, int out_h, int out_w, int resize,
                                            void* stream) {
  COMIC_REQUIRE(packed && infos && planes && desc && dst, "jpeg_preprocess_packed: null pointer");
  COMIC_REQUIRE(n > 0 && n <= 65535 && max_blocks >= 0 && out_h > 0 && out_w > 0 && resize >= out_h && resize >= out_w,
                "jpeg_preprocess_packed: bad sizes");
  COMIC_REQUIRE(((uintptr_t)packed & 3) == 0 && ((uintptr_t)planes & 7) == 0, "jpeg_preprocess_packed: blob alignment");
  hipStream_t st = (hipStream_t)stream;
  if (max_blocks > 0) {
    hipLaunchKernelGGL(jpeg_idct_packed_kernel, dim3(cdiv(max_blocks, 256), n), dim3(256), 0, st, packed,
                       (const comic_jpeg_info*)infos, planes);
    COMIC_LAUNCH_CHECK("jpeg_idct_packed");
  }
  hipLaunchKernelGGL(jpeg_preprocess_kernel, dim3(cdiv(out_h * out_w, 256), n), dim3(256), 0, st, (const comic_jpeg_info*)infos,
                     planes, blob, (const ImgDesc*)desc, dst, out_h, out_w, resize);
  COMIC_LAUNCH_CHECK("jpeg_preprocess");
  return 0;
}

extern "C" int comic_jpeg_pixels(const int16_t* coef, const void* infos, int n, int max_blocks, int max_w, int max_h,
                                 uint8_t* planes, uint8_t* pixels, void* stream) {
  COMIC_REQUIRE(coef && infos && planes && pixels, "jpeg_pixels: null pointer");
  COMIC_REQUIRE(n > 0 && n <= 65535 && max_blocks > 0 && max_w > 0 && max_h > 0 && max_h <= 65535, "jpeg_pixels: bad sizes");
  COMIC_REQUIRE(((uintptr_t)coef & 15) == 0 && ((uintptr_t)planes & 7) == 0, "jpeg_pixels: coefficient / plane blob alignment");
  static_assert(sizeof(comic_jpeg_info) == 512, "comic_jpeg_info layout");
  hipStream_t st = (hipStream_t)stream;
  hipLaunchKernelGGL(jpeg_idct_kernel, dim3(cdiv(max_blocks, 256), n), dim3(256), 0, st, coef, (const comic_jpeg_info*)infos,
                     planes);
  COMIC_LAUNCH_CHECK("jpeg_idct");
  hipLaunchKernelGGL(jpeg_colour_kernel, dim3(cdiv(cdiv(max_w, 4), 64), max_h, n), dim3(64), 0, st,
                     (const comic_jpeg_info*)infos, planes, pixels);
  COMIC_LAUNCH_CHECK("jpeg_colour");
  return 0;
}
