// Shared device/host helpers for the COMIC gfx950 kernels.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

#include "../../include/comic_hip.h"

typedef unsigned short bf16_t;  // raw bf16 storage

typedef __attribute__((ext_vector_type(8))) __bf16 bf16x8_t;
typedef __attribute__((ext_vector_type(4))) float f32x4_t;

void comic_set_error(const char* fmt, ...);

#define COMIC_LAUNCH_CHECK(name)                                               \
  do {                                                                         \
    hipError_t e__ = hipGetLastError();                                        \
    if (e__ != hipSuccess) {                                                   \
      comic_set_error("%s: launch failed: %s", name, hipGetErrorString(e__));  \
      return 1;                                                                \
    }                                                                          \
  } while (0)

#define COMIC_REQUIRE(cond, ...)      \
  do {                                \
    if (!(cond)) {                    \
      comic_set_error(__VA_ARGS__);   \
      return 2;                       \
    }                                 \
  } while (0)

__device__ __forceinline__ float bf16_to_f32(bf16_t v) {
  return __uint_as_float(((uint32_t)v) << 16);
}
// round-to-nearest-even (finite inputs; NaN handling is not needed on this path)
__device__ __forceinline__ bf16_t f32_to_bf16(float f) {
  uint32_t u = __float_as_uint(f);
  u += 0x7FFFu + ((u >> 16) & 1u);
  return (bf16_t)(u >> 16);
}
// two floats -> packed bf16 pair with the gfx950 conversion instruction (v_cvt_pk_bf16_f32, RNE:
// bit-identical to f32_to_bf16 for finite inputs)
typedef __attribute__((ext_vector_type(2))) float f32x2_t;
typedef __attribute__((ext_vector_type(2))) __bf16 bf16x2_t;
__device__ __forceinline__ uint32_t pack_bf16x2(float lo, float hi) {
  const f32x2_t v = {lo, hi};
  return __builtin_bit_cast(uint32_t, __builtin_convertvector(v, bf16x2_t));
}

__device__ __forceinline__ float wave_sum(float v) {
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
  return v;
}
__device__ __forceinline__ float wave_max(float v) {
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) v = fmaxf(v, __shfl_xor(v, o, 64));
  return v;
}

static inline int cdiv(int a, int b) { return (a + b - 1) / b; }
static inline int64_t cdiv64(int64_t a, int64_t b) { return (a + b - 1) / b; }
