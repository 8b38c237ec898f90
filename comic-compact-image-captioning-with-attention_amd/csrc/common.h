// Shared device/host helpers for the COMIC gfx950 kernels.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

#include "../../include/comic_hip.h"

typedef unsigned short bf16_t;  // raw bf16 storage

typedef __attribute__((ext_vector_type(8))) __bf16 bf16x8_t;
typedef __attribute__((ext_vector_type(4))) float f32x4_t;

void comic_set_error(const char* fmt, ...);

// Decode loops (beam search): `steps_executed` in device memory becomes t0 + 1 at the first step t0 after which every
// beam is finished.  The executor publishes (pointer, t) here for the launches of step t, the launch helpers copy it
// into their kernel arguments, and the per-step kernels return at once when the loop has already ended
// (`comic_stopped`): the remaining steps of a fixed-length, graph-replayed loop cost a few microseconds of empty
// launches instead of full decode steps -- dynamic_decode's early exit without a host round trip.  Null elsewhere.
struct ComicStop {
  const int32_t* p = nullptr;
  int t = 0;
};
extern thread_local ComicStop g_comic_stop;
__device__ __forceinline__ bool comic_stopped(const int32_t* p, int t) { return p && p[0] <= t; }

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

// Wave-wide reductions on the DPP data path (VALU-latency lane exchanges) instead of __shfl_xor
// (ds_bpermute: an LDS round trip per step): 4 row rotations give every lane its 16-lane row total,
// row_bcast:15 / row_bcast:31 fold the four rows into lane 63, v_readlane broadcasts it.
template <int CTRL, int ROW_MASK = 0xF>
__device__ __forceinline__ float dpp_move(float old, float v) {
  return __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(__builtin_bit_cast(int, old), __builtin_bit_cast(int, v),
                                                                CTRL, ROW_MASK, 0xF, false));
}
__device__ __forceinline__ float wave_sum(float v) {
  v += dpp_move<0x121>(0.f, v);         // row_ror:1
  v += dpp_move<0x122>(0.f, v);         // row_ror:2
  v += dpp_move<0x124>(0.f, v);         // row_ror:4
  v += dpp_move<0x128>(0.f, v);         // row_ror:8
  v += dpp_move<0x142, 0xA>(0.f, v);    // row_bcast:15 into rows 1, 3
  v += dpp_move<0x143, 0xC>(0.f, v);    // row_bcast:31 into rows 2, 3
  return __builtin_bit_cast(float, __builtin_amdgcn_readlane(__builtin_bit_cast(int, v), 63));
}
__device__ __forceinline__ float wave_max(float v) {
  v = fmaxf(v, dpp_move<0x121>(v, v));
  v = fmaxf(v, dpp_move<0x122>(v, v));
  v = fmaxf(v, dpp_move<0x124>(v, v));
  v = fmaxf(v, dpp_move<0x128>(v, v));
  v = fmaxf(v, dpp_move<0x142, 0xA>(v, v));
  v = fmaxf(v, dpp_move<0x143, 0xC>(v, v));
  return __builtin_bit_cast(float, __builtin_amdgcn_readlane(__builtin_bit_cast(int, v), 63));
}
// sum over aligned groups of `n` consecutive lanes (n = 2, 4, 8 or 16): quad permutes, then the
// half-row and row mirrors pair each lane with one from the other half of its group
__device__ __forceinline__ float group_sum_dpp(float v, int n) {
  v += dpp_move<0xB1>(0.f, v);                    // quad_perm [1,0,3,2]
  if (n >= 4) v += dpp_move<0x4E>(0.f, v);        // quad_perm [2,3,0,1]
  if (n >= 8) v += dpp_move<0x141>(0.f, v);       // row_half_mirror
  if (n >= 16) v += dpp_move<0x140>(0.f, v);      // row_mirror
  return v;
}

static inline int cdiv(int a, int b) { return (a + b - 1) / b; }
static inline int64_t cdiv64(int64_t a, int64_t b) { return (a + b - 1) / b; }

// One-time-per-DEVICE latch (hipFuncSetAttribute(MaxDynamicSharedMemorySize) applies to the current device's copy of a
// kernel): slot() is the current device's flag; when the device cannot be told the flag reads false every time, i.e.
// the attribute is set again before every launch.
struct PerDeviceOnce {
  bool done[64] = {};
  bool never = false;
  bool& slot() {
    int dev = 0;
    if (hipGetDevice(&dev) != hipSuccess || dev < 0 || dev >= 64) {
      never = false;
      return never;
    }
    return done[dev];
  }
};
