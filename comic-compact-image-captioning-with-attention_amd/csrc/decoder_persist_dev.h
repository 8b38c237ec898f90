// Device-side pieces shared by the persistent decoder kernels (decoder_persist.hip, decoder_persist_bwd.hip):
// geometry of a launch, sc1 (write-through / L1-bypassing) buffer accesses, and the validated load of a handed-off
// word ("the data is the flag": see the header of decoder_persist.hip).
#pragma once
#include "decoder_persist.h"

namespace {

constexpr int kThreads = 512;
constexpr int kWaves = 8;
constexpr int kGroupWgs = 64;          // workgroups per batch group
constexpr int kGroupRows = 16;
constexpr int kMaxGroups = 4;         // groups of one launch (256 CUs); larger batches run as consecutive launches
constexpr int kMaxLaunches = 4;       // ... up to this many (beyond, the per-step kernels are better filled)
constexpr int kD = 512;                // = 8 units per workgroup x 64 workgroups
constexpr unsigned kSpinLimit = 1u << 20;
constexpr int kArgRow = 132;           // greedy: floats per row of the partial-argmax hand-off: 64 x (value, column) + stop word
constexpr int kSc1 = 16;               // cache-policy bit of raw buffer loads / stores: sc1

typedef __attribute__((ext_vector_type(4))) unsigned u32x4_t;

__device__ __forceinline__ __amdgpu_buffer_rsrc_t make_rsrc(const void* p, long bytes) {
  return __builtin_amdgcn_make_buffer_rsrc((void*)p, 0, (int)bytes, 0x00020000);
}
__device__ __forceinline__ float4 load16_sc1(__amdgpu_buffer_rsrc_t r, unsigned byte_off) {
  const u32x4_t v = __builtin_amdgcn_raw_buffer_load_b128(r, (int)byte_off, 0, kSc1);
  return make_float4(__uint_as_float(v.x), __uint_as_float(v.y), __uint_as_float(v.z), __uint_as_float(v.w));
}
__device__ __forceinline__ void store16_sc1(__amdgpu_buffer_rsrc_t r, unsigned byte_off, float4 v) {
  const u32x4_t u = {__float_as_uint(v.x), __float_as_uint(v.y), __float_as_uint(v.z), __float_as_uint(v.w)};
  __builtin_amdgcn_raw_buffer_store_b128(u, r, (int)byte_off, 0, kSc1);
}
__device__ __forceinline__ void store4_sc1(__amdgpu_buffer_rsrc_t r, unsigned byte_off, float v) {
  __builtin_amdgcn_raw_buffer_store_b32(__float_as_uint(v), r, (int)byte_off, 0, kSc1);
}

constexpr unsigned kSentinel = COMIC_PERSIST_SENTINEL;   // "not written yet": a NaN pattern no operation produces

__device__ __forceinline__ bool unwritten(float4 v) {
  return __float_as_uint(v.x) == kSentinel || __float_as_uint(v.y) == kSentinel || __float_as_uint(v.z) == kSentinel ||
         __float_as_uint(v.w) == kSentinel;
}

// A wave's view of the launch-wide failure state: once `dead`, waits are skipped (the outputs are garbage anyway and
// comic_persist_check reports it).
struct Waiter {
  unsigned* err;
  bool dead;
  // one more unsuccessful poll; true = give up
  __device__ __forceinline__ bool spin(unsigned& spins) {
    __builtin_amdgcn_s_sleep(1);
    asm volatile("" ::: "memory");                             // the re-read that follows is a new load
    ++spins;
    if ((spins & 255u) == 0 && __hip_atomic_load(err, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) != 0u) dead = true;
    if (spins > kSpinLimit) {
      __hip_atomic_store(err, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      dead = true;
    }
    return dead;
  }
};

// Re-read (sc1) the 16-byte pieces x[i] (i in `want`, a wave-uniform bit set) at byte offsets off[i] until no lane of
// the wave sees an unwritten word in any of them.
template <int N>
__device__ __forceinline__ void wait_written(float4 (&x)[N], __amdgpu_buffer_rsrc_t r, unsigned base,
                                             const unsigned (&off)[N], unsigned want, Waiter& w) {
  if (w.dead) return;
  unsigned spins = 0;
  for (;;) {
    unsigned bad = 0;
#pragma unroll
    for (int i = 0; i < N; ++i)
      if (((want >> i) & 1u) && __any(unwritten(x[i]))) bad |= 1u << i;
    if (!bad) return;
    if (w.spin(spins)) return;
#pragma unroll
    for (int i = 0; i < N; ++i)
      if ((bad >> i) & 1u) x[i] = load16_sc1(r, base + off[i]);
  }
}

}  // namespace
