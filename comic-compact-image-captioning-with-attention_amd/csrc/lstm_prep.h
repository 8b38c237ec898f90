// The decode step's operand gather, shared by lstm_stream.hip (its own launch: step 0, and every step of a decode whose
// top-k is not beam_logits.hip's) and beam_logits.hip (fused behind the beam merge: the entry's workgroup knows the new
// words and parents and prepares the NEXT step's operand rows at once).
#pragma once
#include "common.h"

struct LstmPrepArgs {
  const float* table;      // embedding [V][E]
  const float* att;        // [R][A] attention state written by this step (gathered through the parents)
  const float* h;          // [R][D]
  const float* c;          // [R][D]
  uint4* x_frag;           // [16-row tile][k-step][hi, lo][lane]: the LSTM operand rows as bf16 hi / lo fragments
  float* c_in;             // [R][D] gathered cell state
  int E, A, D, V, KS;
};

// Segment sg of row r (source row src, word id): sg < 4 KS: 8 consecutive operand features (embedding | attention |
// hidden, zero beyond E + A + D) split into hi / lo and stored as lane (r % 16, sg % 4) of k-step sg / 4; the remaining
// D / 8 segments copy the cell state.  E, A and D are multiples of 8, so a segment has one source.
// (load and store halves apart, so that a caller with several segments per thread can have all its loads in flight)
__device__ __forceinline__ void lstm_prep_load(const LstmPrepArgs& p, int src, int id, bool live, int sg, float4& lo4,
                                               float4& hi4) {
  const int Wd = p.E + p.A + p.D;
  lo4 = make_float4(0.f, 0.f, 0.f, 0.f);
  hi4 = lo4;
  if (!live) return;
  const float* s = nullptr;
  if (sg >= p.KS * 4) {
    s = p.c + (size_t)src * p.D + (sg - p.KS * 4) * 8;
  } else {
    const int k = sg * 8;
    if (k >= Wd) return;
    if (k < p.E) {
      if (id >= 0 && id < p.V) s = p.table + (size_t)id * p.E + k;
    } else if (k < p.E + p.A) {
      s = p.att + (size_t)src * p.A + (k - p.E);
    } else {
      s = p.h + (size_t)src * p.D + (k - p.E - p.A);
    }
  }
  if (s) {
    lo4 = *(const float4*)s;
    hi4 = *(const float4*)(s + 4);
  }
}
__device__ __forceinline__ void lstm_prep_store(const LstmPrepArgs& p, int r, bool live, int sg, const float4& lo4,
                                                const float4& hi4) {
  if (sg >= p.KS * 4) {
    if (!live) return;
    float4* q = (float4*)(p.c_in + (size_t)r * p.D + (sg - p.KS * 4) * 8);
    q[0] = lo4;
    q[1] = hi4;
    return;
  }
  const float x[8] = {lo4.x, lo4.y, lo4.z, lo4.w, hi4.x, hi4.y, hi4.z, hi4.w};
  uint32_t wh[4], wl[4];
#pragma unroll
  for (int j = 0; j < 4; ++j) {
    wh[j] = pack_bf16x2(x[2 * j], x[2 * j + 1]);
    wl[j] = pack_bf16x2(x[2 * j] - __uint_as_float(wh[j] << 16), x[2 * j + 1] - __uint_as_float(wh[j] & 0xFFFF0000u));
  }
  uint4* o = p.x_frag + ((size_t)(r >> 4) * p.KS + (sg >> 2)) * 128 + (sg & 3) * 16 + (r & 15);
  o[0] = make_uint4(wh[0], wh[1], wh[2], wh[3]);
  o[64] = make_uint4(wl[0], wl[1], wl[2], wl[3]);
}
__device__ __forceinline__ void lstm_prep_segment(const LstmPrepArgs& p, int r, int src, int id, bool live, int sg) {
  float4 a, b;
  lstm_prep_load(p, src, id, live, sg, a, b);
  lstm_prep_store(p, r, live, sg, a, b);
}
