// Device side of the streaming skinny product (lstm_stream.hip), shared with beam_logits.hip, whose projection launch
// carries the query projection's workgroups on the CUs its own 229 leave idle.
#pragma once
#include "conv_common.h"

constexpr int kStreamStepBytes = 8 * 1024;     // one chunk, one k32-step: 4 tiles x {hi, lo} x 64 lanes x 16 B

struct LstmStreamArgs {
  const uint4* k_frag;
  const uint4* x_frag;
  float* part;             // [S][R][N] partial sums
  int R, N, KS, ksteps, gstride, cstride;
  const int32_t* stop;
  int stop_t;
};

template <int NT>
__device__ __forceinline__ void lstm_stream_wave(const LstmStreamArgs& a, unsigned char* smem, int wave, int lane, int tid,
                                                 int bid) {
  const int fr = lane & 15, fg = lane >> 4;
  const int chunks = a.N / 64, c = bid % chunks, sl = bid / chunks, KS = a.KS;
  const int s0 = sl * a.ksteps, n = min(a.ksteps, KS - s0);
  const unsigned char* wsrc = (const unsigned char*)(a.k_frag + ((size_t)c * KS + s0) * 512);
  for (int i = 0; i < n; ++i) dma16(wsrc + (size_t)i * kStreamStepBytes + tid * 16, smem + i * kStreamStepBytes + wave * 1024);
  constexpr int NR = NT > 0 ? NT : 1;
  int row[NR];
  const uint4* ysrc[NR];
#pragma unroll
  for (int m = 0; m < NR; ++m) {
    const int r = (wave + 8 * m) * 16 + fr;
    row[m] = r < a.R ? r : -1;
    ysrc[m] = a.x_frag + ((size_t)(wave + 8 * m) * KS + s0) * 128 + lane;
  }
  uint4 yf[NR][4][2];
  auto load_y = [&](int s, int slot) {
#pragma unroll
    for (int m = 0; m < NT; ++m) {
      yf[m][slot][0] = ysrc[m][(size_t)s * 128];
      yf[m][slot][1] = ysrc[m][(size_t)s * 128 + 64];
    }
  };
  f32x4_t acc[NR][4];
#pragma unroll
  for (int m = 0; m < NR; ++m)
#pragma unroll
    for (int g = 0; g < 4; ++g) acc[m][g] = (f32x4_t){0.f, 0.f, 0.f, 0.f};
#pragma unroll
  for (int u = 0; u < 4; ++u)
    if (u < n) load_y(u, u);
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  __builtin_amdgcn_s_barrier();                      // the whole slice of the kernel is in the LDS
  if constexpr (NT == 0) return;
  const uint4* wl = (const uint4*)smem + lane;
  for (int g0 = 0; g0 < n; g0 += 4) {
#pragma unroll
    for (int u = 0; u < 4; ++u) {
      const int s = g0 + u;
      if (s < n) {                                   // wave-uniform
        uint4 wa[8];
#pragma unroll
        for (int j = 0; j < 8; ++j) wa[j] = wl[(s * 8 + j) * 64];
#pragma unroll
        for (int g = 0; g < 4; ++g) {
          const bf16x8_t ah = __builtin_bit_cast(bf16x8_t, wa[2 * g]);
          const bf16x8_t al = __builtin_bit_cast(bf16x8_t, wa[2 * g + 1]);
#pragma unroll
          for (int m = 0; m < NT; ++m)
            acc[m][g] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(ah, __builtin_bit_cast(bf16x8_t, yf[m][u][0]), acc[m][g], 0, 0, 0);
#pragma unroll
          for (int m = 0; m < NT; ++m)
            acc[m][g] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(ah, __builtin_bit_cast(bf16x8_t, yf[m][u][1]), acc[m][g], 0, 0, 0);
#pragma unroll
          for (int m = 0; m < NT; ++m)
            acc[m][g] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(al, __builtin_bit_cast(bf16x8_t, yf[m][u][0]), acc[m][g], 0, 0, 0);
        }
        if (s + 4 < n) load_y(s + 4, u);             // this step's registers refill four steps ahead
      }
    }
  }
  // lane (fr, fg) holds, of row `row[m]`, columns g gstride + c cstride + 4 fg + i in acc[m][g][i]
#pragma unroll
  for (int m = 0; m < NT; ++m) {
    if (row[m] < 0) continue;
    float* o = a.part + ((size_t)sl * a.R + row[m]) * a.N + c * a.cstride + 4 * fg;
#pragma unroll
    for (int g = 0; g < 4; ++g)
      *(float4*)(o + (size_t)g * a.gstride) = make_float4(acc[m][g][0], acc[m][g][1], acc[m][g][2], acc[m][g][3]);
  }
}


// workgroup `bid` (= slice * chunks + chunk) of the product p; every wave of the workgroup calls it
__device__ __forceinline__ void lstm_stream_block(const LstmStreamArgs& p, unsigned char* smem, int wave, int lane, int tid,
                                                  int bid) {
  const int tiles = (p.R + 15) >> 4;
  const int nt = (wave < tiles ? 1 : 0) + (wave + 8 < tiles ? 1 : 0);
  if (nt == 2) lstm_stream_wave<2>(p, smem, wave, lane, tid, bid);
  else if (nt == 1) lstm_stream_wave<1>(p, smem, wave, lane, tid, bid);
  else lstm_stream_wave<0>(p, smem, wave, lane, tid, bid);
}
