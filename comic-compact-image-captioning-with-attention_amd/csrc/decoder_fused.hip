// Fused per-step GEMM kernels of the attention-LSTM decoder (fp32, gfx950).
//
// The recurrence is a chain of ~10 us kernels whose cost is latency, not arithmetic: at batch 64
// a step's LSTM product is 100 MFLOP.  These kernels keep one whole output element inside one
// workgroup (8 waves split the reduction dimension, fixed-order LDS combine), so the element-wise
// consumer of the product runs as the epilogue instead of as a second kernel fed by split-K
// partials:
//   lstm_step_fused_kernel ... [x;att;h]_t * K + b -> BasicLSTMCell gates, state select,
//                              output dropout, next step's recurrent operand
//                              (src/model_base.py:618-647, tf BasicLSTMCell; impute_finished)
//   input_grad_fused_kernel .. d gates * K^T -> d embedding (input dropout), d attention state,
//                              d h state   (backward of the cell_input_fn concat, ops_rnn.py:696-701)
// Both use v_mfma_f32_16x16x4_f32 (exact fp32 products) with operands loaded straight from
// global memory into registers: every operand element is used by exactly one wave.
#include <stdlib.h>

#include "common.h"

namespace {

__device__ __forceinline__ float sigmoid_(float x) { return 1.0f / (1.0f + expf(-x)); }

constexpr int kFusedWaves = 8;
constexpr int kFusedThreads = kFusedWaves * 64;
constexpr int kChunk = 8;  // k16-blocks a wave keeps in flight

// Weight panels.  A wave's weight fragment for one k16-block is 16 rows x 16 k (1 KiB); read from
// the row-major parameter it is 16-byte pieces of 16..64 different 128-B lines.  The executors
// therefore repack the LSTM kernel once per call into [n-tile][k16-block][16 rows][16 k] panels, so
// that every fragment load is one contiguous KiB:
//   forward panel  (mode 0): n-tile = 4 hidden units; row r = gate (r & 3) of unit 4*tile + (r >> 2)
//                            -> element K[k][gate*D + unit]
//   backward panel (mode 1): n-tile = 16 input features; row r = feature 16*tile + r -> element K[feature][k]
// mode 1 is generic: a row-major [rows = Wd][cols = 4*D] matrix (also used for W_q with rows = cols).
__global__ void pack_lstm_panels_kernel(const float* __restrict__ K, float* __restrict__ out, int D, int Wd,
                                        int mode, long total, int cols) {
  const long i = (long)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= total) return;
  const int N4 = mode == 0 ? 4 * D : cols;
  const int kk = (int)(i & 15), r = (int)((i >> 4) & 15);
  const long blk = i >> 8;
  float v = 0.f;
  if (mode == 0) {
    const int KB = (Wd + 15) >> 4;
    const int kb = (int)(blk % KB), tile = (int)(blk / KB);
    const int k = kb * 16 + kk, unit = tile * 4 + (r >> 2);
    if (k < Wd && unit < D) v = K[(size_t)k * N4 + (r & 3) * D + unit];
  } else {
    const int KB = N4 >> 4;
    const int kb = (int)(blk % KB), tile = (int)(blk / KB);
    const int n = tile * 16 + r;
    if (n < Wd) v = K[(size_t)n * N4 + kb * 16 + kk];
  }
  out[i] = v;
}

struct LstmStepArgs {
  const float* xh;   // [B][ld_xh] operand rows [x ; att ; h]
  int ld_xh;
  const float* K;    // forward panel of the [Wd][4D] LSTM kernel
  const float* bias; // [4D]
  const float* c_prev;
  const float* h_prev;
  float* gates_act;  // [B][4D] activated gates (i, j, f, o) for the backward pass, or NULL
  float* c_new;      // [B][D] cell state before the finished-row select, or NULL
  float* y;          // [B][D] cell output after output dropout, or NULL
  const float* mask_out;
  float keep_out;
  const int32_t* lens;
  int t;
  float* c_state;
  float* h_state;
  float* xh_next;    // recurrent part of the next step's operand rows, or NULL
  int xh_ld;
  int B, D, Wd;
  const int32_t* stop = nullptr;   // decode loops: see comic_stopped (common.h)
  int stop_t = 0;
};

// grid (D/4, ceil(B/(16*MT))): a workgroup owns 4 hidden units (16 gate columns) of 16*MT batch rows.
template <int MT>
__global__ __launch_bounds__(kFusedThreads) void lstm_step_fused_kernel(LstmStepArgs a) {
  if (comic_stopped(a.stop, a.stop_t)) return;
  __shared__ float4 red[kFusedWaves - 1][MT][64];
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int r = lane & 15, kq = lane >> 4;
  const int u0 = blockIdx.x * 4, m0 = blockIdx.y * 16 * MT;
  const int D = a.D, N4 = 4 * D, Wd = a.Wd;
  const float* xrow[MT];
  bool mok[MT];
#pragma unroll
  for (int j = 0; j < MT; ++j) {
    const int m = m0 + 16 * j + r;
    mok[j] = m < a.B;
    xrow[j] = a.xh + (size_t)(mok[j] ? m : 0) * a.ld_xh;
  }
  const int KB = (Wd + 15) >> 4;
  // panel rows of this unit tile: row r = gate (r & 3) of unit u0 + (r >> 2); lane reads k = 4*kq..+3
  const float* wpanel = a.K + ((size_t)blockIdx.x * KB * 16 + r) * 16 + 4 * kq;
  // wave 0 runs the epilogue for (row m0 + 16j + r, unit u0 + kq): fetch its inputs before the product
  float e_b[4] = {0.f, 0.f, 0.f, 0.f}, e_cp[MT], e_hp[MT], e_mask[MT];
  bool e_fin[MT];
  const int d = u0 + kq;
  const bool e_wave = wave == 0 && d < D;
  if (e_wave) {
    if (a.bias) {
      e_b[0] = a.bias[d]; e_b[1] = a.bias[D + d]; e_b[2] = a.bias[2 * D + d]; e_b[3] = a.bias[3 * D + d];
    }
#pragma unroll
    for (int j = 0; j < MT; ++j) {
      const int m = m0 + 16 * j + r;
      const size_t i = (size_t)(mok[j] ? m : 0) * D + d;
      e_cp[j] = a.c_prev ? a.c_prev[i] : 0.f;
      e_hp[j] = a.h_prev ? a.h_prev[i] : 0.f;
      e_mask[j] = a.mask_out ? a.mask_out[i] : 1.f;
      e_fin[j] = a.lens && mok[j] && (a.t >= a.lens[m]);
    }
  }
  f32x4_t acc[MT];
#pragma unroll
  for (int j = 0; j < MT; ++j) acc[j] = (f32x4_t){0.f, 0.f, 0.f, 0.f};
  constexpr int CH = MT >= 4 ? 4 : kChunk;   // k16-blocks in flight per wave (register budget)
  // a wave takes PAIRS of adjacent k16-blocks: its two activation loads per row share a 128-B line
  for (int kp0 = wave; 2 * kp0 < KB; kp0 += kFusedWaves * (CH / 2)) {
    float4 xa[CH][MT], wb[CH];
#pragma unroll
    for (int i = 0; i < CH; ++i) {
      const int kb = 2 * (kp0 + kFusedWaves * (i >> 1)) + (i & 1);
      const int k = kb * 16 + 4 * kq;
      wb[i] = make_float4(0.f, 0.f, 0.f, 0.f);
      if (kb < KB) wb[i] = *(const float4*)(wpanel + (size_t)kb * 256);
#pragma unroll
      for (int j = 0; j < MT; ++j) {
        xa[i][j] = make_float4(0.f, 0.f, 0.f, 0.f);
        if (kb < KB && mok[j] && k < Wd) xa[i][j] = *(const float4*)(xrow[j] + k);   // Wd % 4 == 0 (host check)
      }
    }
#pragma unroll
    for (int i = 0; i < CH; ++i) {
#pragma unroll
      for (int j = 0; j < MT; ++j) acc[j] = __builtin_amdgcn_mfma_f32_16x16x4f32(wb[i].x, xa[i][j].x, acc[j], 0, 0, 0);
#pragma unroll
      for (int j = 0; j < MT; ++j) acc[j] = __builtin_amdgcn_mfma_f32_16x16x4f32(wb[i].y, xa[i][j].y, acc[j], 0, 0, 0);
#pragma unroll
      for (int j = 0; j < MT; ++j) acc[j] = __builtin_amdgcn_mfma_f32_16x16x4f32(wb[i].z, xa[i][j].z, acc[j], 0, 0, 0);
#pragma unroll
      for (int j = 0; j < MT; ++j) acc[j] = __builtin_amdgcn_mfma_f32_16x16x4f32(wb[i].w, xa[i][j].w, acc[j], 0, 0, 0);
    }
  }
  // D[n][m]: lane (m = lane & 15, kq) holds the 4 gates (n = 4*kq + g) of unit u0 + kq of row m
  if (wave > 0) {
#pragma unroll
    for (int j = 0; j < MT; ++j) red[wave - 1][j][lane] = make_float4(acc[j][0], acc[j][1], acc[j][2], acc[j][3]);
  }
  __syncthreads();
  if (!e_wave) return;
#pragma unroll
  for (int j = 0; j < MT; ++j) {
#pragma unroll
    for (int w = 0; w < kFusedWaves - 1; ++w) {
      const float4 p = red[w][j][lane];
      acc[j][0] += p.x; acc[j][1] += p.y; acc[j][2] += p.z; acc[j][3] += p.w;
    }
    const int b = m0 + 16 * j + r;
    if (b >= a.B) continue;
    const size_t i = (size_t)b * D + d;
    const float gi = acc[j][0] + e_b[0], gj = acc[j][1] + e_b[1], gf = acc[j][2] + e_b[2], go = acc[j][3] + e_b[3];
    const float si = sigmoid_(gi), tj = tanhf(gj);
    const float sf = sigmoid_(gf + 1.0f), so = sigmoid_(go);   // forget_bias = 1
    const float cp = e_cp[j];
    const float c2 = cp * sf + si * tj;
    const float h2 = tanhf(c2) * so;
    if (a.gates_act) {
      float* ga = a.gates_act + (size_t)b * N4;
      ga[d] = si; ga[D + d] = tj; ga[2 * D + d] = sf; ga[3 * D + d] = so;
    }
    if (a.c_new) a.c_new[i] = c2;
    if (a.y) a.y[i] = a.mask_out ? (h2 / a.keep_out) * e_mask[j] : h2;
    const bool fin = e_fin[j];
    if (a.c_state) a.c_state[i] = fin ? cp : c2;
    const float hs = fin ? e_hp[j] : h2;
    if (a.h_state) a.h_state[i] = hs;
    if (a.xh_next) a.xh_next[(size_t)b * a.xh_ld + d] = hs;
  }
}

// Wider variant: a workgroup owns NT adjacent unit tiles (4*NT hidden units) of 16 batch rows, so an
// activation fragment feeds NT weight fragments and there are NT times fewer workgroups; after the
// cross-wave combine, wave j runs the epilogue of unit tile j.
template <int NT>
__global__ __launch_bounds__(kFusedThreads) void lstm_step_fused_wide_kernel(LstmStepArgs a) {
  if (comic_stopped(a.stop, a.stop_t)) return;
  static_assert(NT >= 2 && NT <= kFusedWaves, "unit tiles per workgroup");
  __shared__ float4 red[kFusedWaves][NT][64];
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int r = lane & 15, kq = lane >> 4;
  const int t0 = blockIdx.x * NT, m0 = blockIdx.y * 16;
  const int D = a.D, N4 = 4 * D, Wd = a.Wd;
  const int n_tiles = D / 4;
  const int m = m0 + r;
  const bool mok = m < a.B;
  const float* xrow = a.xh + (size_t)(mok ? m : 0) * a.ld_xh;
  const int KB = (Wd + 15) >> 4;
  const float* wpanel = a.K + ((size_t)t0 * KB * 16 + r) * 16 + 4 * kq;   // tile j: + j*KB*256 floats
  // epilogue operands of wave j < NT: (row m0 + r, unit 4*(t0 + j) + kq)
  const int et = t0 + wave;
  const int d = 4 * et + kq;
  const bool e_wave = wave < NT && et < n_tiles && d < D;
  float e_b[4] = {0.f, 0.f, 0.f, 0.f}, e_cp = 0.f, e_hp = 0.f, e_mask = 1.f;
  bool e_fin = false;
  if (e_wave) {
    if (a.bias) {
      e_b[0] = a.bias[d]; e_b[1] = a.bias[D + d]; e_b[2] = a.bias[2 * D + d]; e_b[3] = a.bias[3 * D + d];
    }
    const size_t i = (size_t)(mok ? m : 0) * D + d;
    e_cp = a.c_prev ? a.c_prev[i] : 0.f;
    e_hp = a.h_prev ? a.h_prev[i] : 0.f;
    e_mask = a.mask_out ? a.mask_out[i] : 1.f;
    e_fin = a.lens && mok && (a.t >= a.lens[m]);
  }
  f32x4_t acc[NT];
#pragma unroll
  for (int j = 0; j < NT; ++j) acc[j] = (f32x4_t){0.f, 0.f, 0.f, 0.f};
  constexpr int CH = 4;
  for (int kp0 = wave; 2 * kp0 < KB; kp0 += kFusedWaves * (CH / 2)) {
    float4 xa[CH], wb[CH][NT];
#pragma unroll
    for (int i = 0; i < CH; ++i) {
      const int kb = 2 * (kp0 + kFusedWaves * (i >> 1)) + (i & 1);
      const int k = kb * 16 + 4 * kq;
      xa[i] = make_float4(0.f, 0.f, 0.f, 0.f);
      if (kb < KB && mok && k < Wd) xa[i] = *(const float4*)(xrow + k);
#pragma unroll
      for (int j = 0; j < NT; ++j) {
        wb[i][j] = make_float4(0.f, 0.f, 0.f, 0.f);
        if (kb < KB && t0 + j < n_tiles) wb[i][j] = *(const float4*)(wpanel + ((size_t)j * KB + kb) * 256);
      }
    }
#pragma unroll
    for (int i = 0; i < CH; ++i) {
#pragma unroll
      for (int j = 0; j < NT; ++j) acc[j] = __builtin_amdgcn_mfma_f32_16x16x4f32(wb[i][j].x, xa[i].x, acc[j], 0, 0, 0);
#pragma unroll
      for (int j = 0; j < NT; ++j) acc[j] = __builtin_amdgcn_mfma_f32_16x16x4f32(wb[i][j].y, xa[i].y, acc[j], 0, 0, 0);
#pragma unroll
      for (int j = 0; j < NT; ++j) acc[j] = __builtin_amdgcn_mfma_f32_16x16x4f32(wb[i][j].z, xa[i].z, acc[j], 0, 0, 0);
#pragma unroll
      for (int j = 0; j < NT; ++j) acc[j] = __builtin_amdgcn_mfma_f32_16x16x4f32(wb[i][j].w, xa[i].w, acc[j], 0, 0, 0);
    }
  }
#pragma unroll
  for (int j = 0; j < NT; ++j) red[wave][j][lane] = make_float4(acc[j][0], acc[j][1], acc[j][2], acc[j][3]);
  __syncthreads();
  if (!e_wave) return;
  float g[4] = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
  for (int w = 0; w < kFusedWaves; ++w) {      // fixed order: deterministic
    const float4 p = red[w][wave][lane];
    g[0] += p.x; g[1] += p.y; g[2] += p.z; g[3] += p.w;
  }
  if (!mok) return;
  const size_t i = (size_t)m * D + d;
  const float si = sigmoid_(g[0] + e_b[0]), tj = tanhf(g[1] + e_b[1]);
  const float sf = sigmoid_(g[2] + e_b[2] + 1.0f), so = sigmoid_(g[3] + e_b[3]);   // forget_bias = 1
  const float c2 = e_cp * sf + si * tj;
  const float h2 = tanhf(c2) * so;
  if (a.gates_act) {
    float* ga = a.gates_act + (size_t)m * N4;
    ga[d] = si; ga[D + d] = tj; ga[2 * D + d] = sf; ga[3 * D + d] = so;
  }
  if (a.c_new) a.c_new[i] = c2;
  if (a.y) a.y[i] = a.mask_out ? (h2 / a.keep_out) * e_mask : h2;
  if (a.c_state) a.c_state[i] = e_fin ? e_cp : c2;
  const float hs = e_fin ? e_hp : h2;
  if (a.h_state) a.h_state[i] = hs;
  if (a.xh_next) a.xh_next[(size_t)m * a.xh_ld + d] = hs;
}

struct InputGradArgs {
  const float* dg;   // [B][4D]
  const float* K;    // backward panel of the [Wd][4D] LSTM kernel
  const float* mask; // [B][E+A] input dropout mask of this step, or NULL
  float keep;
  float* demb;       // [B][E] or NULL
  float* datt;       // [B][A] state gradient (read-modify-write)
  float* dh;         // [B][D] state gradient (accumulated)
  const int32_t* lens;
  int t, carry;
  int B, E, A, D;
};

// grid (ceil(Wd/16), ceil(B/16)): out[m][n] = sum_k dg[m][k] * K[n][k], then the backward of the
// operand concat + input dropout on that element.
__global__ __launch_bounds__(kFusedThreads) void input_grad_fused_kernel(InputGradArgs a) {
  __shared__ float4 red[kFusedWaves - 1][64];
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int r = lane & 15, kq = lane >> 4;
  const int n0 = blockIdx.x * 16, m0 = blockIdx.y * 16;
  const int N4 = 4 * a.D, Wd = a.E + a.A + a.D;
  const bool mok = m0 + r < a.B;
  const float* grow = a.dg + (size_t)(mok ? m0 + r : 0) * N4;
  const int KB = N4 >> 4;   // 4D % 16 == 0 (checked by the host)
  const float* wpanel = a.K + ((size_t)blockIdx.x * KB * 16 + r) * 16 + 4 * kq;
  // epilogue operands of wave 0 (element: row m0 + 4*kq + i, feature c = n0 + r), fetched up front
  const int c = n0 + r, EA = a.E + a.A;
  float e_mk[4] = {1.f, 1.f, 1.f, 1.f}, e_old[4] = {0.f, 0.f, 0.f, 0.f};
  bool e_fin[4] = {false, false, false, false};
  if (wave == 0 && c < Wd) {
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      const int b = m0 + 4 * kq + i;
      if (b >= a.B) continue;
      if (c < EA && a.mask) e_mk[i] = a.mask[(size_t)b * EA + c];
      if (c >= a.E) e_old[i] = c < EA ? a.datt[(size_t)b * a.A + (c - a.E)] : a.dh[(size_t)b * a.D + (c - EA)];
      e_fin[i] = a.lens && a.t >= a.lens[b];
    }
  }
  f32x4_t acc = (f32x4_t){0.f, 0.f, 0.f, 0.f};
  for (int kp0 = wave; 2 * kp0 < KB; kp0 += kFusedWaves * (kChunk / 2)) {
    float4 ga[kChunk], kb_[kChunk];
#pragma unroll
    for (int i = 0; i < kChunk; ++i) {
      const int kb = 2 * (kp0 + kFusedWaves * (i >> 1)) + (i & 1);
      const int k = kb * 16 + 4 * kq;
      ga[i] = kb_[i] = make_float4(0.f, 0.f, 0.f, 0.f);
      if (kb < KB) {
        kb_[i] = *(const float4*)(wpanel + (size_t)kb * 256);
        if (mok) ga[i] = *(const float4*)(grow + k);
      }
    }
#pragma unroll
    for (int i = 0; i < kChunk; ++i) {
      acc = __builtin_amdgcn_mfma_f32_16x16x4f32(ga[i].x, kb_[i].x, acc, 0, 0, 0);
      acc = __builtin_amdgcn_mfma_f32_16x16x4f32(ga[i].y, kb_[i].y, acc, 0, 0, 0);
      acc = __builtin_amdgcn_mfma_f32_16x16x4f32(ga[i].z, kb_[i].z, acc, 0, 0, 0);
      acc = __builtin_amdgcn_mfma_f32_16x16x4f32(ga[i].w, kb_[i].w, acc, 0, 0, 0);
    }
  }
  // D[m][n]: lane holds rows m0 + 4*kq + i (i = 0..3) of column n0 + (lane & 15)
  if (wave > 0) red[wave - 1][lane] = make_float4(acc[0], acc[1], acc[2], acc[3]);
  __syncthreads();
  if (wave != 0) return;
#pragma unroll
  for (int w = 0; w < kFusedWaves - 1; ++w) {
    const float4 p = red[w][lane];
    acc[0] += p.x; acc[1] += p.y; acc[2] += p.z; acc[3] += p.w;
  }
  if (c >= Wd) return;
#pragma unroll
  for (int i = 0; i < 4; ++i) {
    const int b = m0 + 4 * kq + i;
    if (b >= a.B) continue;
    float v = acc[i];
    if (c < EA) {
      if (a.mask) v = (v / a.keep) * e_mk[i];
      if (c < a.E) {
        if (a.demb) a.demb[(size_t)b * a.E + c] = v;
      } else {
        a.datt[(size_t)b * a.A + (c - a.E)] = ((a.carry && !e_fin[i]) ? 0.f : e_old[i]) + v;
      }
    } else {
      a.dh[(size_t)b * a.D + (c - EA)] = e_old[i] + v;
    }
  }
}

// Wider variant (see lstm_step_fused_wide_kernel): NT adjacent feature tiles per workgroup share the
// d-gates fragments; wave j runs the epilogue of tile j.
template <int NT>
__global__ __launch_bounds__(kFusedThreads) void input_grad_fused_wide_kernel(InputGradArgs a) {
  __shared__ float4 red[kFusedWaves][NT][64];
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int r = lane & 15, kq = lane >> 4;
  const int t0 = blockIdx.x * NT, m0 = blockIdx.y * 16;
  const int N4 = 4 * a.D, Wd = a.E + a.A + a.D, EA = a.E + a.A;
  const int n_tiles = (Wd + 15) >> 4;
  const bool mok = m0 + r < a.B;
  const float* grow = a.dg + (size_t)(mok ? m0 + r : 0) * N4;
  const int KB = N4 >> 4;
  const float* wpanel = a.K + ((size_t)t0 * KB * 16 + r) * 16 + 4 * kq;   // tile j: + j*KB*256 floats
  // epilogue operands of wave j < NT (element: row m0 + 4*kq + i, feature c = 16*(t0 + j) + r)
  const int c = 16 * (t0 + wave) + r;
  const bool e_wave = wave < NT && t0 + wave < n_tiles;
  float e_mk[4] = {1.f, 1.f, 1.f, 1.f}, e_old[4] = {0.f, 0.f, 0.f, 0.f};
  bool e_fin[4] = {false, false, false, false};
  if (e_wave && c < Wd) {
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      const int b = m0 + 4 * kq + i;
      if (b >= a.B) continue;
      if (c < EA && a.mask) e_mk[i] = a.mask[(size_t)b * EA + c];
      if (c >= a.E) e_old[i] = c < EA ? a.datt[(size_t)b * a.A + (c - a.E)] : a.dh[(size_t)b * a.D + (c - EA)];
      e_fin[i] = a.lens && a.t >= a.lens[b];
    }
  }
  f32x4_t acc[NT];
#pragma unroll
  for (int j = 0; j < NT; ++j) acc[j] = (f32x4_t){0.f, 0.f, 0.f, 0.f};
  constexpr int CH = 4;
  for (int kp0 = wave; 2 * kp0 < KB; kp0 += kFusedWaves * (CH / 2)) {
    float4 ga[CH], wb[CH][NT];
#pragma unroll
    for (int i = 0; i < CH; ++i) {
      const int kb = 2 * (kp0 + kFusedWaves * (i >> 1)) + (i & 1);
      ga[i] = make_float4(0.f, 0.f, 0.f, 0.f);
      if (kb < KB && mok) ga[i] = *(const float4*)(grow + kb * 16 + 4 * kq);
#pragma unroll
      for (int j = 0; j < NT; ++j) {
        wb[i][j] = make_float4(0.f, 0.f, 0.f, 0.f);
        if (kb < KB && t0 + j < n_tiles) wb[i][j] = *(const float4*)(wpanel + ((size_t)j * KB + kb) * 256);
      }
    }
#pragma unroll
    for (int i = 0; i < CH; ++i) {
#pragma unroll
      for (int j = 0; j < NT; ++j) acc[j] = __builtin_amdgcn_mfma_f32_16x16x4f32(ga[i].x, wb[i][j].x, acc[j], 0, 0, 0);
#pragma unroll
      for (int j = 0; j < NT; ++j) acc[j] = __builtin_amdgcn_mfma_f32_16x16x4f32(ga[i].y, wb[i][j].y, acc[j], 0, 0, 0);
#pragma unroll
      for (int j = 0; j < NT; ++j) acc[j] = __builtin_amdgcn_mfma_f32_16x16x4f32(ga[i].z, wb[i][j].z, acc[j], 0, 0, 0);
#pragma unroll
      for (int j = 0; j < NT; ++j) acc[j] = __builtin_amdgcn_mfma_f32_16x16x4f32(ga[i].w, wb[i][j].w, acc[j], 0, 0, 0);
    }
  }
#pragma unroll
  for (int j = 0; j < NT; ++j) red[wave][j][lane] = make_float4(acc[j][0], acc[j][1], acc[j][2], acc[j][3]);
  __syncthreads();
  if (!e_wave || c >= Wd) return;
  float g[4] = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
  for (int w = 0; w < kFusedWaves; ++w) {
    const float4 p = red[w][wave][lane];
    g[0] += p.x; g[1] += p.y; g[2] += p.z; g[3] += p.w;
  }
#pragma unroll
  for (int i = 0; i < 4; ++i) {
    const int b = m0 + 4 * kq + i;
    if (b >= a.B) continue;
    float v = g[i];
    if (c < EA) {
      if (a.mask) v = (v / a.keep) * e_mk[i];
      if (c < a.E) {
        if (a.demb) a.demb[(size_t)b * a.E + c] = v;
      } else {
        a.datt[(size_t)b * a.A + (c - a.E)] = ((a.carry && !e_fin[i]) ? 0.f : e_old[i]) + v;
      }
    } else {
      a.dh[(size_t)b * a.D + (c - EA)] = e_old[i] + v;
    }
  }
}

struct LstmGradArgs {
  const float* dq;        // [B][D]  d query of this step
  const float* Wq;        // backward panel of W_q [D][D]
  const float* gates_act; // [B][4D]
  const float* c_prev;    // [B][D]
  const float* c_new;     // [B][D]
  const float* dy;        // [B][D]  d cell output from the logits path
  const float* mask_out;
  float keep_out;
  const int32_t* lens;
  int t;
  float* dc_state;
  float* dh_state;
  float* dg;              // [B][4D] d gate pre-activations (out)
  int B, D;
};

// grid (D/16, ceil(B/16)): dy_q[m][n] = sum_k dq[m][k] * W_q[n][k]; epilogue = backward of the output
// dropout and of BasicLSTMCell on element (m, n) (same arithmetic as lstm_gates_bwd_kernel).
__global__ __launch_bounds__(kFusedThreads) void lstm_grad_fused_kernel(LstmGradArgs a) {
  __shared__ float4 red[kFusedWaves - 1][64];
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int r = lane & 15, kq = lane >> 4;
  const int n0 = blockIdx.x * 16, m0 = blockIdx.y * 16;
  const int D = a.D, N4 = 4 * D;
  const bool mok = m0 + r < a.B;
  const float* qrow = a.dq + (size_t)(mok ? m0 + r : 0) * D;
  const int KB = D >> 4;   // D % 16 == 0 (checked by the host)
  const float* wpanel = a.Wq + ((size_t)blockIdx.x * KB * 16 + r) * 16 + 4 * kq;
  // epilogue operands of wave 0: element (row m0 + 4*kq + i, unit d = n0 + r)
  const int d = n0 + r;
  float e_g[4][4], e_cp[4], e_cn[4], e_dy[4], e_mk[4], e_dc[4], e_dh[4], e_live[4];
  if (wave == 0 && d < D) {
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      const int b = m0 + 4 * kq + i;
      const bool ok = b < a.B;
      const size_t e = (size_t)(ok ? b : 0) * D + d;
      const float* ga = a.gates_act + (size_t)(ok ? b : 0) * N4;
      e_g[i][0] = ga[d]; e_g[i][1] = ga[D + d]; e_g[i][2] = ga[2 * D + d]; e_g[i][3] = ga[3 * D + d];
      e_cp[i] = a.c_prev ? a.c_prev[e] : 0.f;
      e_cn[i] = a.c_new[e];
      e_dy[i] = a.dy ? a.dy[e] : 0.f;
      e_mk[i] = a.mask_out ? a.mask_out[e] : 1.f;
      e_dc[i] = a.dc_state[e];
      e_dh[i] = a.dh_state[e];
      e_live[i] = (a.lens && ok && a.t >= a.lens[b]) ? 0.f : 1.f;
    }
  }
  f32x4_t acc = (f32x4_t){0.f, 0.f, 0.f, 0.f};
  for (int kp0 = wave; 2 * kp0 < KB; kp0 += kFusedWaves * (kChunk / 2)) {
    float4 qa[kChunk], wb[kChunk];
#pragma unroll
    for (int i = 0; i < kChunk; ++i) {
      const int kb = 2 * (kp0 + kFusedWaves * (i >> 1)) + (i & 1);
      qa[i] = wb[i] = make_float4(0.f, 0.f, 0.f, 0.f);
      if (kb < KB) {
        wb[i] = *(const float4*)(wpanel + (size_t)kb * 256);
        if (mok) qa[i] = *(const float4*)(qrow + kb * 16 + 4 * kq);
      }
    }
#pragma unroll
    for (int i = 0; i < kChunk; ++i) {
      acc = __builtin_amdgcn_mfma_f32_16x16x4f32(qa[i].x, wb[i].x, acc, 0, 0, 0);
      acc = __builtin_amdgcn_mfma_f32_16x16x4f32(qa[i].y, wb[i].y, acc, 0, 0, 0);
      acc = __builtin_amdgcn_mfma_f32_16x16x4f32(qa[i].z, wb[i].z, acc, 0, 0, 0);
      acc = __builtin_amdgcn_mfma_f32_16x16x4f32(qa[i].w, wb[i].w, acc, 0, 0, 0);
    }
  }
  if (wave > 0) red[wave - 1][lane] = make_float4(acc[0], acc[1], acc[2], acc[3]);
  __syncthreads();
  if (wave != 0 || d >= D) return;
#pragma unroll
  for (int w = 0; w < kFusedWaves - 1; ++w) {
    const float4 p = red[w][lane];
    acc[0] += p.x; acc[1] += p.y; acc[2] += p.z; acc[3] += p.w;
  }
#pragma unroll
  for (int i = 0; i < 4; ++i) {
    const int b = m0 + 4 * kq + i;
    if (b >= a.B) continue;
    const size_t e = (size_t)b * D + d;
    const float si = e_g[i][0], tj = e_g[i][1], sf = e_g[i][2], so = e_g[i][3];
    const float live = e_live[i];
    const float tc = tanhf(e_cn[i]);
    float dyv = e_dy[i] + acc[i];
    if (a.mask_out) dyv = (dyv / a.keep_out) * e_mk[i];
    const float dh2 = e_dh[i] * live + dyv;
    float dc2 = e_dc[i] * live;
    const float dso = dh2 * tc;
    dc2 += dh2 * so * (1.f - tc * tc);
    const float dsf = dc2 * e_cp[i], dsi = dc2 * tj, dtj = dc2 * si;
    float* dgr = a.dg + (size_t)b * N4;
    dgr[d] = dsi * si * (1.f - si);
    dgr[D + d] = dtj * (1.f - tj * tj);
    dgr[2 * D + d] = dsf * sf * (1.f - sf);
    dgr[3 * D + d] = dso * so * (1.f - so);
    a.dc_state[e] = e_dc[i] * (1.f - live) + dc2 * sf;
    a.dh_state[e] = e_dh[i] * (1.f - live);
  }
}

}  // namespace

// 0 when the fused kernels cover this shape (the executors fall back to the split-K chain otherwise)
int comic_fused_step_supported(int D, int Wd) { return (D % 4 == 0 && Wd % 4 == 0) ? 1 : 0; }

// panel sizes in floats (forward, backward) and the repack launch
long comic_lstm_panel_floats(int D, int Wd, int mode) {
  if (mode == 0) return (long)cdiv(D, 4) * cdiv(Wd, 16) * 256;
  return (long)cdiv(Wd, 16) * (4 * D / 16) * 256;
}
int comic_pack_lstm_panels(const float* K, float* fwd_panel, float* bwd_panel, int D, int Wd, hipStream_t st) {
  COMIC_REQUIRE(K && D % 4 == 0, "pack_lstm_panels: bad arguments");
  if (fwd_panel) {
    const long n = comic_lstm_panel_floats(D, Wd, 0);
    hipLaunchKernelGGL(pack_lstm_panels_kernel, dim3((unsigned)cdiv64(n, 256)), dim3(256), 0, st, K, fwd_panel, D, Wd, 0, n,
                       4 * D);
  }
  if (bwd_panel) {
    const long n = comic_lstm_panel_floats(D, Wd, 1);
    hipLaunchKernelGGL(pack_lstm_panels_kernel, dim3((unsigned)cdiv64(n, 256)), dim3(256), 0, st, K, bwd_panel, D, Wd, 1, n,
                       4 * D);
  }
  COMIC_LAUNCH_CHECK("pack_lstm_panels");
  return 0;
}

// W_q [D][D] -> backward panel (D % 16 == 0)
int comic_pack_wq_panel(const float* Wq, float* panel, int D, hipStream_t st) {
  COMIC_REQUIRE(Wq && panel && D % 16 == 0, "pack_wq_panel: D must be a multiple of 16");
  const long n = (long)D * D;
  hipLaunchKernelGGL(pack_lstm_panels_kernel, dim3((unsigned)cdiv64(n, 256)), dim3(256), 0, st, Wq, panel, D, D, 1, n, D);
  COMIC_LAUNCH_CHECK("pack_wq_panel");
  return 0;
}

int comic_lstm_grad_fused(const float* dq, const float* wq_panel, const float* gates_act, const float* c_prev,
                          const float* c_new, const float* dy, const float* mask_out, float keep_out,
                          const int32_t* lens, int t, float* dc_state, float* dh_state, float* dg, int B, int D,
                          hipStream_t st) {
  COMIC_REQUIRE(dq && wq_panel && gates_act && c_new && dc_state && dh_state && dg && B > 0,
                "lstm_grad_fused: bad arguments");
  COMIC_REQUIRE(D % 16 == 0 && ((uintptr_t)dq & 15) == 0, "lstm_grad_fused: D must be a multiple of 16");
  LstmGradArgs a{dq, wq_panel, gates_act, c_prev, c_new, dy, mask_out, keep_out, lens, t, dc_state, dh_state, dg, B, D};
  hipLaunchKernelGGL(lstm_grad_fused_kernel, dim3(D / 16, cdiv(B, 16)), dim3(kFusedThreads), 0, st, a);
  COMIC_LAUNCH_CHECK("lstm_grad_fused");
  return 0;
}

int comic_lstm_step_fused(const float* xh, int ld_xh, const float* K, const float* bias, const float* c_prev,
                          const float* h_prev, float* gates_act, float* c_new, float* y, const float* mask_out,
                          float keep_out, const int32_t* lens, int t, float* c_state, float* h_state, float* xh_next,
                          int xh_ld, int B, int D, int Wd, hipStream_t st) {
  COMIC_REQUIRE(xh && K && B > 0 && D > 0 && Wd > 0, "lstm_step_fused: bad arguments");
  COMIC_REQUIRE(D % 4 == 0 && Wd % 4 == 0 && ld_xh % 4 == 0 && ((uintptr_t)xh & 15) == 0,
                "lstm_step_fused: D, Wd and the operand stride must be multiples of 4 (16-byte rows)");
  LstmStepArgs a{xh, ld_xh, K, bias, c_prev, h_prev, gates_act, c_new, y, mask_out, keep_out, lens, t,
                 c_state, h_state, xh_next, xh_ld, B, D, Wd};
  a.stop = g_comic_stop.p;
  a.stop_t = g_comic_stop.t;
  // rows per workgroup: 16 keeps the most workgroups in flight and measured fastest at 64 rows (training)
  // and at 150-224 rows (beam search); 32 / 64 stay selectable for experiments
  constexpr int mt = 1;
  constexpr int nt_env = 2;     // 2 adjacent unit tiles per workgroup measured fastest (training and beam search)
  if (nt_env == 4) {
    hipLaunchKernelGGL(lstm_step_fused_wide_kernel<4>, dim3(cdiv(D / 4, 4), cdiv(B, 16)), dim3(kFusedThreads), 0, st, a);
  } else if (nt_env == 2) {
    hipLaunchKernelGGL(lstm_step_fused_wide_kernel<2>, dim3(cdiv(D / 4, 2), cdiv(B, 16)), dim3(kFusedThreads), 0, st, a);
  } else if (mt == 4)
    hipLaunchKernelGGL(lstm_step_fused_kernel<4>, dim3(D / 4, cdiv(B, 64)), dim3(kFusedThreads), 0, st, a);
  else if (mt == 2)
    hipLaunchKernelGGL(lstm_step_fused_kernel<2>, dim3(D / 4, cdiv(B, 32)), dim3(kFusedThreads), 0, st, a);
  else
    hipLaunchKernelGGL(lstm_step_fused_kernel<1>, dim3(D / 4, cdiv(B, 16)), dim3(kFusedThreads), 0, st, a);
  COMIC_LAUNCH_CHECK("lstm_step_fused");
  return 0;
}

int comic_input_grad_fused(const float* dg, const float* K, const float* mask, float keep, float* demb, float* datt,
                           float* dh, const int32_t* lens, int t, int carry, int B, int E, int A, int D,
                           hipStream_t st) {
  COMIC_REQUIRE(dg && K && datt && dh && B > 0, "input_grad_fused: bad arguments");
  COMIC_REQUIRE(D % 4 == 0 && ((uintptr_t)dg & 15) == 0 && ((uintptr_t)K & 15) == 0,
                "input_grad_fused: D must be a multiple of 4 and the operands 16-byte aligned");
  InputGradArgs a{dg, K, mask, keep, demb, datt, dh, lens, t, carry, B, E, A, D};
  constexpr int nt = 2;
  const int tiles = cdiv(E + A + D, 16);
  if (nt == 4)
    hipLaunchKernelGGL(input_grad_fused_wide_kernel<4>, dim3(cdiv(tiles, 4), cdiv(B, 16)), dim3(kFusedThreads), 0, st, a);
  else if (nt == 2)
    hipLaunchKernelGGL(input_grad_fused_wide_kernel<2>, dim3(cdiv(tiles, 2), cdiv(B, 16)), dim3(kFusedThreads), 0, st, a);
  else
    hipLaunchKernelGGL(input_grad_fused_kernel, dim3(tiles, cdiv(B, 16)), dim3(kFusedThreads), 0, st, a);
  COMIC_LAUNCH_CHECK("input_grad_fused");
  return 0;
}
